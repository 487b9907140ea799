// evc_h264_stream.cpp -- NAL unit payload handling and the parameter-set / slice-header syntax of ITU-T Rec. H.264
// (7.3.2.1 sequence parameter set, 7.3.2.2 picture parameter set, 7.3.3 slice header, Annex E VUI).
#include <cstdarg>
#include <cstdio>

#include "evc_h264_int.h"

namespace evc {

void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(buf);
}

bool BitReader::more_rbsp_data() const {
    if (pos >= nbits) return false;
    // find the last 1 bit of the payload: everything before it is data
    size_t last = nbits;
    while (last > 0) {
        size_t i = last - 1;
        if ((p[i >> 3] >> (7 - (i & 7))) & 1u) break;
        --last;
    }
    if (last == 0) return false;
    return pos < last - 1;
}

// 7.4.1: removes emulation_prevention_three_byte (00 00 03 -> 00 00)
std::vector<uint8_t> nal_to_rbsp(const uint8_t* d, size_t n) {
    std::vector<uint8_t> out;
    out.reserve(n);
    int zeros = 0;
    for (size_t i = 0; i < n; ++i) {
        if (zeros >= 2 && d[i] == 3) {
            zeros = 0;
            continue;
        }
        out.push_back(d[i]);
        zeros = d[i] == 0 ? zeros + 1 : 0;
    }
    return out;
}

static void skip_hrd(BitReader& b) {
    unsigned cnt = b.ue() + 1;
    if (cnt > 32) fail("sps: cpb_cnt_minus1 out of range");
    b.u(4);
    b.u(4);
    for (unsigned i = 0; i < cnt; ++i) {
        b.ue();
        b.ue();
        b.u1();
    }
    b.u(5);
    b.u(5);
    b.u(5);
    b.u(5);
}

void parse_sps(const uint8_t* rbsp, size_t n, SPS& s, int& sps_id) {
    BitReader b(rbsp, n);
    s = SPS();
    s.profile_idc = b.u(8);
    s.constraint_flags = b.u(8);
    s.level_idc = b.u(8);
    sps_id = b.ue();
    if (sps_id > 31) fail("sps: seq_parameter_set_id %d out of range", sps_id);
    const int p = s.profile_idc;
    if (p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 || p == 118 || p == 128 || p == 138 ||
        p == 139 || p == 134 || p == 135) {
        s.chroma_format_idc = b.ue();
        if (s.chroma_format_idc == 3) b.u1();
        s.bit_depth_luma = 8 + b.ue();
        s.bit_depth_chroma = 8 + b.ue();
        s.transform_bypass = b.u1();
        s.scaling_matrix_present = b.u1();
        if (s.scaling_matrix_present)
            fail("sps: seq_scaling_matrix_present_flag = 1 -- explicit scaling lists are outside this decoder's scope");
    }
    s.log2_max_frame_num = 4 + b.ue();
    s.poc_type = b.ue();
    if (s.poc_type == 0) {
        s.log2_max_poc_lsb = 4 + b.ue();
    } else if (s.poc_type == 1) {
        fail("sps: pic_order_cnt_type 1 is outside this decoder's scope (types 0 and 2 are decoded)");
    } else if (s.poc_type != 2) {
        fail("sps: pic_order_cnt_type %d is not defined", s.poc_type);
    }
    if (s.log2_max_frame_num > 16 || s.log2_max_poc_lsb > 16) fail("sps: log2_max_* out of range");
    s.max_num_ref_frames = b.ue();
    s.gaps_in_frame_num_allowed = b.u1();
    s.mb_w = b.ue() + 1;
    s.mb_h = b.ue() + 1;
    s.frame_mbs_only = b.u1();
    if (!s.frame_mbs_only) fail("sps: frame_mbs_only_flag = 0 -- interlaced / MBAFF coding is outside this decoder's scope");
    s.direct_8x8_inference = b.u1();
    if (b.u1()) {
        int l = b.ue(), r = b.ue(), t = b.ue(), bt = b.ue();
        int ux = s.chroma_format_idc == 0 ? 1 : (s.chroma_format_idc == 3 ? 1 : 2);
        int uy = s.chroma_format_idc == 1 ? 2 : 1;
        s.crop_l = l * ux;
        s.crop_r = r * ux;
        s.crop_t = t * uy;
        s.crop_b = bt * uy;
    }
    if (s.chroma_format_idc != 1) fail("sps: chroma_format_idc %d -- only 4:2:0 is decoded", s.chroma_format_idc);
    if (s.bit_depth_luma != 8 || s.bit_depth_chroma != 8) fail("sps: bit depth %d/%d -- only 8-bit video is decoded", s.bit_depth_luma, s.bit_depth_chroma);
    if (s.transform_bypass) fail("sps: qpprime_y_zero_transform_bypass_flag = 1 is outside this decoder's scope");
    if (s.mb_w < 1 || s.mb_h < 1 || s.mb_w > 1024 || s.mb_h > 1024) fail("sps: picture size %dx%d macroblocks", s.mb_w, s.mb_h);
    if (s.max_num_ref_frames > 16) fail("sps: max_num_ref_frames %d", s.max_num_ref_frames);
    if (s.width() <= 0 || s.height() <= 0) fail("sps: cropping leaves no picture");
    s.vui_present = b.u1();
    if (s.vui_present) {
        if (b.u1()) {  // aspect_ratio_info_present_flag
            if (b.u(8) == 255) {
                b.u(16);
                b.u(16);
            }
        }
        if (b.u1()) b.u1();  // overscan
        if (b.u1()) {        // video_signal_type_present_flag
            b.u(3);
            s.video_full_range = b.u1();
            if (b.u1()) {
                b.u(8);
                b.u(8);
                s.matrix_coefficients = b.u(8);
            }
        }
        if (b.u1()) {  // chroma_loc_info_present_flag
            b.ue();
            b.ue();
        }
        if (b.u1()) {  // timing_info_present_flag
            b.u(32);
            b.u(32);
            b.u1();
        }
        bool nal_hrd = b.u1();
        if (nal_hrd) skip_hrd(b);
        bool vcl_hrd = b.u1();
        if (vcl_hrd) skip_hrd(b);
        if (nal_hrd || vcl_hrd) b.u1();
        b.u1();        // pic_struct_present_flag
        if (b.u1()) {  // bitstream_restriction_flag
            b.u1();
            b.ue();
            b.ue();
            b.ue();
            b.ue();
            s.num_reorder_frames = b.ue();
            s.max_dec_frame_buffering = b.ue();
        }
    }
    s.valid = true;
}

void parse_pps(const uint8_t* rbsp, size_t n, PPS& p, int& pps_id) {
    BitReader b(rbsp, n);
    p = PPS();
    pps_id = b.ue();
    if (pps_id > 255) fail("pps: pic_parameter_set_id %d out of range", pps_id);
    p.sps_id = b.ue();
    if (p.sps_id > 31) fail("pps: seq_parameter_set_id out of range");
    p.cabac = b.u1();
    p.bottom_field_pic_order_present = b.u1();
    if (b.ue() != 0) fail("pps: num_slice_groups_minus1 > 0 -- FMO is outside this decoder's scope");
    p.num_ref_idx_default[0] = b.ue() + 1;
    p.num_ref_idx_default[1] = b.ue() + 1;
    if (p.num_ref_idx_default[0] > 32 || p.num_ref_idx_default[1] > 32) fail("pps: num_ref_idx_default out of range");
    p.weighted_pred = b.u1();
    p.weighted_bipred_idc = b.u(2);
    if (p.weighted_bipred_idc == 3) fail("pps: weighted_bipred_idc 3 is reserved");
    p.pic_init_qp = 26 + b.se();
    b.se();  // pic_init_qs_minus26
    p.chroma_qp_offset[0] = b.se();
    p.chroma_qp_offset[1] = p.chroma_qp_offset[0];
    p.deblocking_control_present = b.u1();
    p.constrained_intra_pred = b.u1();
    p.redundant_pic_cnt_present = b.u1();
    if (b.more_rbsp_data()) {
        p.transform_8x8_mode = b.u1();
        p.scaling_matrix_present = b.u1();
        if (p.scaling_matrix_present)
            fail("pps: pic_scaling_matrix_present_flag = 1 -- explicit scaling lists are outside this decoder's scope");
        p.chroma_qp_offset[1] = b.se();
    }
    if (!p.cabac) fail("pps: entropy_coding_mode_flag = 0 -- CAVLC is outside this decoder's scope (CABAC streams are decoded)");
    if (p.chroma_qp_offset[0] < -12 || p.chroma_qp_offset[0] > 12 || p.chroma_qp_offset[1] < -12 || p.chroma_qp_offset[1] > 12)
        fail("pps: chroma_qp_index_offset out of range");
    p.valid = true;
}

// 7.3.3; `idr` = nal_unit_type 5
void parse_slice_header(BitReader& b, const SPS* spss, const PPS* ppss, int nal_ref_idc, int nal_unit_type, SliceHeader& h) {
    h = SliceHeader();
    h.nal_ref_idc = nal_ref_idc;
    h.nal_unit_type = nal_unit_type;
    h.first_mb = b.ue();
    unsigned st = b.ue();
    if (st > 9) fail("slice: slice_type %u", st);
    st %= 5;
    if (st > 2) fail("slice: SP/SI slices are outside this decoder's scope");
    h.type = (int)st;
    h.pps_id = b.ue();
    if (h.pps_id > 255 || !ppss[h.pps_id].valid) fail("slice: refers to picture parameter set %d which was never sent", h.pps_id);
    const PPS& pps = ppss[h.pps_id];
    if (!spss[pps.sps_id].valid) fail("slice: refers to sequence parameter set %d which was never sent", pps.sps_id);
    const SPS& sps = spss[pps.sps_id];
    h.frame_num = b.u(sps.log2_max_frame_num);
    const bool idr = nal_unit_type == 5;
    if (idr) h.idr_pic_id = b.ue();
    if (sps.poc_type == 0) {
        h.poc_lsb = b.u(sps.log2_max_poc_lsb);
        if (pps.bottom_field_pic_order_present) h.delta_poc_bottom = b.se();
    }
    if (pps.redundant_pic_cnt_present) {
        if (b.ue() != 0) fail("slice: redundant pictures are outside this decoder's scope");
    }
    if (h.type == SLICE_B) h.direct_spatial = b.u1();
    h.num_ref_idx[0] = h.type == SLICE_I ? 0 : pps.num_ref_idx_default[0];
    h.num_ref_idx[1] = h.type == SLICE_B ? pps.num_ref_idx_default[1] : 0;
    if (h.type != SLICE_I) {
        if (b.u1()) {
            h.num_ref_idx[0] = b.ue() + 1;
            if (h.type == SLICE_B) h.num_ref_idx[1] = b.ue() + 1;
        }
        if (h.num_ref_idx[0] > 16 || h.num_ref_idx[1] > 16) fail("slice: num_ref_idx_active out of range for frames");
        for (int l = 0; l < (h.type == SLICE_B ? 2 : 1); ++l) {
            if (b.u1()) {
                for (;;) {
                    unsigned idc = b.ue();
                    if (idc == 3) break;
                    if (idc > 3) fail("slice: modification_of_pic_nums_idc %u", idc);
                    h.mods[l].push_back({(int)idc, (int)b.ue()});
                    if (h.mods[l].size() > 64) fail("slice: reference list modification too long");
                }
            }
        }
    }
    for (int l = 0; l < 2; ++l)
        for (int i = 0; i < 32; ++i) {
            h.luma_w[l][i] = 1;
            h.luma_o[l][i] = 0;
            for (int c = 0; c < 2; ++c) {
                h.chroma_w[l][i][c] = 1;
                h.chroma_o[l][i][c] = 0;
            }
        }
    if ((pps.weighted_pred && h.type == SLICE_P) || (pps.weighted_bipred_idc == 1 && h.type == SLICE_B)) {
        h.luma_log2_denom = b.ue();
        h.chroma_log2_denom = b.ue();
        if (h.luma_log2_denom > 7 || h.chroma_log2_denom > 7) fail("slice: log2_weight_denom out of range");
        for (int l = 0; l < (h.type == SLICE_B ? 2 : 1); ++l)
            for (int i = 0; i < h.num_ref_idx[l]; ++i) {
                h.luma_w[l][i] = 1 << h.luma_log2_denom;
                if (b.u1()) {
                    h.luma_w[l][i] = b.se();
                    h.luma_o[l][i] = b.se();
                }
                h.chroma_w[l][i][0] = h.chroma_w[l][i][1] = 1 << h.chroma_log2_denom;
                if (b.u1()) {
                    for (int c = 0; c < 2; ++c) {
                        h.chroma_w[l][i][c] = b.se();
                        h.chroma_o[l][i][c] = b.se();
                    }
                }
            }
    }
    if (nal_ref_idc != 0) {
        if (idr) {
            h.no_output_of_prior_pics = b.u1();
            h.long_term_reference_flag = b.u1();
        } else {
            h.adaptive_marking = b.u1();
            if (h.adaptive_marking) {
                for (;;) {
                    unsigned op = b.ue();
                    if (op == 0) break;
                    if (op > 6) fail("slice: memory_management_control_operation %u", op);
                    MMCO m{(int)op, 0, 0};
                    if (op == 1 || op == 3) m.a = b.ue();
                    if (op == 2) m.a = b.ue();
                    if (op == 3 || op == 6) m.b = b.ue();
                    if (op == 4) m.a = b.ue();
                    h.mmco.push_back(m);
                    if (h.mmco.size() > 66) fail("slice: too many memory management operations");
                }
            }
        }
    }
    if (h.type != SLICE_I) {
        h.cabac_init_idc = b.ue();
        if (h.cabac_init_idc > 2) fail("slice: cabac_init_idc %d", h.cabac_init_idc);
    }
    h.qp = pps.pic_init_qp + b.se();
    if (h.qp < 0 || h.qp > 51) fail("slice: SliceQPY %d out of range", h.qp);
    if (pps.deblocking_control_present) {
        h.disable_deblock = b.ue();
        if (h.disable_deblock > 2) fail("slice: disable_deblocking_filter_idc %d", h.disable_deblock);
        if (h.disable_deblock != 1) {
            h.alpha_off = 2 * b.se();
            h.beta_off = 2 * b.se();
            if (h.alpha_off < -12 || h.alpha_off > 12 || h.beta_off < -12 || h.beta_off > 12) fail("slice: deblocking offsets out of range");
        }
    }
    h.data_bit_pos = b.pos;
}

void Picture::alloc(int mbw, int mbh) {
    mb_w = mbw;
    mb_h = mbh;
    stride = mbw * 16;
    cstride = mbw * 8;
    Y.assign((size_t)stride * mbh * 16, 0);
    Cb.assign((size_t)cstride * mbh * 8, 0);
    Cr.assign((size_t)cstride * mbh * 8, 0);
    size_t n4 = (size_t)mbw * 4 * mbh * 4;
    for (int l = 0; l < 2; ++l) {
        mv[l].assign(n4 * 2, 0);
        ref[l].assign(n4, -1);
        ref_id[l].assign(n4, -1);
    }
    mb_intra.assign((size_t)mbw * mbh, 0);
}

}  // namespace evc
