// evc_h264_decoder.cpp -- picture-level decoding process of ITU-T Rec. H.264: NAL unit dispatch (7.3.1), detection of
// the first slice of a picture (7.4.1.2.4), picture order count (8.2.1), reference picture list construction and
// modification (8.2.4), reference picture marking (8.2.5) and output in picture-order-count order (C.4.5.3 "bumping").
#include <algorithm>
#include <cstdlib>

#include "evc_h264_int.h"

namespace evc {

std::vector<uint8_t> nal_to_rbsp(const uint8_t* d, size_t n);
void parse_sps(const uint8_t* rbsp, size_t n, SPS& s, int& sps_id);
void parse_pps(const uint8_t* rbsp, size_t n, PPS& p, int& pps_id);
void parse_slice_header(BitReader& b, const SPS* spss, const PPS* ppss, int nal_ref_idc, int nal_unit_type, SliceHeader& h);

struct DecoderImpl {
    SPS sps[32];
    PPS pps[256];
    const SPS* asps = nullptr;
    std::vector<PicPtr> dpb;      // pictures marked "used for reference"
    std::vector<PicPtr> pending;  // decoded, waiting for output
    std::vector<PicPtr> out;
    PicPtr cur;
    SliceHeader cur_sh;  // header of the first slice of the current picture
    const PPS* cur_pps = nullptr;
    std::vector<MbInfo> mbi;
    std::vector<int8_t> ipred;
    std::vector<int16_t> mvd[2];
    std::vector<uint8_t> direct4;
    int slice_count = 0, mbs_decoded = 0;
    int prev_poc_msb = 0, prev_poc_lsb = 0, prev_frame_num = 0, prev_frame_num_offset = 0, prev_ref_frame_num = 0;
    int next_id = 0, idr_epoch = 0;
    bool cur_mmco5 = false;
    Stats stats;

    int max_frame_num() const { return 1 << asps->log2_max_frame_num; }

    // ---------------------------------------------------------------------------------------------- output
    int reorder_depth() const {
        if (asps && asps->num_reorder_frames >= 0) return asps->num_reorder_frames;
        return 16;
    }
    void bump(bool all) {
        while (!pending.empty() && (all || (int)pending.size() > reorder_depth())) {
            size_t best = 0;
            for (size_t i = 1; i < pending.size(); ++i)
                if (pending[i]->poc < pending[best]->poc) best = i;
            out.push_back(pending[best]);
            pending.erase(pending.begin() + best);
        }
    }

    // ---------------------------------------------------------------------------------------------- picture start / end
    void start_picture(const SliceHeader& h, const PPS& p) {
        const SPS& s = sps[p.sps_id];
        if (asps != &s || !cur) {
            if (asps && asps != &s && h.nal_unit_type != 5) fail("a new sequence parameter set becomes active outside an IDR picture");
        }
        asps = &s;
        const bool idr = h.nal_unit_type == 5;
        if (idr) {
            bump(true);
            dpb.clear();
            ++idr_epoch;
        } else if (dpb.empty() && next_id == 0) {
            fail("the stream does not start with an IDR picture");
        }
        const int maxfn = max_frame_num();
        if (!idr && h.frame_num != prev_ref_frame_num && h.frame_num != (prev_ref_frame_num + 1) % maxfn)
            fail("gap in frame_num (%d after %d): frames are missing from the stream", h.frame_num, prev_ref_frame_num);
        cur = std::make_shared<Picture>();
        cur->alloc(s.mb_w, s.mb_h);
        cur->id = next_id++;
        cur->idr_epoch = idr_epoch;
        cur->frame_num = h.frame_num;
        cur->is_idr = idr;
        cur->slice_type_first = h.type;
        // 8.2.1
        if (s.poc_type == 0) {
            const int maxlsb = 1 << s.log2_max_poc_lsb;
            int pmsb = idr ? 0 : prev_poc_msb, plsb = idr ? 0 : prev_poc_lsb;
            int msb;
            if (h.poc_lsb < plsb && plsb - h.poc_lsb >= maxlsb / 2)
                msb = pmsb + maxlsb;
            else if (h.poc_lsb > plsb && h.poc_lsb - plsb > maxlsb / 2)
                msb = pmsb - maxlsb;
            else
                msb = pmsb;
            int top = msb + h.poc_lsb, bottom = top + h.delta_poc_bottom;
            cur->poc = std::min(top, bottom);
            if (h.nal_ref_idc) {
                prev_poc_msb = msb;
                prev_poc_lsb = h.poc_lsb;
            }
        } else {
            int off = idr ? 0 : (prev_frame_num > h.frame_num ? prev_frame_num_offset + maxfn : prev_frame_num_offset);
            cur->poc = idr ? 0 : (h.nal_ref_idc ? 2 * (off + h.frame_num) : 2 * (off + h.frame_num) - 1);
            prev_frame_num_offset = off;
            prev_frame_num = h.frame_num;
        }
        size_t nmb = (size_t)s.mb_w * s.mb_h;
        mbi.assign(nmb, MbInfo());
        for (auto& m : mbi) m.slice_id = 0xFFFF;
        ipred.assign(nmb * 16, -1);
        for (int l = 0; l < 2; ++l) mvd[l].assign(nmb * 32, 0);
        direct4.assign(nmb * 16, 0);
        slice_count = 0;
        mbs_decoded = 0;
        cur_sh = h;
        cur_pps = &p;
        cur_mmco5 = false;
    }

    void finish_picture() {
        if (!cur) return;
        const SPS& s = *asps;
        if (mbs_decoded != s.mb_w * s.mb_h)
            fail("picture %d (frame_num %d): %d of %d macroblocks were present in its slices", cur->id, cur->frame_num, mbs_decoded, s.mb_w * s.mb_h);
        deblock_picture(*cur, mbi, *cur_pps);
        const SliceHeader& h = cur_sh;
        // 8.2.5
        if (h.nal_ref_idc) {
            if (h.nal_unit_type == 5) {
                dpb.clear();
                cur->is_ref = true;
                if (h.long_term_reference_flag) {
                    cur->is_long = true;
                    cur->long_term_idx = 0;
                    ++stats.long_term;
                }
            } else {
                cur->is_ref = true;
                if (h.adaptive_marking) {
                    apply_mmco(h);
                } else {
                    sliding_window();
                }
            }
            dpb.push_back(cur);
            if ((int)dpb.size() > std::max(s.max_num_ref_frames, 1)) {
                // adaptive marking may leave the buffer over-full only in a non-conforming stream
                fail("reference picture buffer overflow (%zu > max_num_ref_frames %d)", dpb.size(), s.max_num_ref_frames);
            }
            prev_ref_frame_num = cur_mmco5 ? 0 : h.frame_num;
        }
        if (cur_mmco5) {
            // 8.2.1: after memory_management_control_operation 5 the picture is treated as having POC 0 / frame_num 0
            bump(true);
            cur->poc = 0;
            cur->frame_num = 0;
            prev_poc_msb = 0;
            prev_poc_lsb = 0;
            prev_frame_num_offset = 0;
            prev_frame_num = 0;
            ++idr_epoch;
        }
        pending.push_back(cur);
        cur.reset();
        bump(false);
    }

    void unmark(Picture* p) {
        p->is_ref = false;
        p->is_long = false;
        for (size_t i = 0; i < dpb.size(); ++i)
            if (dpb[i].get() == p) {
                dpb.erase(dpb.begin() + i);
                return;
            }
    }

    void compute_frame_num_wrap(int frame_num) {
        for (auto& p : dpb)
            if (!p->is_long) p->frame_num_wrap = p->frame_num > frame_num ? p->frame_num - max_frame_num() : p->frame_num;
    }

    void sliding_window() {
        const int limit = std::max(asps->max_num_ref_frames, 1);
        if ((int)dpb.size() < limit) return;
        compute_frame_num_wrap(cur->frame_num);
        Picture* victim = nullptr;
        for (auto& p : dpb)
            if (!p->is_long && (!victim || p->frame_num_wrap < victim->frame_num_wrap)) victim = p.get();
        if (!victim) fail("sliding window marking: the buffer holds only long-term pictures");
        unmark(victim);
    }

    void apply_mmco(const SliceHeader& h) {
        compute_frame_num_wrap(cur->frame_num);
        for (const MMCO& m : h.mmco) {
            ++stats.mmco_ops;
            switch (m.op) {
                case 1: {
                    int picnum = h.frame_num - (m.a + 1);
                    Picture* t = nullptr;
                    for (auto& p : dpb)
                        if (!p->is_long && p->frame_num_wrap == picnum) t = p.get();
                    if (!t) fail("MMCO 1 names short-term picture %d, which is not in the buffer", picnum);
                    unmark(t);
                    break;
                }
                case 2: {
                    Picture* t = nullptr;
                    for (auto& p : dpb)
                        if (p->is_long && p->long_term_idx == m.a) t = p.get();
                    if (!t) fail("MMCO 2 names long-term picture %d, which is not in the buffer", m.a);
                    unmark(t);
                    break;
                }
                case 3: {
                    int picnum = h.frame_num - (m.a + 1);
                    Picture* t = nullptr;
                    for (auto& p : dpb)
                        if (!p->is_long && p->frame_num_wrap == picnum) t = p.get();
                    if (!t) fail("MMCO 3 names short-term picture %d, which is not in the buffer", picnum);
                    for (auto& p : dpb)
                        if (p->is_long && p->long_term_idx == m.b && p.get() != t) {
                            unmark(p.get());
                            break;
                        }
                    t->is_long = true;
                    t->long_term_idx = m.b;
                    ++stats.long_term;
                    break;
                }
                case 4: {
                    for (size_t i = 0; i < dpb.size();) {
                        if (dpb[i]->is_long && dpb[i]->long_term_idx >= m.a)
                            unmark(dpb[i].get());
                        else
                            ++i;
                    }
                    break;
                }
                case 5:
                    while (!dpb.empty()) unmark(dpb.back().get());
                    cur_mmco5 = true;
                    break;
                case 6: {
                    for (auto& p : dpb)
                        if (p->is_long && p->long_term_idx == m.b) {
                            unmark(p.get());
                            break;
                        }
                    cur->is_long = true;
                    cur->long_term_idx = m.b;
                    ++stats.long_term;
                    break;
                }
            }
        }
    }

    // ---------------------------------------------------------------------------------------------- reference lists
    void build_lists(const SliceHeader& h, SliceCtx& sc) {
        sc.list[0].clear();
        sc.list[1].clear();
        if (h.type == SLICE_I) return;
        compute_frame_num_wrap(h.frame_num);
        std::vector<Picture*> st, lt;
        for (auto& p : dpb) (p->is_long ? lt : st).push_back(p.get());
        std::sort(lt.begin(), lt.end(), [](Picture* a, Picture* b) { return a->long_term_idx < b->long_term_idx; });
        std::vector<Picture*> L[2];
        if (h.type == SLICE_P) {
            std::sort(st.begin(), st.end(), [](Picture* a, Picture* b) { return a->frame_num_wrap > b->frame_num_wrap; });
            L[0] = st;
            L[0].insert(L[0].end(), lt.begin(), lt.end());
        } else {
            const int poc = cur->poc;
            std::vector<Picture*> before, after;
            for (Picture* p : st) (p->poc < poc ? before : after).push_back(p);
            std::sort(before.begin(), before.end(), [](Picture* a, Picture* b) { return a->poc > b->poc; });
            std::sort(after.begin(), after.end(), [](Picture* a, Picture* b) { return a->poc < b->poc; });
            L[0] = before;
            L[0].insert(L[0].end(), after.begin(), after.end());
            L[0].insert(L[0].end(), lt.begin(), lt.end());
            L[1] = after;
            L[1].insert(L[1].end(), before.begin(), before.end());
            L[1].insert(L[1].end(), lt.begin(), lt.end());
            if (L[1].size() > 1 && L[1] == L[0]) std::swap(L[1][0], L[1][1]);
        }
        const int curr_pic_num = h.frame_num, max_pic_num = max_frame_num();
        for (int l = 0; l < (h.type == SLICE_B ? 2 : 1); ++l) {
            const int n = h.num_ref_idx[l];
            std::vector<Picture*>& R = sc.list[l];
            R = L[l];
            R.resize((size_t)n + 1, nullptr);  // one spare entry for the insertion procedure of 8.2.4.3
            R[n] = nullptr;
            if (!h.mods[l].empty()) {
                ++stats.list_mods;
                int pred = curr_pic_num, idx = 0;
                for (const auto& m : h.mods[l]) {
                    if (idx >= n) fail("reference list modification has more entries than the list");
                    Picture* t = nullptr;
                    if (m.idc == 0 || m.idc == 1) {
                        int nowrap;
                        if (m.idc == 0) {
                            nowrap = pred - (m.val + 1);
                            if (nowrap < 0) nowrap += max_pic_num;
                        } else {
                            nowrap = pred + (m.val + 1);
                            if (nowrap >= max_pic_num) nowrap -= max_pic_num;
                        }
                        pred = nowrap;
                        int picnum = nowrap > curr_pic_num ? nowrap - max_pic_num : nowrap;
                        for (Picture* p : st)
                            if (p->frame_num_wrap == picnum) t = p;
                        if (!t) fail("reference list modification names short-term picture %d, which is not in the buffer", picnum);
                    } else {
                        for (Picture* p : lt)
                            if (p->long_term_idx == m.val) t = p;
                        if (!t) fail("reference list modification names long-term picture %d, which is not in the buffer", m.val);
                    }
                    for (int c = n; c > idx; --c) R[c] = R[c - 1];
                    R[idx++] = t;
                    int nidx = idx;
                    for (int c = idx; c <= n; ++c)
                        if (R[c] != t) R[nidx++] = R[c];
                    for (int c = nidx; c <= n; ++c) R[c] = nullptr;
                }
            }
            R.resize(n);
        }
    }

    void build_weights(const SliceHeader& h, const PPS& p, SliceCtx& sc) {
        sc.wt.mode = 0;
        for (int i = 0; i < 32; ++i) sc.dist_scale[i] = 256;
        if (h.type == SLICE_P && p.weighted_pred) sc.wt.mode = 1;
        if (h.type == SLICE_B && p.weighted_bipred_idc == 1) sc.wt.mode = 1;
        if (h.type != SLICE_B) return;
        auto dsf = [&](const Picture* p0, const Picture* p1, bool& ok) {
            int tb = clip3(-128, 127, cur->poc - p0->poc), td = clip3(-128, 127, p1->poc - p0->poc);
            if (td == 0 || p0->is_long || p1->is_long) {
                ok = false;
                return 256;
            }
            int tx = (16384 + std::abs(td / 2)) / td;
            ok = true;
            return clip3(-1024, 1023, (tb * tx + 32) >> 6);
        };
        if (p.weighted_bipred_idc == 2) {
            sc.wt.mode = 2;
            for (int i = 0; i < (int)sc.list[0].size(); ++i)
                for (int j = 0; j < (int)sc.list[1].size(); ++j) {
                    int w0 = 32;
                    if (sc.list[0][i] && sc.list[1][j]) {
                        bool ok;
                        int d = dsf(sc.list[0][i], sc.list[1][j], ok);
                        if (ok && (d >> 2) >= -64 && (d >> 2) <= 128) w0 = 64 - (d >> 2);
                    }
                    sc.wt.implicit_w0[i][j] = w0;
                }
        }
        if (!h.direct_spatial && !sc.list[1].empty() && sc.list[1][0]) {
            for (int i = 0; i < (int)sc.list[0].size(); ++i)
                if (sc.list[0][i]) {
                    bool ok;
                    sc.dist_scale[i] = dsf(sc.list[0][i], sc.list[1][0], ok);
                }
        }
    }

    // ---------------------------------------------------------------------------------------------- NAL units
    void decode_slice_nal(const uint8_t* nal, size_t size, int ref_idc, int type) {
        std::vector<uint8_t> rbsp = nal_to_rbsp(nal + 1, size - 1);
        BitReader b(rbsp.data(), rbsp.size());
        SliceHeader h;
        parse_slice_header(b, sps, pps, ref_idc, type, h);
        const PPS& p = pps[h.pps_id];
        bool new_pic = !cur;
        if (cur) {
            const SliceHeader& f = cur_sh;
            new_pic = f.frame_num != h.frame_num || f.pps_id != h.pps_id || (f.nal_ref_idc == 0) != (h.nal_ref_idc == 0) ||
                      (f.nal_unit_type == 5) != (h.nal_unit_type == 5) || f.poc_lsb != h.poc_lsb || f.delta_poc_bottom != h.delta_poc_bottom ||
                      (h.nal_unit_type == 5 && f.idr_pic_id != h.idr_pic_id);
        }
        if (new_pic) {
            finish_picture();
            start_picture(h, p);
        }
        ++stats.slices[h.type];
        SliceCtx sc;
        sc.sps = asps;
        sc.pps = &p;
        sc.sh = &h;
        sc.cur = cur.get();
        build_lists(h, sc);
        for (int l = 0; l < 2; ++l)
            for (int i = 0; i < h.num_ref_idx[l]; ++i)
                if (i >= (int)sc.list[l].size() || !sc.list[l][i])
                    fail("picture %d: RefPicList%d[%d] has no picture (the stream refers to frames that were not decoded)", cur->id, l, i);
        build_weights(h, p, sc);
        sc.mbi = &mbi;
        sc.ipred = &ipred;
        sc.mvd[0] = &mvd[0];
        sc.mvd[1] = &mvd[1];
        sc.direct4 = &direct4;
        sc.slice_id = slice_count++;
        if (sc.slice_id >= 0xFFFF) fail("too many slices in one picture");
        sc.stats = &stats;
        // 7.3.4: cabac_alignment_one_bit up to the next byte boundary
        while (!b.aligned())
            if (b.u1() != 1) fail("slice: cabac_alignment_one_bit is 0");
        sc.data = rbsp.data() + b.pos / 8;
        sc.data_end = rbsp.data() + rbsp.size();
        if (h.first_mb >= asps->mb_w * asps->mb_h) fail("slice: first_mb_in_slice %d outside the picture", h.first_mb);
        mbs_decoded += decode_slice_data(sc);
    }

    void decode_nal(const uint8_t* d, size_t n) {
        if (n < 1) return;
        if (d[0] & 0x80) fail("NAL unit with forbidden_zero_bit set");
        int ref_idc = (d[0] >> 5) & 3, type = d[0] & 31;
        switch (type) {
            case 1:
            case 5:
                decode_slice_nal(d, n, ref_idc, type);
                break;
            case 2:
            case 3:
            case 4:
                fail("slice data partitioning (nal_unit_type %d) is outside this decoder's scope", type);
            case 7: {
                std::vector<uint8_t> r = nal_to_rbsp(d + 1, n - 1);
                SPS s;
                int id;
                parse_sps(r.data(), r.size(), s, id);
                if (cur && asps == &sps[id]) finish_picture();
                sps[id] = s;
                break;
            }
            case 8: {
                std::vector<uint8_t> r = nal_to_rbsp(d + 1, n - 1);
                PPS p;
                int id;
                parse_pps(r.data(), r.size(), p, id);
                if (cur && cur_pps == &pps[id]) finish_picture();
                pps[id] = p;
                break;
            }
            case 9:   // access unit delimiter
            case 10:  // end of sequence
            case 11:  // end of stream
                finish_picture();
                break;
            default:  // SEI (6), filler (12), extensions: nothing the decoding process needs
                break;
        }
    }
};

Decoder::Decoder() : d(new DecoderImpl) {}
Decoder::~Decoder() = default;
void Decoder::decode_nal(const uint8_t* data, size_t size) { d->decode_nal(data, size); }
void Decoder::flush() {
    d->finish_picture();
    d->bump(true);
}
std::vector<PicPtr> Decoder::take_output() {
    std::vector<PicPtr> o;
    o.swap(d->out);
    return o;
}
const SPS* Decoder::active_sps() const { return d->asps; }
const Stats& Decoder::stats() const { return d->stats; }

}  // namespace evc
