// evc_h264.h -- internal declarations of the capture source's H.264 decoder (host C++, no GPU, no third-party code).
//
// What it is for: /root/reference/evenvizion/examples/evenvizion_component.py:132 opens the video with
// cv2.VideoCapture(path) and video_processing.py:58,70 pulls frames with capture.read().  This image has no OpenCV and
// no FFmpeg, so the capture source is written here from ITU-T Rec. H.264 (the decoding process is normative: a
// conforming decoder's pictures are defined bit for bit by the specification, clause 8).
//
// Scope (deliberate): the tools the reference's own test_video.mp4 and comparable x264/MP4 camera files use --
//   progressive frames only (frame_mbs_only_flag = 1), 4:2:0 8-bit, CABAC, I/P/B slices, 4x4 and 8x8 transforms,
//   flat or explicit scaling lists, explicit (P) and implicit (B) weighted prediction, spatial and temporal direct,
//   multiple reference frames with list modification and MMCO, in-loop deblocking, frame cropping, POC types 0 and 2.
// Anything else (CAVLC, interlace/MBAFF, FMO/ASO, 4:2:2/4:4:4, >8 bit, SP/SI, data partitioning) is rejected with an
// explicit error, never decoded approximately.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace evc {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void fail(const char* fmt, ...);

// ------------------------------------------------------------------------------------------------ bit reader (7.2)
struct BitReader {
    const uint8_t* p = nullptr;
    size_t nbits = 0, pos = 0;
    BitReader() = default;
    BitReader(const uint8_t* d, size_t nbytes) : p(d), nbits(nbytes * 8) {}
    unsigned u1() {
        if (pos >= nbits) fail("bitstream: read past the end of the RBSP");
        unsigned b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
        ++pos;
        return b;
    }
    unsigned u(int n) {
        unsigned v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | u1();
        return v;
    }
    unsigned ue() {
        int z = 0;
        while (u1() == 0) {
            if (++z > 32) fail("bitstream: Exp-Golomb prefix too long");
        }
        return z ? ((1u << z) - 1u + u(z)) : 0u;
    }
    int se() {
        unsigned k = ue();
        return (k & 1) ? int((k + 1) >> 1) : -int(k >> 1);
    }
    bool aligned() const { return (pos & 7) == 0; }
    // 7.2 more_rbsp_data(): something other than the rbsp_trailing_bits remains
    bool more_rbsp_data() const;
};

// ------------------------------------------------------------------------------------------------ parameter sets
struct SPS {
    bool valid = false;
    int profile_idc = 0, level_idc = 0, constraint_flags = 0;
    int chroma_format_idc = 1, bit_depth_luma = 8, bit_depth_chroma = 8;
    bool transform_bypass = false;
    bool scaling_matrix_present = false;
    uint8_t scaling4[6][16], scaling8[2][64];  // zig-zag order as transmitted, after the fall-back rules
    int log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4;
    bool delta_pic_order_always_zero = false;
    int max_num_ref_frames = 1;
    bool gaps_in_frame_num_allowed = false;
    int mb_w = 0, mb_h = 0;
    bool frame_mbs_only = true, direct_8x8_inference = true;
    int crop_l = 0, crop_r = 0, crop_t = 0, crop_b = 0;  // in luma samples
    bool vui_present = false;
    int num_reorder_frames = -1, max_dec_frame_buffering = -1;
    bool video_full_range = false;
    int matrix_coefficients = 2;  // 2 = unspecified
    int width() const { return mb_w * 16 - crop_l - crop_r; }
    int height() const { return mb_h * 16 - crop_t - crop_b; }
};

struct PPS {
    bool valid = false;
    int sps_id = 0;
    bool cabac = false, bottom_field_pic_order_present = false;
    int num_ref_idx_default[2] = {1, 1};
    bool weighted_pred = false;
    int weighted_bipred_idc = 0;
    int pic_init_qp = 26, chroma_qp_offset[2] = {0, 0};
    bool deblocking_control_present = false, constrained_intra_pred = false, redundant_pic_cnt_present = false;
    bool transform_8x8_mode = false;
    bool scaling_matrix_present = false;
    uint8_t scaling4[6][16], scaling8[2][64];
};

enum SliceType { SLICE_P = 0, SLICE_B = 1, SLICE_I = 2 };

struct Picture;

struct MMCO {
    int op, a, b;
};

struct SliceHeader {
    int first_mb = 0, type = SLICE_I, pps_id = 0, frame_num = 0, idr_pic_id = 0;
    int nal_ref_idc = 0, nal_unit_type = 1;
    int poc_lsb = 0, delta_poc_bottom = 0;
    bool direct_spatial = true;
    int num_ref_idx[2] = {0, 0};
    struct Mod {
        int idc, val;
    };
    std::vector<Mod> mods[2];
    // explicit weights (7.3.3.2); [list][refIdx]
    int luma_log2_denom = 0, chroma_log2_denom = 0;
    int luma_w[2][32], luma_o[2][32], chroma_w[2][32][2], chroma_o[2][32][2];
    bool no_output_of_prior_pics = false, long_term_reference_flag = false, adaptive_marking = false;
    std::vector<MMCO> mmco;
    int cabac_init_idc = 0, qp = 26, disable_deblock = 0, alpha_off = 0, beta_off = 0;
    size_t data_bit_pos = 0;  // where slice_data() starts in the RBSP
};

// ------------------------------------------------------------------------------------------------ pictures
// Motion data is kept per 4x4 luma block for the whole picture: it is what co-located (direct) prediction and the
// deblocking filter read after the picture is finished.
struct Picture {
    int mb_w = 0, mb_h = 0, stride = 0, cstride = 0;  // planes cover the coded size (mb_w*16 x mb_h*16)
    std::vector<uint8_t> Y, Cb, Cr;
    std::vector<int16_t> mv[2];       // [list][(y4*w4 + x4)*2 + c]
    std::vector<int8_t> ref[2];       // [list][y4*w4 + x4]; -1 = list not used, -2 = intra
    std::vector<int32_t> ref_id[2];   // Picture::id of the referenced picture per 4x4 block (direct, deblocking)
    std::vector<uint8_t> mb_intra;    // per macroblock
    int poc = 0, frame_num = 0, frame_num_wrap = 0, long_term_idx = -1;
    bool is_ref = false, is_long = false, is_idr = false;
    int id = 0;  // unique per decoded picture (decode order)
    int idr_epoch = 0;
    bool output_done = false;
    int slice_type_first = SLICE_I;
    void alloc(int mbw, int mbh);
};
using PicPtr = std::shared_ptr<Picture>;

// ------------------------------------------------------------------------------------------------ CABAC (9.3)
struct Cabac {
    const uint8_t* p = nullptr;
    const uint8_t* end = nullptr;
    uint32_t range = 0, offset = 0;
    int bits_left = 0;  // bits of *p not yet consumed
    uint8_t state[1024];  // (pStateIdx << 1) | valMPS
    void init_engine(const uint8_t* data, const uint8_t* data_end);
    void init_contexts(int slice_type, int cabac_init_idc, int slice_qp);
    int read_bit() {
        if (bits_left == 0) {
            if (p >= end) {
                // 9.3.1.2: a conforming stream never needs bits beyond the slice; count the overrun and feed zeros.
                ++overrun;
                return 0;
            }
            cur = *p++;
            bits_left = 8;
        }
        --bits_left;
        return (cur >> bits_left) & 1;
    }
    int decision(int ctx);
    int bypass();
    int terminate();
    uint8_t cur = 0;
    int overrun = 0;
};

// ------------------------------------------------------------------------------------------------ decoder statistics
// Which tools the stream exercised; the capture source reports them so a test can say what has and has not been run.
struct Stats {
    long mbs = 0, i4 = 0, i8 = 0, i16 = 0, ipcm = 0, p_skip = 0, b_skip = 0, b_direct = 0, inter = 0;
    long t8x8 = 0, bipred_blocks = 0, explicit_wp_blocks = 0, implicit_wp_blocks = 0, sub8x8 = 0;
    long temporal_direct_mbs = 0, spatial_direct_mbs = 0, mmco_ops = 0, list_mods = 0, long_term = 0;
    long slices[3] = {0, 0, 0};
    long cabac_overrun = 0;
    int max_ref_idx = 0;
    int cabac_idc_used[4] = {0, 0, 0, 0};  // [3] = I slices
};

struct DecoderImpl;
class Decoder {
  public:
    Decoder();
    ~Decoder();
    // One NAL unit without start code / length prefix (emulation prevention bytes still inside).
    void decode_nal(const uint8_t* data, size_t size);
    // Marks the end of the stream: everything still waiting for output is released.
    void flush();
    // Pictures in output (POC) order that have become available.
    std::vector<PicPtr> take_output();
    const SPS* active_sps() const;
    const Stats& stats() const;

  private:
    std::unique_ptr<DecoderImpl> d;
};

}  // namespace evc
