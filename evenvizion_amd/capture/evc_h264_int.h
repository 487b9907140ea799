// evc_h264_int.h -- structures shared by the slice decoder, the reconstruction routines and the deblocking filter.
#pragma once
#include "evc_h264.h"

namespace evc {

// ---- tables (evc_h264_tables.cpp); clause numbers refer to ITU-T Rec. H.264
extern const uint8_t kRangeTabLPS[64][4];     // Table 9-44
extern const uint8_t kTransIdxLPS[64];        // Table 9-45
extern const int8_t kCabacInitI[460][2];      // Tables 9-12..9-23, 9-24.., I slices (m, n)
extern const int8_t kCabacInitPB0[460][2];    // same, cabac_init_idc 0
extern const uint8_t kZigzag4x4[16];          // 8.5.6, frame scan: k -> raster index x + 4y
extern const uint8_t kZigzag8x8[64];          // 8.5.7, frame scan
extern const uint8_t kSigCtx8x8[63];          // Table 9-43, frame coded
extern const uint8_t kLastCtx8x8[63];         // Table 9-43
extern const uint8_t kNormAdjust4x4[6][3];    // 8.5.9
extern const uint8_t kNormAdjust8x8[6][6];    // 8.5.9
extern const uint8_t kChromaQp[52];           // Table 8-15
extern const uint8_t kAlpha[52], kBeta[52];   // Table 8-16
extern const uint8_t kTc0[52][3];             // Table 8-17 (bS = 1, 2, 3)

// ---- per-macroblock record that lives for the whole picture (CABAC context derivation, deblocking)
struct MbInfo {
    uint16_t slice_id;       // 0xFFFF = not decoded yet
    uint8_t intra, inxn, i16, ipcm, skip, direct16, t8x8;
    uint8_t cbp;             // bits 0..3 luma 8x8 blocks, bits 4..5 chroma (0,1,2)
    uint8_t chroma_pred_mode;
    uint8_t cbf_dc;          // bit 0 Intra16x16 luma DC, bit 1 Cb DC, bit 2 Cr DC
    uint8_t cbf_cac[2];      // chroma AC coded_block_flag, bit per 4x4 chroma block
    uint16_t cbf_luma;       // bit per 4x4 luma block (raster inside the macroblock); 8x8 blocks set all four
    int8_t qp, qpc[2];
    int8_t disable_deblock, alpha_off, beta_off;  // of the slice that holds the macroblock
};

struct SliceWeights {
    // mode 0: default, 1: explicit, 2: implicit
    int mode = 0;
    int implicit_w0[32][32];  // [refIdxL0][refIdxL1] -> w0 (w1 = 64 - w0)
};

// Everything one slice needs; built by DecoderImpl, consumed by decode_slice_data().
struct SliceCtx {
    const SPS* sps;
    const PPS* pps;
    const SliceHeader* sh;
    Picture* cur;
    std::vector<Picture*> list[2];  // RefPicList0/1 (nullptr = missing reference)
    SliceWeights wt;
    int dist_scale[32];             // temporal direct DistScaleFactor per refIdxL0 (against list1[0])
    std::vector<MbInfo>* mbi;
    std::vector<int8_t>* ipred;     // per 4x4: Intra4x4/8x8PredMode, -1 elsewhere
    std::vector<int16_t>* mvd[2];   // per 4x4, |mvd| pairs
    std::vector<uint8_t>* direct4;  // per 4x4: predicted by direct (B)
    int slice_id;
    Stats* stats;
    const uint8_t* data;
    const uint8_t* data_end;  // RBSP bytes of the slice data (byte aligned start)
};

// returns the number of macroblocks decoded; throws on a stream it cannot follow
int decode_slice_data(SliceCtx& s);

// 8.7: filters the finished picture in place
void deblock_picture(Picture& pic, const std::vector<MbInfo>& mbi, const PPS& pps);

// ---- reconstruction primitives (evc_h264_recon.cpp)
void pred_intra4x4(uint8_t* dst, int stride, int mode, bool left, bool top, bool topright, bool topleft);
void pred_intra8x8(uint8_t* dst, int stride, int mode, bool left, bool top, bool topright, bool topleft);
void pred_intra16x16(uint8_t* dst, int stride, int mode, bool left, bool top, bool topleft);
void pred_intra_chroma(uint8_t* dst, int stride, int mode, bool left, bool top, bool topleft);
void idct4x4_add(uint8_t* dst, int stride, const int32_t* coef /*raster 16*/);
void idct8x8_add(uint8_t* dst, int stride, const int32_t* coef /*raster 64*/);
// luma quarter-sample / chroma eighth-sample prediction of a w x h block at integer position (x,y) + fractional mv
void mc_luma(const Picture& ref, int x, int y, int mvx, int mvy, int w, int h, uint8_t* dst /*stride 16*/);
void mc_chroma(const Picture& ref, int plane, int x, int y, int mvx, int mvy, int w, int h, uint8_t* dst /*stride 8*/);

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip1(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

}  // namespace evc
