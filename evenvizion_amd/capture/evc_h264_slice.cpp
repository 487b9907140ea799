// evc_h264_slice.cpp -- slice data of ITU-T Rec. H.264: the macroblock layer syntax with CABAC parsing (7.3.4, 7.3.5,
// 9.3.2 binarisations, 9.3.3.1 context selection), motion vector prediction (8.4.1), direct prediction (8.4.1.2),
// and the calls that reconstruct each macroblock (8.3 intra, 8.4.2 inter, 8.5 transform).  Frame macroblocks only.
#include <algorithm>
#include <cstdlib>

#include "evc_h264_int.h"

namespace evc {
namespace {

// z-order (decoding order) index of the 4x4 luma block at (bx,by) inside a macroblock (6.4.3)
inline int zorder(int bx, int by) { return ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1); }

struct Nb {
    int ref;  // -2 partition not available, -1 available but list unused / intra, >= 0 reference index
    int mx, my;
};

inline int median3(int a, int b, int c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }

struct MbDecoder {
    SliceCtx& s;
    Cabac c;
    Picture& pic;
    const SPS& sps;
    const PPS& pps;
    const SliceHeader& sh;
    std::vector<MbInfo>& mbi;
    std::vector<int8_t>& ipred;
    std::vector<uint8_t>& direct4;
    Stats& st;
    int mb_w, mb_h, w4, h4;
    int mbx = 0, mby = 0, mbaddr = 0;
    MbInfo* mb = nullptr;
    int qp = 0;
    bool last_dqp_nz = false;

    // coefficients of the current macroblock, dequantised, raster order inside each block
    int32_t cl[16][16];  // luma 4x4 blocks, index by*4+bx
    int32_t cl8[4][64];  // luma 8x8 blocks
    int32_t cc[2][4][16];
    bool nz_l[16], nz_l8[4], nz_c[2][4];

    explicit MbDecoder(SliceCtx& sc)
        : s(sc), pic(*sc.cur), sps(*sc.sps), pps(*sc.pps), sh(*sc.sh), mbi(*sc.mbi), ipred(*sc.ipred), direct4(*sc.direct4),
          st(*sc.stats) {
        mb_w = sps.mb_w;
        mb_h = sps.mb_h;
        w4 = mb_w * 4;
        h4 = mb_h * 4;
    }

    // ------------------------------------------------------------------------------------------ availability (6.4.x)
    bool mb_avail(int x, int y) const {
        if (x < 0 || y < 0 || x >= mb_w || y >= mb_h) return false;
        return mbi[(size_t)y * mb_w + x].slice_id == (uint16_t)s.slice_id;
    }
    const MbInfo* nbmb(int x, int y) const { return mb_avail(x, y) ? &mbi[(size_t)y * mb_w + x] : nullptr; }
    bool intra_nb_usable(int x, int y) const {
        const MbInfo* n = nbmb(x, y);
        if (!n) return false;
        if (pps.constrained_intra_pred && !n->intra) return false;
        return true;
    }

    // ------------------------------------------------------------------------------------------ syntax elements
    int parse_mb_skip() {
        const MbInfo* a = nbmb(mbx - 1, mby);
        const MbInfo* b = nbmb(mbx, mby - 1);
        int inc = (a && !a->skip) + (b && !b->skip);
        return c.decision((sh.type == SLICE_P ? 11 : 24) + inc);
    }

    // 9.3.2.5: mb_type of an intra macroblock; prefix for P/B slices uses ctx_base 17 / 32
    int parse_intra_mb_type(int ctx_base, bool intra_slice) {
        int base = ctx_base;
        if (intra_slice) {
            const MbInfo* a = nbmb(mbx - 1, mby);
            const MbInfo* b = nbmb(mbx, mby - 1);
            int inc = (a && !a->inxn) + (b && !b->inxn);
            if (!c.decision(ctx_base + inc)) return 0;
            base = ctx_base + 2;
        } else {
            if (!c.decision(ctx_base)) return 0;
        }
        if (c.terminate()) return 25;
        int t = 1;
        t += 12 * c.decision(base + 1);
        if (c.decision(base + 2)) t += 4 + 4 * c.decision(base + 2 + (intra_slice ? 1 : 0));
        t += 2 * c.decision(base + 3 + (intra_slice ? 1 : 0));
        t += c.decision(base + 3 + (intra_slice ? 2 : 0));
        return t;
    }

    int parse_mb_type_p() {  // returns 0..3 inter, 5.. intra (mb_type as Table 7-13 + 5)
        if (!c.decision(14)) {
            if (!c.decision(15)) return 3 * c.decision(16);
            return 2 - c.decision(17);
        }
        return 5 + parse_intra_mb_type(17, false);
    }

    int parse_mb_type_b() {
        const MbInfo* a = nbmb(mbx - 1, mby);
        const MbInfo* b = nbmb(mbx, mby - 1);
        int inc = (a && !a->direct16) + (b && !b->direct16);
        if (!c.decision(27 + inc)) return 0;
        if (!c.decision(27 + 3)) return 1 + c.decision(27 + 5);
        int bits = c.decision(27 + 4) << 3;
        bits |= c.decision(27 + 5) << 2;
        bits |= c.decision(27 + 5) << 1;
        bits |= c.decision(27 + 5);
        if (bits < 8) return bits + 3;
        if (bits == 13) return 23 + parse_intra_mb_type(32, false);
        if (bits == 14) return 11;
        if (bits == 15) return 22;
        bits = (bits << 1) | c.decision(27 + 5);
        return bits - 4;
    }

    int parse_sub_mb_type_p() {
        if (c.decision(21)) return 0;
        if (!c.decision(22)) return 1;
        return c.decision(23) ? 2 : 3;
    }

    int parse_sub_mb_type_b() {
        if (!c.decision(36)) return 0;
        if (!c.decision(37)) return 1 + c.decision(39);
        int t = 3;
        if (c.decision(38)) {
            if (c.decision(39)) return 11 + c.decision(39);
            t += 4;
        }
        t += 2 * c.decision(39);
        t += c.decision(39);
        return t;
    }

    int parse_transform8x8() {
        const MbInfo* a = nbmb(mbx - 1, mby);
        const MbInfo* b = nbmb(mbx, mby - 1);
        return c.decision(399 + (a && a->t8x8) + (b && b->t8x8));
    }

    int parse_intra_pred_mode(int pred) {
        if (c.decision(68)) return pred;
        int rem = c.decision(69);
        rem |= c.decision(69) << 1;
        rem |= c.decision(69) << 2;
        return rem < pred ? rem : rem + 1;
    }

    int parse_chroma_pred_mode() {
        const MbInfo* a = nbmb(mbx - 1, mby);
        const MbInfo* b = nbmb(mbx, mby - 1);
        int inc = (a && a->intra && !a->ipcm && a->chroma_pred_mode != 0) + (b && b->intra && !b->ipcm && b->chroma_pred_mode != 0);
        if (!c.decision(64 + inc)) return 0;
        if (!c.decision(64 + 3)) return 1;
        return c.decision(64 + 3) ? 3 : 2;
    }

    int parse_cbp() {
        const MbInfo* a = nbmb(mbx - 1, mby);
        const MbInfo* b = nbmb(mbx, mby - 1);
        // unavailable neighbours count as "all coded" (condTermFlag 0); I_PCM stores 0x2f; skipped macroblocks store 0
        int ca = a ? a->cbp : 0x0f, cb = b ? b->cbp : 0x0f;
        int cbp = 0;
        cbp |= c.decision(73 + !(ca & 2) + 2 * !(cb & 4));
        cbp |= c.decision(73 + !(cbp & 1) + 2 * !(cb & 8)) << 1;
        cbp |= c.decision(73 + !(ca & 8) + 2 * !(cbp & 1)) << 2;
        cbp |= c.decision(73 + !(cbp & 4) + 2 * !(cbp & 2)) << 3;
        int cha = a ? (a->cbp >> 4) : 0, chb = b ? (b->cbp >> 4) : 0;
        int chroma = 0;
        if (c.decision(77 + (cha != 0) + 2 * (chb != 0))) chroma = 1 + c.decision(77 + 4 + (cha == 2) + 2 * (chb == 2));
        return cbp | (chroma << 4);
    }

    int parse_qp_delta() {
        int ctx = last_dqp_nz ? 1 : 0;
        int val = 0;
        while (c.decision(60 + ctx)) {
            ctx = 2 + (ctx >> 1);
            if (++val > 104) fail("mb %d: mb_qp_delta out of range", mbaddr);
        }
        return (val & 1) ? ((val + 1) >> 1) : -((val + 1) >> 1);
    }

    // refIdx of the 4x4 block for the ref_idx context: > 0 and not direct-predicted (9.3.3.1.1.6)
    int ref_gt0(int l, int ax, int ay) const {
        if (ax < 0 || ay < 0 || ax >= w4 || ay >= h4) return 0;
        if (!mb_avail(ax >> 2, ay >> 2)) return 0;
        size_t i = (size_t)ay * w4 + ax;
        if (direct4[i]) return 0;
        return pic.ref[l][i] > 0;
    }
    int parse_ref_idx(int l, int rx, int ry) {
        int ax = mbx * 4 + rx, ay = mby * 4 + ry;
        int ctx = ref_gt0(l, ax - 1, ay) + 2 * ref_gt0(l, ax, ay - 1);
        int ref = 0;
        while (c.decision(54 + ctx)) {
            ++ref;
            ctx = (ctx >> 2) + 4;
            if (ref >= 32) fail("mb %d: ref_idx out of range", mbaddr);
        }
        return ref;
    }

    int mvd_abs(int l, int ax, int ay, int comp) const {
        if (ax < 0 || ay < 0 || ax >= w4 || ay >= h4) return 0;
        if (!mb_avail(ax >> 2, ay >> 2)) return 0;
        return (*s.mvd[l])[((size_t)ay * w4 + ax) * 2 + comp];
    }
    int parse_mvd(int l, int rx, int ry, int comp) {
        int ax = mbx * 4 + rx, ay = mby * 4 + ry;
        int sum = mvd_abs(l, ax - 1, ay, comp) + mvd_abs(l, ax, ay - 1, comp);
        int base = comp ? 47 : 40;
        int inc = sum < 3 ? 0 : (sum > 32 ? 2 : 1);
        if (!c.decision(base + inc)) return 0;
        int mvd = 1, ctx = base + 3;
        while (mvd < 9 && c.decision(ctx)) {
            if (mvd < 4) ++ctx;
            ++mvd;
        }
        if (mvd >= 9) {
            int k = 3;
            while (c.bypass()) {
                mvd += 1 << k;
                if (++k > 24) fail("mb %d: mvd escape too long", mbaddr);
            }
            while (k--) mvd += c.bypass() << k;
        }
        return c.bypass() ? -mvd : mvd;
    }

    // ------------------------------------------------------------------------------------------ residual (7.3.5.3.3)
    // levels come back in scan order; returns the number of non-zero coefficients
    int residual_block(int cat, int cbf_inc, int* lev) {
        static const int cbf_off[5] = {0, 4, 8, 12, 16}, sig_off[5] = {0, 15, 29, 44, 47}, abs_off[5] = {0, 10, 20, 30, 39};
        static const int maxc[6] = {16, 15, 16, 4, 15, 64};
        const int n = maxc[cat];
        for (int i = 0; i < n; ++i) lev[i] = 0;
        if (cat != 5) {
            if (!c.decision(85 + cbf_off[cat] + cbf_inc)) return 0;
        }
        const int sig = cat == 5 ? 402 : 105 + sig_off[cat];
        const int last = cat == 5 ? 417 : 166 + sig_off[cat];
        const int absb = cat == 5 ? 426 : 227 + abs_off[cat];
        int pos[64], np = 0;
        bool ended = false;
        for (int i = 0; i < n - 1; ++i) {
            int is = cat == 5 ? kSigCtx8x8[i] : (cat == 3 ? std::min(i, 2) : i);
            int il = cat == 5 ? kLastCtx8x8[i] : (cat == 3 ? std::min(i, 2) : i);
            if (c.decision(sig + is)) {
                pos[np++] = i;
                if (c.decision(last + il)) {
                    ended = true;
                    break;
                }
            }
        }
        if (!ended) pos[np++] = n - 1;
        int gt1 = 0, eq1 = 0;
        for (int k = np - 1; k >= 0; --k) {
            int ctx = absb + (gt1 ? 0 : std::min(4, 1 + eq1));
            int level;
            if (!c.decision(ctx)) {
                level = 1;
                ++eq1;
            } else {
                int ctx2 = absb + 5 + std::min(4 - (cat == 3 ? 1 : 0), gt1);
                level = 2;
                while (level < 15 && c.decision(ctx2)) ++level;
                if (level >= 15) {
                    int j = 0;
                    while (c.bypass()) {
                        level += 1 << j;
                        if (++j > 24) fail("mb %d: coefficient escape too long", mbaddr);
                    }
                    while (j--) level += c.bypass() << j;
                }
                ++gt1;
            }
            lev[pos[k]] = c.bypass() ? -level : level;
        }
        return np;
    }

    // coded_block_flag context increments (9.3.3.1.1.9)
    int cbf_unavail() const { return mb->intra ? 1 : 0; }
    int cbf_luma_nb(int ax, int ay) const {  // absolute 4x4 coordinates of the neighbouring block
        if (ax < 0 || ay < 0) return cbf_unavail();
        const MbInfo* n = nbmb(ax >> 2, ay >> 2);
        if (!n) return cbf_unavail();
        return (n->cbf_luma >> ((ay & 3) * 4 + (ax & 3))) & 1;
    }
    int cbf_inc_luma(int bx, int by) const {
        int ax = mbx * 4 + bx, ay = mby * 4 + by;
        return cbf_luma_nb(ax - 1, ay) + 2 * cbf_luma_nb(ax, ay - 1);
    }
    int cbf_inc_dc(int bit) const {
        const MbInfo* a = nbmb(mbx - 1, mby);
        const MbInfo* b = nbmb(mbx, mby - 1);
        int ca = a ? ((a->cbf_dc >> bit) & 1) : cbf_unavail();
        int cb = b ? ((b->cbf_dc >> bit) & 1) : cbf_unavail();
        return ca + 2 * cb;
    }
    int cbf_inc_cac(int comp, int cx, int cy) const {
        int ca, cb;
        if (cx > 0) {
            ca = (mb->cbf_cac[comp] >> (cy * 2 + cx - 1)) & 1;
        } else {
            const MbInfo* a = nbmb(mbx - 1, mby);
            ca = a ? ((a->cbf_cac[comp] >> (cy * 2 + 1)) & 1) : cbf_unavail();
        }
        if (cy > 0) {
            cb = (mb->cbf_cac[comp] >> ((cy - 1) * 2 + cx)) & 1;
        } else {
            const MbInfo* b = nbmb(mbx, mby - 1);
            cb = b ? ((b->cbf_cac[comp] >> (2 + cx)) & 1) : cbf_unavail();
        }
        return ca + 2 * cb;
    }

    // 8.5.12.1 scaling of one 4x4 coefficient (flat scaling lists: weightScale = 16)
    static int dq4(int level, int q, int raster) {
        int x = raster & 3, y = raster >> 2;
        int cls = (!(x & 1) && !(y & 1)) ? 0 : ((x & 1) && (y & 1) ? 1 : 2);
        int ls = 16 * kNormAdjust4x4[q % 6][cls];
        if (q >= 24) return (level * ls) * (1 << (q / 6 - 4));
        return (level * ls + (1 << (3 - q / 6))) >> (4 - q / 6);
    }
    static int dq8(int level, int q, int raster) {
        static const uint8_t cls[16] = {0, 3, 4, 3, 3, 1, 5, 1, 4, 5, 2, 5, 3, 1, 5, 1};
        int x = raster & 7, y = raster >> 3;
        int ls = 16 * kNormAdjust8x8[q % 6][cls[((y & 3) << 2) | (x & 3)]];
        if (q >= 36) return (level * ls) * (1 << (q / 6 - 6));
        return (level * ls + (1 << (5 - q / 6))) >> (6 - q / 6);
    }

    void parse_residual(int cbp) {
        int lev[64];
        std::memset(nz_l, 0, sizeof nz_l);
        std::memset(nz_l8, 0, sizeof nz_l8);
        std::memset(nz_c, 0, sizeof nz_c);
        int32_t dcl[16];
        bool have_dc = false;
        if (mb->i16) {
            int n = residual_block(0, cbf_inc_dc(0), lev);
            if (n) {
                mb->cbf_dc |= 1;
                have_dc = true;
                // 8.5.10: inverse scan, 4x4 Hadamard-like transform, scaling
                int cm[16], t[16];
                for (int i = 0; i < 16; ++i) cm[i] = 0;
                for (int k = 0; k < 16; ++k) cm[kZigzag4x4[k]] = lev[k];
                for (int r = 0; r < 4; ++r) {
                    int a = cm[r * 4], b = cm[r * 4 + 1], cc_ = cm[r * 4 + 2], d = cm[r * 4 + 3];
                    t[r * 4 + 0] = a + b + cc_ + d;
                    t[r * 4 + 1] = a + b - cc_ - d;
                    t[r * 4 + 2] = a - b - cc_ + d;
                    t[r * 4 + 3] = a - b + cc_ - d;
                }
                int ls = 16 * kNormAdjust4x4[qp % 6][0];
                for (int col = 0; col < 4; ++col) {
                    int a = t[col], b = t[4 + col], cc_ = t[8 + col], d = t[12 + col];
                    int f[4] = {a + b + cc_ + d, a + b - cc_ - d, a - b - cc_ + d, a - b + cc_ - d};
                    for (int r = 0; r < 4; ++r) {
                        int v = f[r];
                        if (qp >= 36)
                            v = (v * ls) * (1 << (qp / 6 - 6));
                        else
                            v = (v * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
                        dcl[r * 4 + col] = v;  // block at column `col`, row `r`
                    }
                }
            }
        }
        for (int b8 = 0; b8 < 4; ++b8) {
            if (!((cbp >> b8) & 1)) {
                continue;
            }
            int ox = (b8 & 1) * 2, oy = (b8 >> 1) * 2;
            if (mb->t8x8) {
                int n = residual_block(5, 0, lev);
                if (n) {
                    nz_l8[b8] = true;
                    for (int i = 0; i < 64; ++i) cl8[b8][i] = 0;
                    for (int k = 0; k < 64; ++k)
                        if (lev[k]) cl8[b8][kZigzag8x8[k]] = dq8(lev[k], qp, kZigzag8x8[k]);
                    for (int j = 0; j < 4; ++j) mb->cbf_luma |= 1u << ((oy + (j >> 1)) * 4 + ox + (j & 1));
                }
            } else {
                for (int j = 0; j < 4; ++j) {
                    int bx = ox + (j & 1), by = oy + (j >> 1), bi = by * 4 + bx;
                    int n;
                    if (mb->i16) {
                        n = residual_block(1, cbf_inc_luma(bx, by), lev);
                        if (n) {
                            for (int i = 0; i < 16; ++i) cl[bi][i] = 0;
                            for (int k = 0; k < 15; ++k)
                                if (lev[k]) cl[bi][kZigzag4x4[k + 1]] = dq4(lev[k], qp, kZigzag4x4[k + 1]);
                        }
                    } else {
                        n = residual_block(2, cbf_inc_luma(bx, by), lev);
                        if (n) {
                            for (int i = 0; i < 16; ++i) cl[bi][i] = 0;
                            for (int k = 0; k < 16; ++k)
                                if (lev[k]) cl[bi][kZigzag4x4[k]] = dq4(lev[k], qp, kZigzag4x4[k]);
                        }
                    }
                    if (n) {
                        nz_l[bi] = true;
                        mb->cbf_luma |= 1u << bi;
                    }
                }
            }
        }
        if (mb->i16 && have_dc) {
            for (int bi = 0; bi < 16; ++bi) {
                if (!nz_l[bi]) {
                    for (int i = 0; i < 16; ++i) cl[bi][i] = 0;
                    nz_l[bi] = dcl[bi] != 0;
                }
                cl[bi][0] = dcl[bi];
            }
        }
        const int chroma = cbp >> 4;
        if (chroma) {
            int32_t dcc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            for (int comp = 0; comp < 2; ++comp) {
                int n = residual_block(3, cbf_inc_dc(1 + comp), lev);
                if (n) {
                    mb->cbf_dc |= 2 << comp;
                    int q = mb->qpc[comp];
                    int a = lev[0], b = lev[1], cc_ = lev[2], d = lev[3];
                    int f[4] = {a + b + cc_ + d, a - b + cc_ - d, a + b - cc_ - d, a - b - cc_ + d};
                    int ls = 16 * kNormAdjust4x4[q % 6][0];
                    for (int i = 0; i < 4; ++i) dcc[comp][i] = ((f[i] * ls) * (1 << (q / 6))) >> 5;
                }
            }
            for (int comp = 0; comp < 2; ++comp)
                for (int j = 0; j < 4; ++j) {
                    for (int i = 0; i < 16; ++i) cc[comp][j][i] = 0;
                    bool any = false;
                    if (chroma == 2) {
                        int n = residual_block(4, cbf_inc_cac(comp, j & 1, j >> 1), lev);
                        if (n) {
                            mb->cbf_cac[comp] |= 1u << j;
                            int q = mb->qpc[comp];
                            for (int k = 0; k < 15; ++k)
                                if (lev[k]) cc[comp][j][kZigzag4x4[k + 1]] = dq4(lev[k], q, kZigzag4x4[k + 1]);
                            any = true;
                        }
                    }
                    cc[comp][j][0] = dcc[comp][j];
                    nz_c[comp][j] = any || dcc[comp][j] != 0;
                }
        }
    }

    // ------------------------------------------------------------------------------------------ motion data
    void fill_motion(int l, int rx, int ry, int w, int h, int ref, int mx, int my) {
        int id = -1;
        if (ref >= 0) {
            if (ref >= (int)s.list[l].size() || !s.list[l][ref]) fail("mb %d: reference index %d of list %d has no picture", mbaddr, ref, l);
            id = s.list[l][ref]->id;
        }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                size_t i = (size_t)(mby * 4 + ry + y) * w4 + mbx * 4 + rx + x;
                pic.ref[l][i] = (int8_t)ref;
                pic.ref_id[l][i] = id;
                pic.mv[l][i * 2] = (int16_t)mx;
                pic.mv[l][i * 2 + 1] = (int16_t)my;
            }
    }
    void fill_mv(int l, int rx, int ry, int w, int h, int mx, int my) {
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                size_t i = (size_t)(mby * 4 + ry + y) * w4 + mbx * 4 + rx + x;
                pic.mv[l][i * 2] = (int16_t)mx;
                pic.mv[l][i * 2 + 1] = (int16_t)my;
            }
    }
    void fill_mvd(int l, int rx, int ry, int w, int h, int dx, int dy) {
        int ax = std::min(std::abs(dx), 32767), ay = std::min(std::abs(dy), 32767);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                size_t i = (size_t)(mby * 4 + ry + y) * w4 + mbx * 4 + rx + x;
                (*s.mvd[l])[i * 2] = (int16_t)ax;
                (*s.mvd[l])[i * 2 + 1] = (int16_t)ay;
            }
    }
    void fill_direct(int rx, int ry, int w, int h, int v) {
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) direct4[(size_t)(mby * 4 + ry + y) * w4 + mbx * 4 + rx + x] = (uint8_t)v;
    }

    Nb nb(int l, int ax, int ay) const {
        if (ax < 0 || ay < 0 || ax >= w4 || ay >= h4) return {-2, 0, 0};
        if (!mb_avail(ax >> 2, ay >> 2)) return {-2, 0, 0};
        size_t i = (size_t)ay * w4 + ax;
        int r = pic.ref[l][i];
        if (r < 0) return {-1, 0, 0};
        return {r, pic.mv[l][i * 2], pic.mv[l][i * 2 + 1]};
    }
    // neighbour C of the partition at (rx,ry) (relative 4x4 units) of width w, falling back to D (6.4.11.7)
    Nb nbC(int l, int rx, int ry, int w) const {
        int cx = rx + w, cy = ry - 1;
        bool try_c;
        if (cy < 0)
            try_c = true;  // row above: macroblock B or C decides
        else if (cx >= 4)
            try_c = false;  // right-hand macroblock: not decoded yet
        else
            try_c = zorder(cx, cy) < zorder(rx, ry);
        if (try_c) {
            Nb n = nb(l, mbx * 4 + cx, mby * 4 + cy);
            if (n.ref != -2) return n;
        }
        return nb(l, mbx * 4 + rx - 1, mby * 4 + ry - 1);
    }

    // 8.4.1.3: motion vector prediction; shape 0 = median, 1/2 = upper/lower 16x8, 3/4 = left/right 8x16
    void predict_mv(int l, int rx, int ry, int w, int ref, int shape, int& px, int& py) const {
        Nb a = nb(l, mbx * 4 + rx - 1, mby * 4 + ry);
        Nb b = nb(l, mbx * 4 + rx, mby * 4 + ry - 1);
        Nb cn = nbC(l, rx, ry, w);
        if (shape == 1 && b.ref == ref) {
            px = b.mx, py = b.my;
            return;
        }
        if (shape == 2 && a.ref == ref) {
            px = a.mx, py = a.my;
            return;
        }
        if (shape == 3 && a.ref == ref) {
            px = a.mx, py = a.my;
            return;
        }
        if (shape == 4 && cn.ref == ref) {
            px = cn.mx, py = cn.my;
            return;
        }
        median_mv(a, b, cn, ref, px, py);
    }
    static void median_mv(Nb a, Nb b, Nb cn, int ref, int& px, int& py) {
        if (b.ref == -2 && cn.ref == -2 && a.ref != -2) {
            px = a.mx, py = a.my;
            return;
        }
        int match = (a.ref == ref) + (b.ref == ref) + (cn.ref == ref);
        if (match == 1) {
            const Nb& m = a.ref == ref ? a : (b.ref == ref ? b : cn);
            px = m.mx, py = m.my;
            return;
        }
        px = median3(a.mx, b.mx, cn.mx);
        py = median3(a.my, b.my, cn.my);
    }

    // 8.4.1.1: P_Skip
    void pskip_motion() {
        Nb a = nb(0, mbx * 4 - 1, mby * 4);
        Nb b = nb(0, mbx * 4, mby * 4 - 1);
        int px = 0, py = 0;
        if (a.ref == -2 || b.ref == -2 || (a.ref == 0 && a.mx == 0 && a.my == 0) || (b.ref == 0 && b.mx == 0 && b.my == 0)) {
            px = py = 0;
        } else {
            Nb cn = nbC(0, 0, 0, 4);
            median_mv(a, b, cn, 0, px, py);
        }
        fill_motion(0, 0, 0, 4, 4, 0, px, py);
        fill_motion(1, 0, 0, 4, 4, -1, 0, 0);
    }

    // 8.4.1.2: direct prediction of the 8x8 quadrants in `mask` (bit per quadrant); writes motion + direct flags
    void direct_motion(int mask) {
        const Picture* col = s.list[1].empty() ? nullptr : s.list[1][0];
        if (!col) fail("mb %d: direct prediction without RefPicList1[0]", mbaddr);
        const bool inf8 = sps.direct_8x8_inference;
        if (sh.direct_spatial) {
            ++st.spatial_direct_mbs;
            int ref[2], pmx[2] = {0, 0}, pmy[2] = {0, 0};
            for (int l = 0; l < 2; ++l) {
                Nb a = nb(l, mbx * 4 - 1, mby * 4);
                Nb b = nb(l, mbx * 4, mby * 4 - 1);
                Nb cn = nbC(l, 0, 0, 4);
                auto minpos = [](int x, int y) { return (x >= 0 && y >= 0) ? std::min(x, y) : std::max(x, y); };
                ref[l] = minpos(a.ref, minpos(b.ref, cn.ref));
                if (ref[l] >= 0) median_mv(a, b, cn, ref[l], pmx[l], pmy[l]);
                else ref[l] = -1;
            }
            bool zero_pred = false;
            if (ref[0] < 0 && ref[1] < 0) {
                ref[0] = ref[1] = 0;
                zero_pred = true;
            }
            for (int q = 0; q < 4; ++q) {
                if (!((mask >> q) & 1)) continue;
                int qx = (q & 1) * 2, qy = (q >> 1) * 2;
                int step = inf8 ? 2 : 1;
                for (int sy = 0; sy < 2; sy += step)
                    for (int sx = 0; sx < 2; sx += step) {
                        int bx = qx + sx, by = qy + sy;
                        // co-located 4x4 block: the corner block of the quadrant with direct_8x8_inference (8.4.1.2.1)
                        int cxb = inf8 ? (q & 1) * 3 : bx, cyb = inf8 ? (q >> 1) * 3 : by;
                        size_t ci = (size_t)(mby * 4 + cyb) * w4 + mbx * 4 + cxb;
                        bool col_zero = false;
                        if (!zero_pred && !col->is_long) {
                            int rc = col->ref[0][ci];
                            int lc = 0;
                            if (rc < 0) {
                                rc = col->ref[1][ci];
                                lc = 1;
                            }
                            if (rc == 0) {
                                int mx = col->mv[lc][ci * 2], my = col->mv[lc][ci * 2 + 1];
                                col_zero = mx >= -1 && mx <= 1 && my >= -1 && my <= 1;
                            }
                        }
                        for (int l = 0; l < 2; ++l) {
                            int mx = 0, my = 0;
                            if (!zero_pred && ref[l] >= 0 && !(ref[l] == 0 && col_zero)) {
                                mx = pmx[l];
                                my = pmy[l];
                            }
                            fill_motion(l, bx, by, step, step, ref[l], ref[l] >= 0 ? mx : 0, ref[l] >= 0 ? my : 0);
                        }
                    }
                fill_direct(qx, qy, 2, 2, 1);
            }
        } else {
            ++st.temporal_direct_mbs;
            for (int q = 0; q < 4; ++q) {
                if (!((mask >> q) & 1)) continue;
                int qx = (q & 1) * 2, qy = (q >> 1) * 2;
                int step = inf8 ? 2 : 1;
                for (int sy = 0; sy < 2; sy += step)
                    for (int sx = 0; sx < 2; sx += step) {
                        int bx = qx + sx, by = qy + sy;
                        int cxb = inf8 ? (q & 1) * 3 : bx, cyb = inf8 ? (q >> 1) * 3 : by;
                        size_t ci = (size_t)(mby * 4 + cyb) * w4 + mbx * 4 + cxb;
                        int rc = col->ref[0][ci], lc = 0;
                        if (rc < 0) {
                            rc = col->ref[1][ci];
                            lc = 1;
                        }
                        int r0 = 0, m0x = 0, m0y = 0, m1x = 0, m1y = 0;
                        if (rc >= 0) {
                            int want = col->ref_id[lc][ci];
                            r0 = -1;
                            for (int i = 0; i < (int)s.list[0].size(); ++i)
                                if (s.list[0][i] && s.list[0][i]->id == want) {
                                    r0 = i;
                                    break;
                                }
                            if (r0 < 0) fail("mb %d: temporal direct: the co-located block's reference picture is not in RefPicList0", mbaddr);
                            int cmx = col->mv[lc][ci * 2], cmy = col->mv[lc][ci * 2 + 1];
                            const Picture* p0 = s.list[0][r0];
                            if (p0->is_long || col->poc == p0->poc) {
                                m0x = cmx;
                                m0y = cmy;
                            } else {
                                int dsf = s.dist_scale[r0];
                                m0x = (dsf * cmx + 128) >> 8;
                                m0y = (dsf * cmy + 128) >> 8;
                                m1x = m0x - cmx;
                                m1y = m0y - cmy;
                            }
                        }
                        fill_motion(0, bx, by, step, step, r0, m0x, m0y);
                        fill_motion(1, bx, by, step, step, 0, m1x, m1y);
                    }
                fill_direct(qx, qy, 2, 2, 1);
            }
        }
    }

    // ------------------------------------------------------------------------------------------ inter prediction (8.4.2)
    void inter_pred_block(int rx, int ry, int w, int h /*4x4 units*/) {
        size_t i = (size_t)(mby * 4 + ry) * w4 + mbx * 4 + rx;
        int ref0 = pic.ref[0][i], ref1 = pic.ref[1][i];
        uint8_t py[2][16 * 16], pcb[2][8 * 8], pcr[2][8 * 8];
        int pw = w * 4, ph = h * 4, cw = w * 2, chh = h * 2;
        int X = mbx * 16 + rx * 4, Y = mby * 16 + ry * 4;
        const int refs[2] = {ref0, ref1};
        for (int l = 0; l < 2; ++l) {
            if (refs[l] < 0) continue;
            const Picture* rp = s.list[l][refs[l]];
            if (!rp) fail("mb %d: missing reference picture", mbaddr);
            int mx = pic.mv[l][i * 2], my = pic.mv[l][i * 2 + 1];
            mc_luma(*rp, X, Y, mx, my, pw, ph, py[l]);
            mc_chroma(*rp, 0, X / 2, Y / 2, mx, my, cw, chh, pcb[l]);
            mc_chroma(*rp, 1, X / 2, Y / 2, mx, my, cw, chh, pcr[l]);
            if (refs[l] > st.max_ref_idx) st.max_ref_idx = refs[l];
        }
        uint8_t* dy = &pic.Y[(size_t)Y * pic.stride + X];
        uint8_t* dcb = &pic.Cb[(size_t)(Y / 2) * pic.cstride + X / 2];
        uint8_t* dcr = &pic.Cr[(size_t)(Y / 2) * pic.cstride + X / 2];
        const int mode = s.wt.mode;
        if (ref0 >= 0 && ref1 >= 0) {
            ++st.bipred_blocks;
            int w0 = 32, w1 = 32, o0 = 0, o1 = 0, lwd = 5;
            int cw0[2] = {32, 32}, cw1[2] = {32, 32}, co0[2] = {0, 0}, co1[2] = {0, 0}, clwd = 5;
            bool weighted = false;
            if (mode == 2) {
                w0 = s.wt.implicit_w0[ref0][ref1];
                w1 = 64 - w0;
                cw0[0] = cw0[1] = w0;
                cw1[0] = cw1[1] = w1;
                weighted = w0 != 32;
                if (weighted) ++st.implicit_wp_blocks;
            } else if (mode == 1) {
                weighted = true;
                ++st.explicit_wp_blocks;
                lwd = sh.luma_log2_denom;
                clwd = sh.chroma_log2_denom;
                w0 = sh.luma_w[0][ref0], w1 = sh.luma_w[1][ref1], o0 = sh.luma_o[0][ref0], o1 = sh.luma_o[1][ref1];
                for (int k = 0; k < 2; ++k) {
                    cw0[k] = sh.chroma_w[0][ref0][k], cw1[k] = sh.chroma_w[1][ref1][k];
                    co0[k] = sh.chroma_o[0][ref0][k], co1[k] = sh.chroma_o[1][ref1][k];
                }
            }
            for (int y = 0; y < ph; ++y)
                for (int x = 0; x < pw; ++x) {
                    int a = py[0][y * 16 + x], b = py[1][y * 16 + x];
                    dy[(size_t)y * pic.stride + x] =
                        weighted ? (uint8_t)clip1(((a * w0 + b * w1 + (1 << lwd)) >> (lwd + 1)) + ((o0 + o1 + 1) >> 1)) : (uint8_t)((a + b + 1) >> 1);
                }
            for (int y = 0; y < chh; ++y)
                for (int x = 0; x < cw; ++x) {
                    int a = pcb[0][y * 8 + x], b = pcb[1][y * 8 + x];
                    dcb[(size_t)y * pic.cstride + x] =
                        weighted ? (uint8_t)clip1(((a * cw0[0] + b * cw1[0] + (1 << clwd)) >> (clwd + 1)) + ((co0[0] + co1[0] + 1) >> 1)) : (uint8_t)((a + b + 1) >> 1);
                    a = pcr[0][y * 8 + x], b = pcr[1][y * 8 + x];
                    dcr[(size_t)y * pic.cstride + x] =
                        weighted ? (uint8_t)clip1(((a * cw0[1] + b * cw1[1] + (1 << clwd)) >> (clwd + 1)) + ((co0[1] + co1[1] + 1) >> 1)) : (uint8_t)((a + b + 1) >> 1);
                }
        } else {
            int l = ref0 >= 0 ? 0 : 1;
            int r = refs[l];
            if (r < 0) fail("mb %d: inter block without a reference", mbaddr);
            bool weighted = mode == 1;
            int wl = 1, ol = 0, lwd = 0, wc[2] = {1, 1}, oc[2] = {0, 0}, clwd = 0;
            if (weighted) {
                lwd = sh.luma_log2_denom;
                clwd = sh.chroma_log2_denom;
                wl = sh.luma_w[l][r];
                ol = sh.luma_o[l][r];
                for (int k = 0; k < 2; ++k) wc[k] = sh.chroma_w[l][r][k], oc[k] = sh.chroma_o[l][r][k];
                ++st.explicit_wp_blocks;
            }
            auto wp = [](int p, int w_, int o, int d) { return d >= 1 ? clip1(((p * w_ + (1 << (d - 1))) >> d) + o) : clip1(p * w_ + o); };
            for (int y = 0; y < ph; ++y)
                for (int x = 0; x < pw; ++x) {
                    int a = py[l][y * 16 + x];
                    dy[(size_t)y * pic.stride + x] = weighted ? (uint8_t)wp(a, wl, ol, lwd) : (uint8_t)a;
                }
            for (int y = 0; y < chh; ++y)
                for (int x = 0; x < cw; ++x) {
                    int a = pcb[l][y * 8 + x], b = pcr[l][y * 8 + x];
                    dcb[(size_t)y * pic.cstride + x] = weighted ? (uint8_t)wp(a, wc[0], oc[0], clwd) : (uint8_t)a;
                    dcr[(size_t)y * pic.cstride + x] = weighted ? (uint8_t)wp(b, wc[1], oc[1], clwd) : (uint8_t)b;
                }
        }
    }

    void inter_pred_mb() {
        // uniform motion regions give the same samples whether predicted whole or in pieces, so the macroblock is cut
        // into 8x8 quadrants, and a quadrant into 4x4 blocks only when its motion differs inside
        for (int q = 0; q < 4; ++q) {
            int qx = (q & 1) * 2, qy = (q >> 1) * 2;
            size_t i0 = (size_t)(mby * 4 + qy) * w4 + mbx * 4 + qx;
            bool uniform = true;
            for (int k = 1; k < 4 && uniform; ++k) {
                size_t i = i0 + (size_t)(k >> 1) * w4 + (k & 1);
                for (int l = 0; l < 2; ++l)
                    if (pic.ref[l][i] != pic.ref[l][i0] || pic.mv[l][i * 2] != pic.mv[l][i0 * 2] || pic.mv[l][i * 2 + 1] != pic.mv[l][i0 * 2 + 1])
                        uniform = false;
            }
            if (uniform) {
                inter_pred_block(qx, qy, 2, 2);
            } else {
                ++st.sub8x8;
                for (int k = 0; k < 4; ++k) inter_pred_block(qx + (k & 1), qy + (k >> 1), 1, 1);
            }
        }
    }

    // ------------------------------------------------------------------------------------------ residual add
    void add_residual_luma() {
        uint8_t* base = &pic.Y[(size_t)mby * 16 * pic.stride + mbx * 16];
        if (mb->t8x8) {
            for (int b8 = 0; b8 < 4; ++b8)
                if (nz_l8[b8]) idct8x8_add(base + (size_t)(b8 >> 1) * 8 * pic.stride + (b8 & 1) * 8, pic.stride, cl8[b8]);
        } else {
            for (int bi = 0; bi < 16; ++bi)
                if (nz_l[bi]) idct4x4_add(base + (size_t)(bi >> 2) * 4 * pic.stride + (bi & 3) * 4, pic.stride, cl[bi]);
        }
    }
    void add_residual_chroma() {
        for (int comp = 0; comp < 2; ++comp) {
            std::vector<uint8_t>& pl = comp ? pic.Cr : pic.Cb;
            uint8_t* base = &pl[(size_t)mby * 8 * pic.cstride + mbx * 8];
            for (int j = 0; j < 4; ++j)
                if (nz_c[comp][j]) idct4x4_add(base + (size_t)(j >> 1) * 4 * pic.cstride + (j & 1) * 4, pic.cstride, cc[comp][j]);
        }
    }

    // ------------------------------------------------------------------------------------------ intra
    int ipred_nb(int ax, int ay) const {
        if (ax < 0 || ay < 0) return -2;
        const MbInfo* n = nbmb(ax >> 2, ay >> 2);
        if (!n) return -2;
        if (!n->intra && pps.constrained_intra_pred) return -2;
        int m = ipred[(size_t)ay * w4 + ax];
        return m < 0 ? 2 : m;
    }
    int predict_ipred(int bx, int by) const {
        int a = ipred_nb(mbx * 4 + bx - 1, mby * 4 + by), b = ipred_nb(mbx * 4 + bx, mby * 4 + by - 1);
        if (a == -2 || b == -2) return 2;
        return std::min(a, b);
    }
    void set_ipred(int bx, int by, int w, int m) {
        for (int y = 0; y < w; ++y)
            for (int x = 0; x < w; ++x) ipred[(size_t)(mby * 4 + by + y) * w4 + mbx * 4 + bx + x] = (int8_t)m;
    }

    void recon_intra_nxn(const int* modes) {
        bool A = intra_nb_usable(mbx - 1, mby), B = intra_nb_usable(mbx, mby - 1), C = intra_nb_usable(mbx + 1, mby - 1),
             D = intra_nb_usable(mbx - 1, mby - 1);
        uint8_t* base = &pic.Y[(size_t)mby * 16 * pic.stride + mbx * 16];
        if (mb->t8x8) {
            for (int b8 = 0; b8 < 4; ++b8) {
                int bx = b8 & 1, by = b8 >> 1;
                bool left = bx ? true : A, top = by ? true : B;
                bool tl = (bx && by) ? true : (bx ? B : (by ? A : D));
                bool tr = by == 0 ? (bx == 0 ? B : C) : (bx == 0);
                uint8_t* d = base + (size_t)by * 8 * pic.stride + bx * 8;
                pred_intra8x8(d, pic.stride, modes[b8], left, top, tr, tl);
                if (nz_l8[b8]) idct8x8_add(d, pic.stride, cl8[b8]);
            }
        } else {
            for (int z = 0; z < 16; ++z) {
                int bx = ((z >> 2) & 1) * 2 + (z & 1), by = (z >> 3) * 2 + ((z >> 1) & 1);
                bool left = bx ? true : A, top = by ? true : B;
                bool tl = (bx && by) ? true : (bx ? B : (by ? A : D));
                bool tr;
                if (by == 0)
                    tr = bx < 3 ? B : C;
                else if (bx == 3)
                    tr = false;
                else
                    tr = zorder(bx + 1, by - 1) < z;
                uint8_t* d = base + (size_t)by * 4 * pic.stride + bx * 4;
                pred_intra4x4(d, pic.stride, modes[by * 4 + bx], left, top, tr, tl);
                if (nz_l[by * 4 + bx]) idct4x4_add(d, pic.stride, cl[by * 4 + bx]);
            }
        }
    }

    void recon_intra_chroma() {
        bool A = intra_nb_usable(mbx - 1, mby), B = intra_nb_usable(mbx, mby - 1), D = intra_nb_usable(mbx - 1, mby - 1);
        pred_intra_chroma(&pic.Cb[(size_t)mby * 8 * pic.cstride + mbx * 8], pic.cstride, mb->chroma_pred_mode, A, B, D);
        pred_intra_chroma(&pic.Cr[(size_t)mby * 8 * pic.cstride + mbx * 8], pic.cstride, mb->chroma_pred_mode, A, B, D);
        add_residual_chroma();
    }

    // ------------------------------------------------------------------------------------------ macroblock layer (7.3.5)
    void set_qp(int q) {
        qp = q;
        mb->qp = (int8_t)q;
        for (int k = 0; k < 2; ++k) mb->qpc[k] = (int8_t)kChromaQp[clip3(0, 51, q + pps.chroma_qp_offset[k])];
    }

    void begin_mb() {
        mbx = mbaddr % mb_w;
        mby = mbaddr / mb_w;
        mb = &mbi[mbaddr];
        if (mb->slice_id != 0xFFFF) fail("mb %d decoded twice in one picture", mbaddr);
        std::memset(mb, 0, sizeof *mb);
        mb->slice_id = (uint16_t)s.slice_id;
        mb->disable_deblock = (int8_t)sh.disable_deblock;
        mb->alpha_off = (int8_t)sh.alpha_off;
        mb->beta_off = (int8_t)sh.beta_off;
        set_qp(qp);
        for (int l = 0; l < 2; ++l) fill_mvd(l, 0, 0, 4, 4, 0, 0);
        fill_direct(0, 0, 4, 4, 0);
        set_ipred(0, 0, 4, -1);
        pic.mb_intra[mbaddr] = 0;
        ++st.mbs;
    }

    void decode_skip() {
        mb->skip = 1;
        last_dqp_nz = false;
        std::memset(nz_l, 0, sizeof nz_l);
        std::memset(nz_l8, 0, sizeof nz_l8);
        std::memset(nz_c, 0, sizeof nz_c);
        if (sh.type == SLICE_P) {
            ++st.p_skip;
            pskip_motion();
        } else {
            ++st.b_skip;
            mb->direct16 = 1;
            direct_motion(15);
        }
        inter_pred_mb();
    }

    void decode_ipcm() {
        ++st.ipcm;
        mb->intra = mb->ipcm = 1;
        pic.mb_intra[mbaddr] = 1;
        // 9.3.1.2: the bin that announced I_PCM ended the arithmetic codeword (its last bit has been read);
        // pcm_alignment_zero_bits follow up to the byte boundary, then 384 raw bytes, then a fresh arithmetic codeword
        c.bits_left = 0;
        if (c.end - c.p < 384) fail("mb %d: I_PCM samples run past the slice", mbaddr);
        uint8_t* y = &pic.Y[(size_t)mby * 16 * pic.stride + mbx * 16];
        for (int r = 0; r < 16; ++r, y += pic.stride) {
            std::memcpy(y, c.p, 16);
            c.p += 16;
        }
        for (int comp = 0; comp < 2; ++comp) {
            uint8_t* d = &(comp ? pic.Cr : pic.Cb)[(size_t)mby * 8 * pic.cstride + mbx * 8];
            for (int r = 0; r < 8; ++r, d += pic.cstride) {
                std::memcpy(d, c.p, 8);
                c.p += 8;
            }
        }
        c.init_engine(c.p, c.end);
        mb->cbp = 0x2f;
        mb->cbf_luma = 0xFFFF;
        mb->cbf_dc = 7;
        mb->cbf_cac[0] = mb->cbf_cac[1] = 0xF;
        mb->qp = 0;
        mb->qpc[0] = mb->qpc[1] = (int8_t)kChromaQp[clip3(0, 51, pps.chroma_qp_offset[0])];
        mb->qpc[1] = (int8_t)kChromaQp[clip3(0, 51, pps.chroma_qp_offset[1])];
        last_dqp_nz = false;
        for (int l = 0; l < 2; ++l) fill_motion(l, 0, 0, 4, 4, -1, 0, 0);
    }

    void decode_intra(int itype) {
        mb->intra = 1;
        pic.mb_intra[mbaddr] = 1;
        for (int l = 0; l < 2; ++l) fill_motion(l, 0, 0, 4, 4, -1, 0, 0);
        int modes[16];
        int cbp;
        if (itype == 0) {
            mb->inxn = 1;
            if (pps.transform_8x8_mode) mb->t8x8 = (uint8_t)parse_transform8x8();
            if (mb->t8x8) {
                ++st.i8;
                for (int b8 = 0; b8 < 4; ++b8) {
                    int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2;
                    modes[b8] = parse_intra_pred_mode(predict_ipred(bx, by));
                    set_ipred(bx, by, 2, modes[b8]);
                }
            } else {
                ++st.i4;
                for (int z = 0; z < 16; ++z) {
                    int bx = ((z >> 2) & 1) * 2 + (z & 1), by = (z >> 3) * 2 + ((z >> 1) & 1);
                    int m = parse_intra_pred_mode(predict_ipred(bx, by));
                    modes[by * 4 + bx] = m;
                    set_ipred(bx, by, 1, m);
                }
            }
            mb->chroma_pred_mode = (uint8_t)parse_chroma_pred_mode();
            cbp = parse_cbp();
        } else {
            ++st.i16;
            mb->i16 = 1;
            int t = itype - 1;
            modes[0] = t & 3;
            cbp = (((t >> 2) % 3) << 4) | (t >= 12 ? 15 : 0);
            mb->chroma_pred_mode = (uint8_t)parse_chroma_pred_mode();
        }
        mb->cbp = (uint8_t)cbp;
        if (cbp || mb->i16) {
            int d = parse_qp_delta();
            last_dqp_nz = d != 0;
            if (d < -26 || d > 25) fail("mb %d: mb_qp_delta %d", mbaddr, d);
            set_qp((qp + d + 52) % 52);
            parse_residual(cbp);
        } else {
            last_dqp_nz = false;
            std::memset(nz_l, 0, sizeof nz_l);
            std::memset(nz_l8, 0, sizeof nz_l8);
            std::memset(nz_c, 0, sizeof nz_c);
        }
        if (mb->i16) {
            bool A = intra_nb_usable(mbx - 1, mby), B = intra_nb_usable(mbx, mby - 1), D = intra_nb_usable(mbx - 1, mby - 1);
            pred_intra16x16(&pic.Y[(size_t)mby * 16 * pic.stride + mbx * 16], pic.stride, modes[0], A, B, D);
            add_residual_luma();
        } else {
            recon_intra_nxn(modes);
        }
        recon_intra_chroma();
    }

    struct Part {
        int x, y, w, h, pred /*1 L0, 2 L1, 3 Bi*/, shape;
    };

    void decode_inter(int t) {
        ++st.inter;
        const bool isB = sh.type == SLICE_B;
        bool no_sub_lt8 = true;
        bool direct16 = false;
        Part parts[16];
        int np = 0;
        int sub_direct_mask = 0;
        if (isB && t == 0) {
            direct16 = true;
            mb->direct16 = 1;
            ++st.b_direct;
            direct_motion(15);
        } else if ((!isB && t <= 2) || (isB && t <= 21)) {
            int shape, p0, p1;
            if (!isB) {
                shape = t;
                p0 = p1 = 1;
            } else if (t <= 3) {
                shape = 0;
                p0 = p1 = t;
            } else {
                static const int pr[9][2] = {{1, 1}, {2, 2}, {1, 2}, {2, 1}, {1, 3}, {2, 3}, {3, 1}, {3, 2}, {3, 3}};
                shape = ((t - 4) & 1) ? 2 : 1;
                p0 = pr[(t - 4) >> 1][0];
                p1 = pr[(t - 4) >> 1][1];
            }
            if (shape == 0) {
                parts[np++] = {0, 0, 4, 4, p0, 0};
            } else if (shape == 1) {
                parts[np++] = {0, 0, 4, 2, p0, 1};
                parts[np++] = {0, 2, 4, 2, p1, 2};
            } else {
                parts[np++] = {0, 0, 2, 4, p0, 3};
                parts[np++] = {2, 0, 2, 4, p1, 4};
            }
            for (int l = 0; l < 2; ++l) fill_motion(l, 0, 0, 4, 4, -1, 0, 0);
            for (int l = 0; l < 2; ++l)
                for (int i = 0; i < np; ++i)
                    if (parts[i].pred & (1 << l)) {
                        int r = sh.num_ref_idx[l] > 1 ? parse_ref_idx(l, parts[i].x, parts[i].y) : 0;
                        if (r >= sh.num_ref_idx[l]) fail("mb %d: ref_idx %d >= num_ref_idx_active", mbaddr, r);
                        fill_motion(l, parts[i].x, parts[i].y, parts[i].w, parts[i].h, r, 0, 0);
                    }
            for (int l = 0; l < 2; ++l)
                for (int i = 0; i < np; ++i)
                    if (parts[i].pred & (1 << l)) {
                        const Part& p = parts[i];
                        int dx = parse_mvd(l, p.x, p.y, 0), dy = parse_mvd(l, p.x, p.y, 1);
                        int r = pic.ref[l][(size_t)(mby * 4 + p.y) * w4 + mbx * 4 + p.x];
                        int px, py;
                        predict_mv(l, p.x, p.y, p.w, r, p.shape, px, py);
                        fill_mv(l, p.x, p.y, p.w, p.h, px + dx, py + dy);
                        fill_mvd(l, p.x, p.y, p.w, p.h, dx, dy);
                    }
        } else {
            // P_8x8 / B_8x8
            int sub[4], spred[4], sshape[4];
            static const int bsub[13][2] = {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {1, 1}, {1, 2}, {2, 1}, {2, 2}, {3, 1}, {3, 2}, {1, 3}, {2, 3}, {3, 3}};
            for (int q = 0; q < 4; ++q) {
                if (isB) {
                    sub[q] = parse_sub_mb_type_b();
                    spred[q] = bsub[sub[q]][0];
                    sshape[q] = bsub[sub[q]][1];
                    if (sub[q] == 0) {
                        sub_direct_mask |= 1 << q;
                        if (!sps.direct_8x8_inference) no_sub_lt8 = false;
                    } else if (sshape[q] != 0) {
                        no_sub_lt8 = false;
                    }
                } else {
                    sub[q] = parse_sub_mb_type_p();
                    spred[q] = 1;
                    sshape[q] = sub[q];
                    if (sshape[q] != 0) no_sub_lt8 = false;
                }
            }
            for (int l = 0; l < 2; ++l) fill_motion(l, 0, 0, 4, 4, -1, 0, 0);
            if (sub_direct_mask) direct_motion(sub_direct_mask);
            for (int l = 0; l < 2; ++l)
                for (int q = 0; q < 4; ++q)
                    if (!((sub_direct_mask >> q) & 1) && (spred[q] & (1 << l))) {
                        int qx = (q & 1) * 2, qy = (q >> 1) * 2;
                        int r = sh.num_ref_idx[l] > 1 ? parse_ref_idx(l, qx, qy) : 0;
                        if (r >= sh.num_ref_idx[l]) fail("mb %d: ref_idx %d >= num_ref_idx_active", mbaddr, r);
                        fill_motion(l, qx, qy, 2, 2, r, 0, 0);
                    }
            for (int l = 0; l < 2; ++l)
                for (int q = 0; q < 4; ++q)
                    if (!((sub_direct_mask >> q) & 1) && (spred[q] & (1 << l))) {
                        int qx = (q & 1) * 2, qy = (q >> 1) * 2;
                        int r = pic.ref[l][(size_t)(mby * 4 + qy) * w4 + mbx * 4 + qx];
                        int nsub = sshape[q] == 0 ? 1 : (sshape[q] == 3 ? 4 : 2);
                        for (int k = 0; k < nsub; ++k) {
                            int x = qx, y = qy, w = 2, h = 2;
                            if (sshape[q] == 1) {
                                h = 1;
                                y += k;
                            } else if (sshape[q] == 2) {
                                w = 1;
                                x += k;
                            } else if (sshape[q] == 3) {
                                w = h = 1;
                                x += k & 1;
                                y += k >> 1;
                            }
                            int dx = parse_mvd(l, x, y, 0), dy = parse_mvd(l, x, y, 1);
                            int px, py;
                            predict_mv(l, x, y, w, r, 0, px, py);
                            fill_mv(l, x, y, w, h, px + dx, py + dy);
                            fill_mvd(l, x, y, w, h, dx, dy);
                        }
                    }
        }
        int cbp = parse_cbp();
        mb->cbp = (uint8_t)cbp;
        if ((cbp & 15) && pps.transform_8x8_mode && no_sub_lt8 && (!direct16 || sps.direct_8x8_inference)) mb->t8x8 = (uint8_t)parse_transform8x8();
        if (cbp) {
            int d = parse_qp_delta();
            last_dqp_nz = d != 0;
            if (d < -26 || d > 25) fail("mb %d: mb_qp_delta %d", mbaddr, d);
            set_qp((qp + d + 52) % 52);
            parse_residual(cbp);
        } else {
            last_dqp_nz = false;
            std::memset(nz_l, 0, sizeof nz_l);
            std::memset(nz_l8, 0, sizeof nz_l8);
            std::memset(nz_c, 0, sizeof nz_c);
        }
        inter_pred_mb();
        add_residual_luma();
        add_residual_chroma();
    }

    int run() {
        c.init_engine(s.data, s.data_end);
        c.init_contexts(sh.type, sh.cabac_init_idc, sh.qp);
        ++st.cabac_idc_used[sh.type == SLICE_I ? 3 : sh.cabac_init_idc];
        qp = sh.qp;
        mbaddr = sh.first_mb;
        const int total = mb_w * mb_h;
        int count = 0;
        for (;;) {
            if (mbaddr >= total) fail("slice runs past the last macroblock of the picture");
            begin_mb();
            bool skip = false;
            if (sh.type != SLICE_I) skip = parse_mb_skip();
            if (skip) {
                decode_skip();
            } else {
                int t;
                if (sh.type == SLICE_I) {
                    t = parse_intra_mb_type(3, true);
                    if (t == 25) decode_ipcm(); else decode_intra(t);
                } else if (sh.type == SLICE_P) {
                    t = parse_mb_type_p();
                    if (t >= 5) {
                        if (t - 5 == 25) decode_ipcm(); else decode_intra(t - 5);
                    } else {
                        decode_inter(t);
                    }
                } else {
                    t = parse_mb_type_b();
                    if (t >= 23) {
                        if (t - 23 == 25) decode_ipcm(); else decode_intra(t - 23);
                    } else {
                        decode_inter(t);
                    }
                }
            }
            if (mb->t8x8) ++st.t8x8;
            ++count;
            if (c.terminate()) break;
            ++mbaddr;
        }
        // 7.3.2.10 rbsp_slice_trailing_bits: rbsp_stop_one_bit, alignment zeros, then only cabac_zero_words
        // 9.3.4.5: the encoder's flush ends the arithmetic codeword with a 1 bit that doubles as rbsp_stop_one_bit, so a
        // decoder in step has at most the tail of that flush left: the bin that ended the slice was read from the last
        // bytes of the payload.  x264 pads its flush with up to two more bytes (a pseudo-random final bit), and
        // cabac_zero_words (0x0000) may follow; anything beyond that means the decoder lost synchronisation earlier.
        const uint8_t* last = c.end;
        while (last > c.p && last[-1] == 0) --last;
        long unread = c.bits_left + 8L * (last - c.p);
        if (unread > 24) fail("slice ending at mb %d: %ld bits of slice data are left over (CABAC desynchronised)", mbaddr, unread);
        if (c.overrun) {
            st.cabac_overrun += c.overrun;
            fail("slice ending at mb %d: the arithmetic decoder read %d bits past the end of the slice", mbaddr, c.overrun);
        }
        return count;
    }
};

}  // namespace

int decode_slice_data(SliceCtx& s) {
    MbDecoder d(s);
    return d.run();
}

}  // namespace evc
