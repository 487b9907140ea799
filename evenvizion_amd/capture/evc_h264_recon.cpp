// evc_h264_recon.cpp -- sample reconstruction of ITU-T Rec. H.264: intra prediction (8.3.1 4x4, 8.3.2 8x8, 8.3.3 16x16,
// 8.3.4 chroma), fractional sample interpolation (8.4.2.2), the inverse transforms (8.5.12, 8.5.13) and the deblocking
// filter (8.7).  Every routine follows the formulae of the clause it names; nothing is approximated.
#include <algorithm>
#include <cstdlib>

#include "evc_h264_int.h"

namespace evc {

// =================================================================================================== intra 4x4 (8.3.1.2)
void pred_intra4x4(uint8_t* dst, int stride, int mode, bool left, bool top, bool topright, bool topleft) {
    // t[-1] = p[-1,-1], t[0..7] = p[0..7,-1]; l[-1] = p[-1,-1], l[0..3] = p[-1,0..3]
    int tbuf[10], lbuf[6];                 // (one spare slot in front: at -O3 the unrolled mode loops make the compiler
    int* t = tbuf + 2;                     //  see subscripts it cannot prove unreachable)
    int* l = lbuf + 2;
    if (top) {
        for (int x = 0; x < 4; ++x) t[x] = dst[-stride + x];
        if (topright)
            for (int x = 4; x < 8; ++x) t[x] = dst[-stride + x];
        else
            for (int x = 4; x < 8; ++x) t[x] = t[3];
    }
    if (left)
        for (int y = 0; y < 4; ++y) l[y] = dst[y * stride - 1];
    if (topleft) t[-1] = l[-1] = dst[-stride - 1];
    auto need = [&](bool ok, const char* what) {
        if (!ok) fail("intra 4x4 prediction mode %d needs the %s samples, which are not available", mode, what);
    };
    int p[4][4];
    switch (mode) {
        case 0:
            need(top, "upper");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) p[y][x] = t[x];
            break;
        case 1:
            need(left, "left");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) p[y][x] = l[y];
            break;
        case 2: {
            int v;
            if (top && left)
                v = (t[0] + t[1] + t[2] + t[3] + l[0] + l[1] + l[2] + l[3] + 4) >> 3;
            else if (left)
                v = (l[0] + l[1] + l[2] + l[3] + 2) >> 2;
            else if (top)
                v = (t[0] + t[1] + t[2] + t[3] + 2) >> 2;
            else
                v = 128;
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) p[y][x] = v;
            break;
        }
        case 3:  // diagonal down left
            need(top, "upper");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x)
                    p[y][x] = (x == 3 && y == 3) ? (t[6] + 3 * t[7] + 2) >> 2 : (t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2;
            break;
        case 4:  // diagonal down right
            need(top && left && topleft, "upper, left and corner");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) {
                    if (x > y)
                        p[y][x] = (t[x - y - 2] + 2 * t[x - y - 1] + t[x - y] + 2) >> 2;
                    else if (x < y)
                        p[y][x] = (l[y - x - 2] + 2 * l[y - x - 1] + l[y - x] + 2) >> 2;
                    else
                        p[y][x] = (t[0] + 2 * t[-1] + l[0] + 2) >> 2;
                }
            break;
        case 5:  // vertical right
            need(top && left && topleft, "upper, left and corner");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) {
                    int z = 2 * x - y;
                    if (z >= 0 && !(z & 1))
                        p[y][x] = (t[x - (y >> 1) - 1] + t[x - (y >> 1)] + 1) >> 1;
                    else if (z >= 0)
                        p[y][x] = (t[x - (y >> 1) - 2] + 2 * t[x - (y >> 1) - 1] + t[x - (y >> 1)] + 2) >> 2;
                    else if (z == -1)
                        p[y][x] = (l[0] + 2 * t[-1] + t[0] + 2) >> 2;
                    else
                        p[y][x] = (l[y - 1] + 2 * l[y - 2] + l[y - 3] + 2) >> 2;
                }
            break;
        case 6:  // horizontal down
            need(top && left && topleft, "upper, left and corner");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) {
                    int z = 2 * y - x;
                    if (z >= 0 && !(z & 1))
                        p[y][x] = (l[y - (x >> 1) - 1] + l[y - (x >> 1)] + 1) >> 1;
                    else if (z >= 0)
                        p[y][x] = (l[y - (x >> 1) - 2] + 2 * l[y - (x >> 1) - 1] + l[y - (x >> 1)] + 2) >> 2;
                    else if (z == -1)
                        p[y][x] = (l[0] + 2 * t[-1] + t[0] + 2) >> 2;
                    else
                        p[y][x] = (t[x - 1] + 2 * t[x - 2] + t[x - 3] + 2) >> 2;
                }
            break;
        case 7:  // vertical left
            need(top, "upper");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) {
                    int i = x + (y >> 1);
                    p[y][x] = (y & 1) ? (t[i] + 2 * t[i + 1] + t[i + 2] + 2) >> 2 : (t[i] + t[i + 1] + 1) >> 1;
                }
            break;
        case 8:  // horizontal up
            need(left, "left");
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) {
                    int z = x + 2 * y;
                    int i = y + (x >> 1);
                    if (z > 5)
                        p[y][x] = l[3];
                    else if (z == 5)
                        p[y][x] = (l[2] + 3 * l[3] + 2) >> 2;
                    else if (z & 1)
                        p[y][x] = (l[i] + 2 * l[i + 1] + l[i + 2] + 2) >> 2;
                    else
                        p[y][x] = (l[i] + l[i + 1] + 1) >> 1;
                }
            break;
        default:
            fail("intra 4x4 prediction mode %d", mode);
    }
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) dst[y * stride + x] = (uint8_t)p[y][x];
}

// =================================================================================================== intra 8x8 (8.3.2.2)
void pred_intra8x8(uint8_t* dst, int stride, int mode, bool left, bool top, bool topright, bool topleft) {
    int rt[17], rl[9];  // unfiltered: rt[0] = corner, rt[1+x] = p[x,-1] (x = 0..15); rl[0] = corner, rl[1+y] = p[-1,y]
    int ft[17], fl[9];  // filtered (8.3.2.2.1)
    if (top) {
        for (int x = 0; x < 8; ++x) rt[1 + x] = dst[-stride + x];
        if (topright)
            for (int x = 8; x < 16; ++x) rt[1 + x] = dst[-stride + x];
        else
            for (int x = 8; x < 16; ++x) rt[1 + x] = rt[8];
    }
    if (left)
        for (int y = 0; y < 8; ++y) rl[1 + y] = dst[y * stride - 1];
    if (topleft) rt[0] = rl[0] = dst[-stride - 1];
    if (top) {
        ft[1] = topleft ? (rt[0] + 2 * rt[1] + rt[2] + 2) >> 2 : (3 * rt[1] + rt[2] + 2) >> 2;
        for (int x = 1; x < 15; ++x) ft[1 + x] = (rt[x] + 2 * rt[1 + x] + rt[2 + x] + 2) >> 2;
        ft[16] = (rt[15] + 3 * rt[16] + 2) >> 2;
    }
    if (topleft) {
        if (top && left)
            ft[0] = (rt[1] + 2 * rt[0] + rl[1] + 2) >> 2;
        else if (top)
            ft[0] = (3 * rt[0] + rt[1] + 2) >> 2;
        else if (left)
            ft[0] = (3 * rt[0] + rl[1] + 2) >> 2;
        else
            ft[0] = rt[0];
        fl[0] = ft[0];
    }
    if (left) {
        fl[1] = topleft ? (rl[0] + 2 * rl[1] + rl[2] + 2) >> 2 : (3 * rl[1] + rl[2] + 2) >> 2;
        for (int y = 1; y < 7; ++y) fl[1 + y] = (rl[y] + 2 * rl[1 + y] + rl[2 + y] + 2) >> 2;
        fl[8] = (rl[7] + 3 * rl[8] + 2) >> 2;
    }
    const int* t = ft + 1;  // t[-1] = corner
    const int* l = fl + 1;
    auto need = [&](bool ok, const char* what) {
        if (!ok) fail("intra 8x8 prediction mode %d needs the %s samples, which are not available", mode, what);
    };
    int p[8][8];
    switch (mode) {
        case 0:
            need(top, "upper");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) p[y][x] = t[x];
            break;
        case 1:
            need(left, "left");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) p[y][x] = l[y];
            break;
        case 2: {
            int v, st = 0, sl = 0;
            if (top)
                for (int x = 0; x < 8; ++x) st += t[x];
            if (left)
                for (int y = 0; y < 8; ++y) sl += l[y];
            if (top && left)
                v = (st + sl + 8) >> 4;
            else if (left)
                v = (sl + 4) >> 3;
            else if (top)
                v = (st + 4) >> 3;
            else
                v = 128;
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) p[y][x] = v;
            break;
        }
        case 3:
            need(top, "upper");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x)
                    p[y][x] = (x == 7 && y == 7) ? (t[14] + 3 * t[15] + 2) >> 2 : (t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2;
            break;
        case 4:
            need(top && left && topleft, "upper, left and corner");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    if (x > y)
                        p[y][x] = (t[x - y - 2] + 2 * t[x - y - 1] + t[x - y] + 2) >> 2;
                    else if (x < y)
                        p[y][x] = (l[y - x - 2] + 2 * l[y - x - 1] + l[y - x] + 2) >> 2;
                    else
                        p[y][x] = (t[0] + 2 * t[-1] + l[0] + 2) >> 2;
                }
            break;
        case 5:
            need(top && left && topleft, "upper, left and corner");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    int z = 2 * x - y;
                    if (z >= 0 && !(z & 1))
                        p[y][x] = (t[x - (y >> 1) - 1] + t[x - (y >> 1)] + 1) >> 1;
                    else if (z >= 0)
                        p[y][x] = (t[x - (y >> 1) - 2] + 2 * t[x - (y >> 1) - 1] + t[x - (y >> 1)] + 2) >> 2;
                    else if (z == -1)
                        p[y][x] = (l[0] + 2 * t[-1] + t[0] + 2) >> 2;
                    else
                        p[y][x] = (l[y - 2 * x - 1] + 2 * l[y - 2 * x - 2] + l[y - 2 * x - 3] + 2) >> 2;
                }
            break;
        case 6:
            need(top && left && topleft, "upper, left and corner");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    int z = 2 * y - x;
                    if (z >= 0 && !(z & 1))
                        p[y][x] = (l[y - (x >> 1) - 1] + l[y - (x >> 1)] + 1) >> 1;
                    else if (z >= 0)
                        p[y][x] = (l[y - (x >> 1) - 2] + 2 * l[y - (x >> 1) - 1] + l[y - (x >> 1)] + 2) >> 2;
                    else if (z == -1)
                        p[y][x] = (l[0] + 2 * t[-1] + t[0] + 2) >> 2;
                    else
                        p[y][x] = (t[x - 2 * y - 1] + 2 * t[x - 2 * y - 2] + t[x - 2 * y - 3] + 2) >> 2;
                }
            break;
        case 7:
            need(top, "upper");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    int i = x + (y >> 1);
                    p[y][x] = (y & 1) ? (t[i] + 2 * t[i + 1] + t[i + 2] + 2) >> 2 : (t[i] + t[i + 1] + 1) >> 1;
                }
            break;
        case 8:
            need(left, "left");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    int z = x + 2 * y;
                    int i = y + (x >> 1);
                    if (z > 13)
                        p[y][x] = l[7];
                    else if (z == 13)
                        p[y][x] = (l[6] + 3 * l[7] + 2) >> 2;
                    else if (z & 1)
                        p[y][x] = (l[i] + 2 * l[i + 1] + l[i + 2] + 2) >> 2;
                    else
                        p[y][x] = (l[i] + l[i + 1] + 1) >> 1;
                }
            break;
        default:
            fail("intra 8x8 prediction mode %d", mode);
    }
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) dst[y * stride + x] = (uint8_t)p[y][x];
}

// =================================================================================================== intra 16x16 (8.3.3)
void pred_intra16x16(uint8_t* dst, int stride, int mode, bool left, bool top, bool topleft) {
    switch (mode) {
        case 0:
            if (!top) fail("intra 16x16 vertical prediction without upper samples");
            for (int y = 15; y >= 0; --y)
                for (int x = 0; x < 16; ++x) dst[y * stride + x] = dst[-stride + x];
            break;
        case 1:
            if (!left) fail("intra 16x16 horizontal prediction without left samples");
            for (int y = 0; y < 16; ++y) {
                int v = dst[y * stride - 1];
                for (int x = 0; x < 16; ++x) dst[y * stride + x] = (uint8_t)v;
            }
            break;
        case 2: {
            int st = 0, sl = 0, v;
            if (top)
                for (int x = 0; x < 16; ++x) st += dst[-stride + x];
            if (left)
                for (int y = 0; y < 16; ++y) sl += dst[y * stride - 1];
            if (top && left)
                v = (st + sl + 16) >> 5;
            else if (left)
                v = (sl + 8) >> 4;
            else if (top)
                v = (st + 8) >> 4;
            else
                v = 128;
            for (int y = 0; y < 16; ++y)
                for (int x = 0; x < 16; ++x) dst[y * stride + x] = (uint8_t)v;
            break;
        }
        case 3: {
            if (!(top && left && topleft)) fail("intra 16x16 plane prediction without its neighbours");
            int H = 0, V = 0;
            for (int i = 0; i < 8; ++i) {
                int a = dst[-stride + 8 + i], b = i == 7 ? dst[-stride - 1] : dst[-stride + 6 - i];
                H += (i + 1) * (a - b);
                int cc = dst[(8 + i) * stride - 1], d = i == 7 ? dst[-stride - 1] : dst[(6 - i) * stride - 1];
                V += (i + 1) * (cc - d);
            }
            int a = 16 * (dst[15 * stride - 1] + dst[-stride + 15]);
            int b = (5 * H + 32) >> 6, cc = (5 * V + 32) >> 6;
            for (int y = 0; y < 16; ++y)
                for (int x = 0; x < 16; ++x) dst[y * stride + x] = (uint8_t)clip1((a + b * (x - 7) + cc * (y - 7) + 16) >> 5);
            break;
        }
        default:
            fail("intra 16x16 prediction mode %d", mode);
    }
}

// =================================================================================================== intra chroma (8.3.4)
void pred_intra_chroma(uint8_t* dst, int stride, int mode, bool left, bool top, bool topleft) {
    switch (mode) {
        case 0:
            for (int blk = 0; blk < 4; ++blk) {
                int xo = (blk & 1) * 4, yo = (blk >> 1) * 4;
                int st = 0, sl = 0, v;
                if (top)
                    for (int x = 0; x < 4; ++x) st += dst[-stride + xo + x];
                if (left)
                    for (int y = 0; y < 4; ++y) sl += dst[(yo + y) * stride - 1];
                if (blk == 0 || blk == 3) {
                    if (top && left)
                        v = (st + sl + 4) >> 3;
                    else if (left)
                        v = (sl + 2) >> 2;
                    else if (top)
                        v = (st + 2) >> 2;
                    else
                        v = 128;
                } else if (blk == 1) {
                    v = top ? (st + 2) >> 2 : (left ? (sl + 2) >> 2 : 128);
                } else {
                    v = left ? (sl + 2) >> 2 : (top ? (st + 2) >> 2 : 128);
                }
                for (int y = 0; y < 4; ++y)
                    for (int x = 0; x < 4; ++x) dst[(yo + y) * stride + xo + x] = (uint8_t)v;
            }
            break;
        case 1:
            if (!left) fail("intra chroma horizontal prediction without left samples");
            for (int y = 0; y < 8; ++y) {
                int v = dst[y * stride - 1];
                for (int x = 0; x < 8; ++x) dst[y * stride + x] = (uint8_t)v;
            }
            break;
        case 2:
            if (!top) fail("intra chroma vertical prediction without upper samples");
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) dst[y * stride + x] = dst[-stride + x];
            break;
        case 3: {
            if (!(top && left && topleft)) fail("intra chroma plane prediction without its neighbours");
            int H = 0, V = 0;
            for (int i = 0; i < 4; ++i) {
                int a = dst[-stride + 4 + i], b = i == 3 ? dst[-stride - 1] : dst[-stride + 2 - i];
                H += (i + 1) * (a - b);
                int cc = dst[(4 + i) * stride - 1], d = i == 3 ? dst[-stride - 1] : dst[(2 - i) * stride - 1];
                V += (i + 1) * (cc - d);
            }
            int a = 16 * (dst[7 * stride - 1] + dst[-stride + 7]);
            int b = (34 * H + 32) >> 6, cc = (34 * V + 32) >> 6;
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) dst[y * stride + x] = (uint8_t)clip1((a + b * (x - 3) + cc * (y - 3) + 16) >> 5);
            break;
        }
        default:
            fail("intra chroma prediction mode %d", mode);
    }
}

// =================================================================================================== transforms
// 8.5.12.2: 4x4 residual transform, rows then columns, (x + 32) >> 6, added to the prediction
void idct4x4_add(uint8_t* dst, int stride, const int32_t* c) {
    int f[16], h[16];
    for (int i = 0; i < 4; ++i) {
        int d0 = c[i * 4], d1 = c[i * 4 + 1], d2 = c[i * 4 + 2], d3 = c[i * 4 + 3];
        int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
        f[i * 4] = e0 + e3;
        f[i * 4 + 1] = e1 + e2;
        f[i * 4 + 2] = e1 - e2;
        f[i * 4 + 3] = e0 - e3;
    }
    for (int j = 0; j < 4; ++j) {
        int f0 = f[j], f1 = f[4 + j], f2 = f[8 + j], f3 = f[12 + j];
        int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = f1 + (f3 >> 1);
        h[j] = g0 + g3;
        h[4 + j] = g1 + g2;
        h[8 + j] = g1 - g2;
        h[12 + j] = g0 - g3;
    }
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) dst[y * stride + x] = (uint8_t)clip1(dst[y * stride + x] + ((h[y * 4 + x] + 32) >> 6));
}

// 8.5.13: one-dimensional 8-point inverse transform
static inline void it8(const int* d, int* o) {
    int a0 = d[0] + d[4], a4 = d[0] - d[4], a2 = (d[2] >> 1) - d[6], a6 = d[2] + (d[6] >> 1);
    int b0 = a0 + a6, b2 = a4 + a2, b4 = a4 - a2, b6 = a0 - a6;
    int a1 = -d[3] + d[5] - d[7] - (d[7] >> 1);
    int a3 = d[1] + d[7] - d[3] - (d[3] >> 1);
    int a5 = -d[1] + d[7] + d[5] + (d[5] >> 1);
    int a7 = d[3] + d[5] + d[1] + (d[1] >> 1);
    int b1 = a1 + (a7 >> 2), b7 = a7 - (a1 >> 2), b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5;
    o[0] = b0 + b7;
    o[1] = b2 + b5;
    o[2] = b4 + b3;
    o[3] = b6 + b1;
    o[4] = b6 - b1;
    o[5] = b4 - b3;
    o[6] = b2 - b5;
    o[7] = b0 - b7;
}

void idct8x8_add(uint8_t* dst, int stride, const int32_t* c) {
    int g[64], m[64];
    for (int i = 0; i < 8; ++i) {
        int in[8], out[8];
        for (int k = 0; k < 8; ++k) in[k] = c[i * 8 + k];
        it8(in, out);
        for (int k = 0; k < 8; ++k) g[i * 8 + k] = out[k];
    }
    for (int j = 0; j < 8; ++j) {
        int in[8], out[8];
        for (int k = 0; k < 8; ++k) in[k] = g[k * 8 + j];
        it8(in, out);
        for (int k = 0; k < 8; ++k) m[k * 8 + j] = out[k];
    }
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) dst[y * stride + x] = (uint8_t)clip1(dst[y * stride + x] + ((m[y * 8 + x] + 32) >> 6));
}

// =================================================================================================== interpolation
// 8.4.2.2.1: luma sample interpolation.  Reference samples outside the picture take the nearest picture sample.
void mc_luma(const Picture& ref, int x, int y, int mvx, int mvy, int w, int h, uint8_t* dst) {
    const int W = ref.mb_w * 16, H = ref.mb_h * 16;
    const int fx = mvx & 3, fy = mvy & 3;
    const int ix = x + (mvx >> 2), iy = y + (mvy >> 2);
    // integer samples needed: columns ix-2 .. ix+w+2, rows iy-2 .. iy+h+2 (clamped to the picture: 8.4.2.2.1)
    int16_t g[21][24];
    const int tw = w + 5, th = h + 5;
    if (ix >= 2 && iy >= 2 && ix + w + 3 <= W && iy + h + 3 <= H) {
        for (int r = 0; r < th; ++r) {
            const uint8_t* row = &ref.Y[(size_t)(iy - 2 + r) * ref.stride + ix - 2];
            for (int c = 0; c < tw; ++c) g[r][c] = row[c];
        }
    } else {
        for (int r = 0; r < th; ++r) {
            const uint8_t* row = &ref.Y[(size_t)clip3(0, H - 1, iy - 2 + r) * ref.stride];
            for (int c = 0; c < tw; ++c) g[r][c] = row[clip3(0, W - 1, ix - 2 + c)];
        }
    }
    // G(xx, yy) = g[yy + 2][xx + 2].  The sixteen positions (Table 8-12) are built from three intermediate planes, each
    // computed once per block: b1 (horizontal 6-tap, unrounded), h1 (vertical 6-tap, unrounded), j (6-tap over b1).
    if ((fx | fy) == 0) {
        for (int yy = 0; yy < h; ++yy)
            for (int xx = 0; xx < w; ++xx) dst[yy * 16 + xx] = (uint8_t)g[yy + 2][xx + 2];
        return;
    }
    int16_t b1[21][16];    // b1[r][xx]: horizontal half sample right of G(xx, r - 2), rows r = 0 .. h + 4
    int16_t h1[16][17];    // h1[yy][xx]: vertical half sample below G(xx, yy), columns xx = 0 .. w
    uint8_t bq[17][16], hq[16][17], jq[16][16];
    const bool need_b = fx != 0, need_h = fy != 0, need_j = (fx == 2 && fy != 0) || (fy == 2 && fx != 0);
    if (need_b) {
        const int r0 = need_j ? 0 : 2, r1 = need_j ? th : h + 3;            // rows yy = 0 .. h (yy + 1 is used by p, q, r)
        for (int r = r0; r < r1; ++r)
            for (int xx = 0; xx < w; ++xx) {
                const int16_t* q = &g[r][xx];
                b1[r][xx] = (int16_t)(q[0] - 5 * q[1] + 20 * q[2] + 20 * q[3] - 5 * q[4] + q[5]);
            }
        for (int yy = 0; yy <= h; ++yy)
            for (int xx = 0; xx < w; ++xx) bq[yy][xx] = (uint8_t)clip1((b1[yy + 2][xx] + 16) >> 5);
    }
    if (need_h) {
        for (int yy = 0; yy < h; ++yy)
            for (int xx = 0; xx <= w; ++xx) {
                const int v = g[yy][xx + 2] - 5 * g[yy + 1][xx + 2] + 20 * g[yy + 2][xx + 2] + 20 * g[yy + 3][xx + 2] - 5 * g[yy + 4][xx + 2] + g[yy + 5][xx + 2];
                h1[yy][xx] = (int16_t)v;
                hq[yy][xx] = (uint8_t)clip1((v + 16) >> 5);
            }
    }
    if (need_j) {
        for (int yy = 0; yy < h; ++yy)
            for (int xx = 0; xx < w; ++xx) {
                const int v = b1[yy][xx] - 5 * b1[yy + 1][xx] + 20 * b1[yy + 2][xx] + 20 * b1[yy + 3][xx] - 5 * b1[yy + 4][xx] + b1[yy + 5][xx];
                jq[yy][xx] = (uint8_t)clip1((v + 512) >> 10);
            }
    }
    (void)h1;
#define MC_LOOP(expr)                                   \
    for (int yy = 0; yy < h; ++yy)                      \
        for (int xx = 0; xx < w; ++xx) dst[yy * 16 + xx] = (uint8_t)(expr);
#define GG(xx_, yy_) g[(yy_) + 2][(xx_) + 2]
    switch (fy * 4 + fx) {
        case 1: MC_LOOP((GG(xx, yy) + bq[yy][xx] + 1) >> 1) break;               // a
        case 2: MC_LOOP(bq[yy][xx]) break;                                        // b
        case 3: MC_LOOP((GG(xx + 1, yy) + bq[yy][xx] + 1) >> 1) break;           // c
        case 4: MC_LOOP((GG(xx, yy) + hq[yy][xx] + 1) >> 1) break;               // d
        case 5: MC_LOOP((bq[yy][xx] + hq[yy][xx] + 1) >> 1) break;               // e
        case 6: MC_LOOP((bq[yy][xx] + jq[yy][xx] + 1) >> 1) break;               // f
        case 7: MC_LOOP((bq[yy][xx] + hq[yy][xx + 1] + 1) >> 1) break;           // g
        case 8: MC_LOOP(hq[yy][xx]) break;                                        // h
        case 9: MC_LOOP((hq[yy][xx] + jq[yy][xx] + 1) >> 1) break;               // i
        case 10: MC_LOOP(jq[yy][xx]) break;                                       // j
        case 11: MC_LOOP((jq[yy][xx] + hq[yy][xx + 1] + 1) >> 1) break;          // k
        case 12: MC_LOOP((GG(xx, yy + 1) + hq[yy][xx] + 1) >> 1) break;          // n
        case 13: MC_LOOP((hq[yy][xx] + bq[yy + 1][xx] + 1) >> 1) break;          // p
        case 14: MC_LOOP((jq[yy][xx] + bq[yy + 1][xx] + 1) >> 1) break;          // q
        default: MC_LOOP((hq[yy][xx + 1] + bq[yy + 1][xx] + 1) >> 1) break;      // r
    }
#undef GG
#undef MC_LOOP
}

// 8.4.2.2.2: chroma sample interpolation (4:2:0: the luma vector in units of 1/8 chroma sample)
void mc_chroma(const Picture& ref, int plane, int x, int y, int mvx, int mvy, int w, int h, uint8_t* dst) {
    const int W = ref.mb_w * 8, H = ref.mb_h * 8;
    const std::vector<uint8_t>& pl = plane ? ref.Cr : ref.Cb;
    const int fx = mvx & 7, fy = mvy & 7;
    const int ix = x + (mvx >> 3), iy = y + (mvy >> 3);
    const int cA = (8 - fx) * (8 - fy), cB = fx * (8 - fy), cC = (8 - fx) * fy, cD = fx * fy;
    if (ix >= 0 && iy >= 0 && ix + w + 1 <= W && iy + h + 1 <= H) {           // wholly inside: no clamping
        for (int yy = 0; yy < h; ++yy) {
            const uint8_t* r0 = &pl[(size_t)(iy + yy) * ref.cstride + ix];
            const uint8_t* r1 = r0 + ref.cstride;
            for (int xx = 0; xx < w; ++xx)
                dst[yy * 8 + xx] = (uint8_t)((cA * r0[xx] + cB * r0[xx + 1] + cC * r1[xx] + cD * r1[xx + 1] + 32) >> 6);
        }
        return;
    }
    for (int yy = 0; yy < h; ++yy) {
        int y0 = clip3(0, H - 1, iy + yy), y1 = clip3(0, H - 1, iy + yy + 1);
        for (int xx = 0; xx < w; ++xx) {
            int x0 = clip3(0, W - 1, ix + xx), x1 = clip3(0, W - 1, ix + xx + 1);
            int A = pl[(size_t)y0 * ref.cstride + x0], B = pl[(size_t)y0 * ref.cstride + x1];
            int C = pl[(size_t)y1 * ref.cstride + x0], D = pl[(size_t)y1 * ref.cstride + x1];
            dst[yy * 8 + xx] = (uint8_t)((cA * A + cB * B + cC * C + cD * D + 32) >> 6);
        }
    }
}

// =================================================================================================== deblocking (8.7)
namespace {

struct Deblock {
    Picture& pic;
    const std::vector<MbInfo>& mbi;
    const PPS& pps;
    int mb_w, mb_h, w4;

    // 8.7.2.1: boundary filtering strength for the edge between 4x4 blocks p (ap) and q (aq), absolute 4x4 coordinates
    int strength(int px, int py, int qx, int qy, bool mb_edge) const {
        const MbInfo& mp = mbi[(size_t)(py >> 2) * mb_w + (px >> 2)];
        const MbInfo& mq = mbi[(size_t)(qy >> 2) * mb_w + (qx >> 2)];
        if (mp.intra || mq.intra) return mb_edge ? 4 : 3;
        if ((mp.cbf_luma >> ((py & 3) * 4 + (px & 3))) & 1) return 2;
        if ((mq.cbf_luma >> ((qy & 3) * 4 + (qx & 3))) & 1) return 2;
        size_t ip = (size_t)py * w4 + px, iq = (size_t)qy * w4 + qx;
        // reference pictures (as pictures, not indices) and motion vectors
        int rp[2] = {pic.ref[0][ip] >= 0 ? pic.ref_id[0][ip] : -1, pic.ref[1][ip] >= 0 ? pic.ref_id[1][ip] : -1};
        int rq[2] = {pic.ref[0][iq] >= 0 ? pic.ref_id[0][iq] : -1, pic.ref[1][iq] >= 0 ? pic.ref_id[1][iq] : -1};
        int np = (rp[0] >= 0) + (rp[1] >= 0), nq = (rq[0] >= 0) + (rq[1] >= 0);
        if (np != nq) return 1;
        auto mvfar = [&](int lp, int lq) {
            return std::abs(pic.mv[lp][ip * 2] - pic.mv[lq][iq * 2]) >= 4 || std::abs(pic.mv[lp][ip * 2 + 1] - pic.mv[lq][iq * 2 + 1]) >= 4;
        };
        if (np == 1) {
            int lp = rp[0] >= 0 ? 0 : 1, lq = rq[0] >= 0 ? 0 : 1;
            if (rp[lp] != rq[lq]) return 1;
            return mvfar(lp, lq) ? 1 : 0;
        }
        // two motion vectors each
        bool same = (rp[0] == rq[0] && rp[1] == rq[1]) || (rp[0] == rq[1] && rp[1] == rq[0]);
        if (!same) return 1;
        if (rp[0] != rp[1]) {
            // two different reference pictures: compare the vectors that refer to the same picture
            if (rp[0] == rq[0]) return (mvfar(0, 0) || mvfar(1, 1)) ? 1 : 0;
            return (mvfar(0, 1) || mvfar(1, 0)) ? 1 : 0;
        }
        // both vectors refer to the same picture: either pairing may match
        bool straight = mvfar(0, 0) || mvfar(1, 1);
        bool crossed = mvfar(0, 1) || mvfar(1, 0);
        return (straight && crossed) ? 1 : 0;
    }

    // filters one line of samples across an edge; `step` is the distance between p0 and p1 (1 for vertical edges)
    static void filter_luma(uint8_t* q0p, int step, int bs, int alpha, int beta, int tc0) {
        int p0 = q0p[-step], p1 = q0p[-2 * step], p2 = q0p[-3 * step], p3 = q0p[-4 * step];
        int q0 = q0p[0], q1 = q0p[step], q2 = q0p[2 * step], q3 = q0p[3 * step];
        if (!(std::abs(p0 - q0) < alpha && std::abs(p1 - p0) < beta && std::abs(q1 - q0) < beta)) return;
        int ap = std::abs(p2 - p0), aq = std::abs(q2 - q0);
        if (bs < 4) {
            int tc = tc0 + (ap < beta) + (aq < beta);
            int delta = clip3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
            q0p[-step] = (uint8_t)clip1(p0 + delta);
            q0p[0] = (uint8_t)clip1(q0 - delta);
            if (ap < beta) q0p[-2 * step] = (uint8_t)(p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
            if (aq < beta) q0p[step] = (uint8_t)(q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
        } else {
            bool small = std::abs(p0 - q0) < ((alpha >> 2) + 2);
            if (ap < beta && small) {
                q0p[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                q0p[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
                q0p[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            } else {
                q0p[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            }
            if (aq < beta && small) {
                q0p[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                q0p[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
                q0p[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
            } else {
                q0p[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
            }
        }
    }
    static void filter_chroma(uint8_t* q0p, int step, int bs, int alpha, int beta, int tc0) {
        int p0 = q0p[-step], p1 = q0p[-2 * step], q0 = q0p[0], q1 = q0p[step];
        if (!(std::abs(p0 - q0) < alpha && std::abs(p1 - p0) < beta && std::abs(q1 - q0) < beta)) return;
        if (bs < 4) {
            int tc = tc0 + 1;
            int delta = clip3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
            q0p[-step] = (uint8_t)clip1(p0 + delta);
            q0p[0] = (uint8_t)clip1(q0 - delta);
        } else {
            q0p[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            q0p[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    }

    void run() {
        for (int mby = 0; mby < mb_h; ++mby)
            for (int mbx = 0; mbx < mb_w; ++mbx) {
                const MbInfo& m = mbi[(size_t)mby * mb_w + mbx];
                if (m.slice_id == 0xFFFF) continue;  // never decoded (reported elsewhere)
                if (m.disable_deblock == 1) continue;
                for (int dir = 0; dir < 2; ++dir) {  // 0: vertical edges (filter horizontally), 1: horizontal edges
                    for (int e = 0; e < 4; ++e) {
                        if (m.t8x8 && (e & 1)) continue;
                        int nx = mbx - (dir == 0 && e == 0), ny = mby - (dir == 1 && e == 0);
                        if (e == 0) {
                            if (nx < 0 || ny < 0) continue;
                            const MbInfo& n = mbi[(size_t)ny * mb_w + nx];
                            if (n.slice_id == 0xFFFF) continue;
                            if (m.disable_deblock == 2 && n.slice_id != m.slice_id) continue;
                        }
                        const MbInfo& pm = e == 0 ? mbi[(size_t)ny * mb_w + nx] : m;
                        int bs[4];
                        bool any = false;
                        for (int k = 0; k < 4; ++k) {
                            int qx = mbx * 4 + (dir == 0 ? e : k), qy = mby * 4 + (dir == 0 ? k : e);
                            int px = qx - (dir == 0), py = qy - (dir == 1);
                            bs[k] = strength(px, py, qx, qy, e == 0);
                            any |= bs[k] != 0;
                        }
                        if (!any) continue;
                        // luma
                        {
                            int qpav = (pm.qp + m.qp + 1) >> 1;
                            int ia = clip3(0, 51, qpav + m.alpha_off), ib = clip3(0, 51, qpav + m.beta_off);
                            int alpha = kAlpha[ia], beta = kBeta[ib];
                            for (int k = 0; k < 4; ++k) {
                                if (!bs[k]) continue;
                                int tc0 = bs[k] < 4 ? kTc0[ia][bs[k] - 1] : 0;
                                for (int i = 0; i < 4; ++i) {
                                    int X = mbx * 16 + (dir == 0 ? e * 4 : k * 4 + i), Y = mby * 16 + (dir == 0 ? k * 4 + i : e * 4);
                                    filter_luma(&pic.Y[(size_t)Y * pic.stride + X], dir == 0 ? 1 : pic.stride, bs[k], alpha, beta, tc0);
                                }
                            }
                        }
                        // chroma: edges 0 and 2 of the luma grid are the chroma 4x4 block edges (4:2:0)
                        if (e & 1) continue;
                        for (int comp = 0; comp < 2; ++comp) {
                            int qpav = (pm.qpc[comp] + m.qpc[comp] + 1) >> 1;
                            int ia = clip3(0, 51, qpav + m.alpha_off), ib = clip3(0, 51, qpav + m.beta_off);
                            int alpha = kAlpha[ia], beta = kBeta[ib];
                            std::vector<uint8_t>& pl = comp ? pic.Cr : pic.Cb;
                            for (int k = 0; k < 4; ++k) {
                                if (!bs[k]) continue;
                                int tc0 = bs[k] < 4 ? kTc0[ia][bs[k] - 1] : 0;
                                for (int i = 0; i < 2; ++i) {
                                    int X = mbx * 8 + (dir == 0 ? e * 2 : k * 2 + i), Y = mby * 8 + (dir == 0 ? k * 2 + i : e * 2);
                                    filter_chroma(&pl[(size_t)Y * pic.cstride + X], dir == 0 ? 1 : pic.cstride, bs[k], alpha, beta, tc0);
                                }
                            }
                        }
                    }
                }
            }
    }
};

}  // namespace

void deblock_picture(Picture& pic, const std::vector<MbInfo>& mbi, const PPS& pps) {
    Deblock d{pic, mbi, pps, pic.mb_w, pic.mb_h, pic.mb_w * 4};
    d.run();
}

}  // namespace evc
