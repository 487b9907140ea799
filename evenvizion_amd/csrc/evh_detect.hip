// evh_detect.hip -- ORB detect + describe for gfx950 (MI355X): gray conversion, 8-level pyramid, FAST-9/16 with
// corner score + 3x3 NMS, per-level selection (FAST score, then Harris), orientation, steered BRIEF.
// Replaces cv2.ORB_create().detectAndCompute (reference: evenvizion/processing/frame_processing.py:59-61).
// Integer stages are exact; float stages use one IEEE operation at a time (-ffp-contract=off).
#include "evh_internal.h"
#include "evh_devmath.h"

namespace {

__device__ __forceinline__ uint32_t gdot4(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_udot4(a, b, acc, false);
}
// 4 gray pixels from 12 BGR bytes in three dwords: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
__device__ __forceinline__ uint32_t gray_bgr12(uint32_t w0, uint32_t w1, uint32_t w2) {
  // weights split in bytes (1868 = 7*256 + 76, 9617 = 37*256 + 145, 4899 = 19*256 + 35): two v_dot4_u32_u8 per
  // pixel on the dword that holds its B,G,R (the fourth byte meets a zero weight); same integers as the scalar form
  const uint32_t WL = 76u | (145u << 8) | (35u << 16), WH = 7u | (37u << 8) | (19u << 16);
  const uint32_t p1 = __builtin_amdgcn_alignbyte(w1, w0, 3), p2 = __builtin_amdgcn_alignbyte(w2, w1, 2);
  const uint32_t y0 = (gdot4(w0, WL, 8192u) + (gdot4(w0, WH, 0u) << 8)) >> 14;
  const uint32_t y1 = (gdot4(p1, WL, 8192u) + (gdot4(p1, WH, 0u) << 8)) >> 14;
  const uint32_t y2 = (gdot4(p2, WL, 8192u) + (gdot4(p2, WH, 0u) << 8)) >> 14;
  const uint32_t y3 = (gdot4(w2, WL << 8, 8192u) + (gdot4(w2, WH << 8, 0u) << 8)) >> 14;
  return __builtin_amdgcn_perm(y1, y0, 0x0C0C0400u) | __builtin_amdgcn_perm(y3, y2, 0x04000C0Cu);
}
// 4 gray pixels (n < 4 at the right edge) from `s`: Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14, or a plain copy
__device__ __forceinline__ uint32_t gray_quad(const uint8_t* __restrict__ s, int channels, int n, int aligned4) {
  uint32_t out = 0;
  if (aligned4 && n == 4) {
    if (channels == 1) out = *reinterpret_cast<const uint32_t*>(s);
    else {
      const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
      out = gray_bgr12(s4[0], s4[1], s4[2]);
    }
  } else if (channels == 1) {
    for (int i = 0; i < n; i++) out |= (uint32_t)s[i] << (8 * i);
  } else {
    for (int i = 0; i < n; i++) {
      uint32_t b = s[3 * i], g = s[3 * i + 1], r = s[3 * i + 2];
      out |= ((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14) << (8 * i);
    }
  }
  return out;
}

// ------------------------------------------------------------------------------------------------------------
// K1: BGR -> gray (Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14) or gray copy, into pyramid level 0.
// One thread = 4 output pixels: three aligned dword loads (12 BGR bytes) -> one dword store; grid.y = frame.
__global__ void k_gray_level0(const uint8_t* __restrict__ src, int channels, int64_t row_stride, int64_t frame_stride,
                              uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes, int w, int h, int dst_stride,
                              int aligned4) {
  const int f = blockIdx.y;
  const int qpr = (w + 3) >> 2;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= qpr * h) return;
  const int y = q / qpr, x = (q - y * qpr) * 4;
  const uint8_t* s = src + (int64_t)f * frame_stride + (int64_t)y * row_stride + (int64_t)x * channels;
  uint8_t* d = pyr + (int64_t)f * pyr_frame_bytes + (int64_t)y * dst_stride + x;
  const uint32_t out = gray_quad(s, channels, min(4, w - x), aligned4);
  *reinterpret_cast<uint32_t*>(d) = out;  // rows are 64-byte aligned and padded, a full dword is always in range
}

// ------------------------------------------------------------------------------------------------------------
// K2: pyramid level l from level l-1, resize(INTER_LINEAR_EXACT): 8.8 fixed-point weights per axis,
// out = ((c0*s00 + c1*s01)*m0 + (c0*s10 + c1*s11)*m1 + 32768) >> 16.  Tables (host-computed): per dst column
// (xofs, xc1), per dst row (yofs, yc1); edge replication is encoded in the tables.
// Workgroup = 128 x 64 output pixels (PDN_H); the source footprint (<= 176 x 82 bytes at scale 1.2) is staged in LDS
// with 16-byte loads, each thread then produces 8 rows x 4 pixels from LDS bytes and stores one dword per row.
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {   // a*b + c, a,b < 2^24 (half-rate VALU;
  uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;   // v_mul_lo_u32 / v_mad_u64_u32 are far slower)
}
// NEVER feed a v_dot4 result to these asm forms: a VALU read of a DOT result needs three wait states on gfx950 and the
// hazard recogniser does not look inside an asm statement (measured in round 2: stale reads, wrong pixels).
__device__ __forceinline__ int mad24s(int a, int b, int c) {   // signed a*b + c, |a|,|b| < 2^23
  int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
// XCD-aware workgroup order.  The dispatcher deals workgroups round-robin to the 8 XCDs (linear id n -> XCD n % 8),
// each with its own L2: neighbouring tiles of one image then sit on eight different L2s, every shared cache line is
// fetched (and every partial line written back) once per XCD and DRAM sees eight interleaved walks.  Remapped, XCD k
// works through one contiguous eighth of the frames, tile after tile.  Measured with tools/ubench/bw_tile.hip on the
// shape of k_gray_pyr1 (480 B x 40 rows): 3.8 -> 5.2 TB/s; on 240 B x 80 rows: 2.5 -> 4.8 TB/s.
// Division-free: BOTH grid dimensions are launched rounded up to a multiple of 8 (xcd_grid), so XCD = blockIdx.x & 7,
// and the caller drops the (tile, frame) pairs past the real counts.
__device__ __forceinline__ void xcd_order(int& tile, int& frame) {
  const uint32_t x = blockIdx.x, y = blockIdx.y;
  tile = (int)((y & 7u) * (gridDim.x >> 3) + (x >> 3));
  frame = (int)((x & 7u) * (gridDim.y >> 3) + (y >> 3));
}
// q = v / d for v * d < 2^20 (tile index / tiles per row), magic = floor(2^20 / d) + 1 from the host
__device__ __forceinline__ int div_magic20(int v, int magic) { return (int)(((uint32_t)v * (uint32_t)magic) >> 20); }
#define PD_W 128
#define PD_H 32
#define PD_SW 192   // staged source row bytes (multiple of 16, >= 1.2*128 + 4 + 15 of slack and alignment)
#define PD_SH 44    // staged source rows (>= 1.2*32 + 5)
// k_pyr_down: output rows per tile.  64 amortises the per-workgroup set-up (tap tables, footprint, ~160 scalar and
// vector instructions) and the two halo rows over twice the pixels: 2.18 -> 1.95 ms for the six launches; 48 rows
// 2.02, 96 rows 2.13, 128 rows 2.5 (LDS then allows 5 workgroups per CU).  k_gray_pyr1 keeps PD_H = 32: its BGR
// staging lives in registers.
#define PDN_H 64
#define PDN_SH (PDN_H * 121 / 100 + 5)   // staged source rows
__global__ __launch_bounds__(256) void k_pyr_down(uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes, int64_t src_off,
                                                  int src_stride, int64_t dst_off, int dst_stride, int dw,
                                                  int dh, int tiles_x, int tx_magic, int ntiles, int nframes,
                                                  const int* __restrict__ xofs, const int* __restrict__ xc1,
                                                  const int* __restrict__ yofs, const int* __restrict__ yc1) {
  __shared__ uint32_t tile32[PDN_SH * PD_SW / 4];
  __shared__ int xo_s[PD_W]; __shared__ int xc_s[PD_W]; __shared__ int yo_s[PDN_H]; __shared__ int yc_s[PDN_H];
  int f, bt;
  xcd_order(bt, f);
  if (bt >= ntiles || f >= nframes) return;             // grid padding (workgroup-uniform)
  const int ty = div_magic20(bt, tx_magic), tx = bt - ty * tiles_x;
  const int x0 = tx * PD_W, y0 = ty * PDN_H;
  const int x1 = min(x0 + PD_W, dw) - 1, y1 = min(y0 + PDN_H, dh) - 1;
  // source footprint straight from the tap tables (scalar loads; a right / bottom edge tap is encoded as
  // (size - 2, weight 256), so ofs + 1 is always inside the source)
  const int sx0 = xofs[x0] & ~15, sy0 = yofs[y0];
  const int ex = xofs[x1] + 1, ey = yofs[y1] + 1;
  const int ncol16 = (ex - sx0) / 16 + 1, nrow = ey - sy0 + 1;     // <= PD_SW/16 = 11, <= PDN_SH
  uint8_t* base = pyr + (int64_t)f * pyr_frame_bytes;   // wave-uniform 64-bit bases; per-lane offsets stay 32-bit
  const uint8_t* simg = base + src_off + sx0;
  uint8_t* dimg = base + dst_off;
  // exact taps of this tile's 128 columns / 32 rows from the host tables (independent of the staging loads)
  if (threadIdx.x < PD_W) {
    const int xi = min(x0 + (int)threadIdx.x, dw - 1);
    xo_s[threadIdx.x] = xofs[xi] - sx0; xc_s[threadIdx.x] = xc1[xi];
  } else if (threadIdx.x < PD_W + PDN_H) {
    const int r = threadIdx.x - PD_W, yi = min(y0 + r, dh - 1);
    yo_s[r] = yofs[yi] - sy0; yc_s[r] = yc1[yi];
  }
  {
    // 16-byte loads: a thread moves one (row, 16-byte column) cell; 16 threads cover a source row of <= 176 bytes
    const int c16 = threadIdx.x & 15;
    if (c16 < ncol16) {
      const uint4* col = reinterpret_cast<const uint4*>(simg) + c16;
      const int stride16 = src_stride >> 4;
      for (int r = threadIdx.x >> 4; r < nrow; r += 16)
        *reinterpret_cast<uint4*>(&tile32[r * (PD_SW / 4) + c16 * 4]) =
            col[mad24((uint32_t)(sy0 + r), (uint32_t)stride16, 0u)];
    }
  }
  __syncthreads();
  const int qx = threadIdx.x & 31, qy = threadIdx.x >> 5;     // 32 quads across, 8 groups of PDN_H/8 rows down
  const int x = x0 + qx * 4;
  if (x >= dw) return;
  const uint8_t* tile = reinterpret_cast<const uint8_t*>(tile32);
#pragma unroll
  for (int rr = 0; rr < PDN_H / 8; rr++) {
    const int y = y0 + qy * (PDN_H / 8) + rr;
    if (y >= dh) break;
    const uint8_t* r0 = tile + yo_s[qy * (PDN_H / 8) + rr] * PD_SW;
    const uint8_t* r1 = r0 + PD_SW;
    const int m1 = yc_s[qy * (PDN_H / 8) + rr];
    // (c0*a + c1*b)*m0 + (c0*a' + c1*b')*m1 + 32768, c0 = 256 - c1, m0 = 256 - m1: 24-bit multiply-adds only (a
    // multiply-add costs the same issue slot as a shift here); the result byte sits in bits 16..23 of v[i] and two
    // v_perm_b32 gather the four of them
    const uint32_t m0 = 256u - (uint32_t)m1;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int o = xo_s[qx * 4 + i];
      const uint32_t c1 = (uint32_t)xc_s[qx * 4 + i], c0 = 256u - c1;
      const uint32_t a0 = r0[o], b0 = r0[o + 1], a1 = r1[o], b1 = r1[o + 1];
      const uint32_t h0 = mad24(a0, c0, mad24(b0, c1, 0u));
      const uint32_t h1 = mad24(a1, c0, mad24(b1, c1, 0u));
      v[i] = mad24(h0, m0, mad24(h1, (uint32_t)m1, 32768u));
    }
    const uint32_t out = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020C0Cu);
    reinterpret_cast<uint32_t*>(dimg)[mad24((uint32_t)y, (uint32_t)(dst_stride >> 2), (uint32_t)(x >> 2))] = out;
  }
}

// K2, row-walking form (round 3).  Same arithmetic as k_pyr_down, different work split: a workgroup owns a 256 x 32 output
// tile, a WAVE owns 8 consecutive output rows of it and a lane 4 output columns.  At scale 1.2 consecutive output rows
// share a source row five times out of six: the wave walks down its rows keeping the horizontal pass h(row) of the two
// source rows in registers and recomputes only the row that is new (10.6 horizontal row-passes per 8 output rows
// instead of 16; the row index is wave-uniform, so the reuse test is a scalar branch and the y tables come through
// scalar loads).  VALU slots per output quad 52 -> 41, LDS byte reads 16 -> 10.6; results bit-identical.
#define PW_W 256
#define PW_H 32
#define PW_SW 352                        // staged source row bytes: >= 1.21*256 + 15 + 2, multiple of 16
#define PW_SH (PW_H * 121 / 100 + 5)     // staged source rows
__global__ __launch_bounds__(256) void k_pyr_walk(uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes, int64_t src_off,
                                                  int src_stride, int64_t dst_off, int dst_stride, int dw, int dh,
                                                  int tiles_x, int tx_magic, int ntiles, int nframes,
                                                  const int* __restrict__ xofs, const int* __restrict__ xc1,
                                                  const int* __restrict__ yofs, const int* __restrict__ yc1) {
  __shared__ uint32_t tile32[PW_SH * PW_SW / 4];
  int f, bt;
  xcd_order(bt, f);
  if (bt >= ntiles || f >= nframes) return;             // grid padding (workgroup-uniform)
  const int ty = div_magic20(bt, tx_magic), tx = bt - ty * tiles_x;
  const int x0 = tx * PW_W, y0 = ty * PW_H;
  const int x1 = min(x0 + PW_W, dw) - 1, y1 = min(y0 + PW_H, dh) - 1;
  const int sx0 = xofs[x0] & ~15, sy0 = yofs[y0];
  const int ex = xofs[x1] + 1, ey = yofs[y1] + 1;
  const int ncol16 = (ex - sx0) / 16 + 1, nrow = ey - sy0 + 1;     // <= PW_SW/16 = 22, <= PW_SH
  uint8_t* base = pyr + (int64_t)f * pyr_frame_bytes;
  const uint8_t* simg = base + src_off + sx0;
  uint8_t* dimg = base + dst_off;
  // this lane's four columns: tap offset inside the staged row and the right-tap weight
  const int lane = threadIdx.x & 63;
  const int x = x0 + lane * 4;
  int o[4]; uint32_t c1[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int xi = min(x + i, dw - 1);
    o[i] = xofs[xi] - sx0; c1[i] = (uint32_t)xc1[xi];
  }
  // this wave's eight output rows: source row and bottom-tap weight, fetched (scalar loads: the row index is
  // wave-uniform) before the staging loads so that the row loop below never waits on memory
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int yb = y0 + w * (PW_H / 4);
  int yo[PW_H / 4]; uint32_t ym[PW_H / 4];
#pragma unroll
  for (int rr = 0; rr < PW_H / 4; rr++) {
    const int yi = min(yb + rr, dh - 1);
    yo[rr] = yofs[yi] - sy0; ym[rr] = (uint32_t)yc1[yi];
  }
  {
    // 32 threads per source row, <= 22 of them move a 16-byte cell; ALL of a thread's cells (<= 6 rows, 8 apart) are
    // requested before the first is stored: one memory round trip per workgroup
    const int c16 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const uint4* col = reinterpret_cast<const uint4*>(simg) + c16;
    const int stride16 = src_stride >> 4;
    // (clamped addresses, unconditional loads: the six values stay in registers; the stores carry the bounds)
    const uint4* colc = reinterpret_cast<const uint4*>(simg) + min(c16, ncol16 - 1);
    uint4 v0, v1, v2, v3, v4, v5;
    static_assert((PW_SH + 7) / 8 == 6, "six staged rows per thread");
#define PW_LD(k) colc[mad24((uint32_t)(sy0 + min(r0 + 8 * (k), nrow - 1)), (uint32_t)stride16, 0u)]
    v0 = PW_LD(0); v1 = PW_LD(1); v2 = PW_LD(2); v3 = PW_LD(3); v4 = PW_LD(4); v5 = PW_LD(5);
#undef PW_LD
    (void)col;
    if (c16 < ncol16) {
      uint4* d = reinterpret_cast<uint4*>(&tile32[r0 * (PW_SW / 4) + c16 * 4]);
      if (r0 < nrow) d[0] = v0;
      if (r0 + 8 < nrow) d[8 * (PW_SW / 16)] = v1;
      if (r0 + 16 < nrow) d[16 * (PW_SW / 16)] = v2;
      if (r0 + 24 < nrow) d[24 * (PW_SW / 16)] = v3;
      if (r0 + 32 < nrow) d[32 * (PW_SW / 16)] = v4;
      if (r0 + 40 < nrow) d[40 * (PW_SW / 16)] = v5;
    }
  }
  __syncthreads();
  const uint8_t* tile = reinterpret_cast<const uint8_t*>(tile32);
  uint32_t h0[4] = {0, 0, 0, 0}, h1[4] = {0, 0, 0, 0};
  auto hpass = [&](int r, uint32_t (&h)[4]) {           // horizontal pass of staged source row r on this lane's columns
    const uint8_t* row = tile + r * PW_SW;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t a = row[o[i]], b = row[o[i] + 1];
      h[i] = mad24(a, 256u - c1[i], mad24(b, c1[i], 0u));
    }
  };
  int prev = -9;
#pragma unroll
  for (int rr = 0; rr < PW_H / 4; rr++) {
    const int y = yb + rr;
    if (y >= dh) break;                                  // wave-uniform
    const int r = yo[rr];
    const uint32_t m1 = ym[rr], m0 = 256u - m1;
    if (r == prev + 1) {
#pragma unroll
      for (int i = 0; i < 4; i++) h0[i] = h1[i];
      hpass(r + 1, h1);
    } else if (r != prev) {
      hpass(r, h0); hpass(r + 1, h1);
    }
    prev = r;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = mad24(h0[i], m0, mad24(h1[i], m1, 32768u));
    const uint32_t out = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020C0Cu);
    if (x < dw) reinterpret_cast<uint32_t*>(dimg)[mad24((uint32_t)y, (uint32_t)(dst_stride >> 2), (uint32_t)(x >> 2))] = out;
  }
}

// K1+K2 fused for level 1: a workgroup owns one 128 x 32 tile of level 1. It converts the level-0 footprint of that
// tile straight from the BGR/gray input into LDS (gray never re-read from HBM), writes the level-0 pixels it OWNS
// (columns [xofs[x0] & ~3, xofs[x0 + 128] & ~3), rows [yofs[y0], yofs[y0 + 32]); the monotone tap tables make these
// ranges a partition of level 0) and then produces its level-1 tile from LDS exactly as k_pyr_down does.
#define GP_SW 176   // LDS gray row bytes (>= 1.21*128 + 7, multiple of 4)
#define GP_SH 44    // LDS gray rows      (>= 1.21*32 + 3)
// BGR4 = 3-channel input, 4-byte aligned rows, width a multiple of 4: every quad is 12 aligned bytes, and ALL of a
// thread's quads (<= 8, one global_load_dwordx3 each) are issued before the first is converted -- one memory round
// trip per workgroup instead of seven (measured -0.13 ms of 2.2 on 2048 720p frames, on top of the XCD order).
template <bool BGR4>
__global__ __launch_bounds__(256) void k_gray_pyr1(const uint8_t* __restrict__ src, int channels, int64_t row_stride,
                                                   int64_t frame_stride, int aligned4, uint8_t* __restrict__ pyr,
                                                   int64_t pyr_frame_bytes, int s_stride, int sw, int sh, int64_t dst_off,
                                                   int dst_stride, int dw, int dh, int tiles_x, int tiles_y,
                                                   int tx_magic, int nframes,
                                                   const int* __restrict__ xofs, const int* __restrict__ xc1,
                                                   const int* __restrict__ yofs, const int* __restrict__ yc1) {
  __shared__ uint32_t tile32[GP_SH * GP_SW / 4];
  __shared__ int xo_s[PD_W]; __shared__ int xc_s[PD_W]; __shared__ int yo_s[PD_H]; __shared__ int yc_s[PD_H];
  int f, bt;
  xcd_order(bt, f);
  if (bt >= tiles_x * tiles_y || f >= nframes) return;  // grid padding (workgroup-uniform)
  const int ty = div_magic20(bt, tx_magic), tx = bt - ty * tiles_x;
  const int x0 = tx * PD_W, y0 = ty * PD_H;
  const int x1 = min(x0 + PD_W, dw) - 1, y1 = min(y0 + PD_H, dh) - 1;
  const int rx0 = xofs[x0] & ~3, ry0 = yofs[y0];
  const int own_x1 = (tx == tiles_x - 1) ? ((sw + 3) & ~3) : (xofs[x0 + PD_W] & ~3);
  const int own_y1 = (ty == tiles_y - 1) ? sh : yofs[y0 + PD_H];
  const int rx1 = max(own_x1, min(xofs[x1] + 2, sw)), ry1 = max(own_y1, min(yofs[y1] + 2, sh));   // exclusive
  const int nqx = (rx1 - rx0 + 3) >> 2, nr = ry1 - ry0;                                             // <= 44, <= 44
  auto stage_taps = [&]() {
    if (threadIdx.x < PD_W) {
      const int xi = min(x0 + (int)threadIdx.x, dw - 1);
      xo_s[threadIdx.x] = xofs[xi] - rx0; xc_s[threadIdx.x] = xc1[xi];
    } else if (threadIdx.x < PD_W + PD_H) {
      const int r = threadIdx.x - PD_W, yi = min(y0 + r, dh - 1);
      yo_s[r] = yofs[yi] - ry0; yc_s[r] = yc1[yi];
    }
  };
  const uint8_t* sframe = src + (int64_t)f * frame_stride;
  uint8_t* base = pyr + (int64_t)f * pyr_frame_bytes;
  const float inv = 1.0f / (float)nqx;
  if constexpr (BGR4) {
    constexpr int GP_IT = (GP_SH * (GP_SW / 4) + 255) / 256;     // 8
    const int nq = nqx * nr;
    uint32_t w0[GP_IT], w1[GP_IT], w2[GP_IT];
#pragma unroll
    for (int it = 0; it < GP_IT; it++) {
      const int q = min((int)threadIdx.x + 256 * it, nq - 1);     // clamped: no load behind a branch
      const int r = (int)(((float)q + 0.5f) * inv);
      const int qx = q - (int)mad24((uint32_t)r, (uint32_t)nqx, 0u);
      const uint32_t so = mad24((uint32_t)(ry0 + r), (uint32_t)row_stride, 3u * (uint32_t)(rx0 + 4 * qx));
      const uint32_t* s4 = reinterpret_cast<const uint32_t*>(sframe + so);
      w0[it] = s4[0]; w1[it] = s4[1]; w2[it] = s4[2];
    }
    stage_taps();
#pragma unroll
    for (int it = 0; it < GP_IT; it++) {
      const int q = (int)threadIdx.x + 256 * it;
      if (q < nq) {
        const int r = (int)(((float)q + 0.5f) * inv);     // exact: q + 0.5 is at least 0.5 away from a multiple of nqx
        const int qx = q - (int)mad24((uint32_t)r, (uint32_t)nqx, 0u);
        const int x = rx0 + 4 * qx, y = ry0 + r;
        const uint32_t g = gray_bgr12(w0[it], w1[it], w2[it]);
        tile32[r * (GP_SW / 4) + qx] = g;
        if (x < own_x1 && y < own_y1)
          *reinterpret_cast<uint32_t*>(base + mad24((uint32_t)y, (uint32_t)s_stride, (uint32_t)x)) = g;
      }
    }
  } else {
  stage_taps();
  for (int q = threadIdx.x; q < nqx * nr; q += 256) {
    const int r = (int)(((float)q + 0.5f) * inv);     // exact: q + 0.5 is at least 0.5 away from a multiple of nqx
    const int qx = q - (int)mad24((uint32_t)r, (uint32_t)nqx, 0u);
    const int x = rx0 + 4 * qx, y = ry0 + r;
    if (x >= sw) continue;                             // padding quad: never read by a tap, never stored
    // per-lane offsets stay 32-bit (frames are < 4096 x 4096 x 3 bytes)
    const uint32_t so = mad24((uint32_t)y, (uint32_t)row_stride, channels == 1 ? (uint32_t)x : 3u * (uint32_t)x);
    const uint32_t g = gray_quad(sframe + so, channels, min(4, sw - x), aligned4);
    tile32[r * (GP_SW / 4) + qx] = g;
    if (x < own_x1 && y < own_y1)
      *reinterpret_cast<uint32_t*>(base + mad24((uint32_t)y, (uint32_t)s_stride, (uint32_t)x)) = g;
  }
  }
  __syncthreads();
  uint8_t* dimg = base + dst_off;
  const uint8_t* tile = reinterpret_cast<const uint8_t*>(tile32);
  const int qx = threadIdx.x & 31, qy = threadIdx.x >> 5;
  const int x = x0 + qx * 4;
  if (x >= dw) return;
#pragma unroll
  for (int rr = 0; rr < PD_H / 8; rr++) {
    const int y = y0 + qy * (PD_H / 8) + rr;
    if (y >= dh) break;
    const uint8_t* r0 = tile + yo_s[qy * (PD_H / 8) + rr] * GP_SW;
    const uint8_t* r1 = r0 + GP_SW;
    const int m1 = yc_s[qy * (PD_H / 8) + rr];
    const uint32_t m0 = 256u - (uint32_t)m1;      // same arithmetic as k_pyr_down
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int o = xo_s[qx * 4 + i];
      const uint32_t c1 = (uint32_t)xc_s[qx * 4 + i], c0 = 256u - c1;
      const uint32_t a0 = r0[o], b0 = r0[o + 1], a1 = r1[o], b1 = r1[o + 1];
      const uint32_t h0 = mad24(a0, c0, mad24(b0, c1, 0u));
      const uint32_t h1 = mad24(a1, c0, mad24(b1, c1, 0u));
      v[i] = mad24(h0, m0, mad24(h1, (uint32_t)m1, 32768u));
    }
    const uint32_t out = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020C0Cu);
    reinterpret_cast<uint32_t*>(dimg)[mad24((uint32_t)y, (uint32_t)(dst_stride >> 2), (uint32_t)(x >> 2))] = out;
  }
}

// K2, two levels per launch (round 3): a workgroup owns one 128 x 32 tile of level l+1.  It stages the
// level l-1 footprint of the level-l pixels that tile needs, produces those level-l pixels into LDS, stores the ones it
// OWNS (the partition of k_gray_pyr1: columns [xofs[x0] & ~3, xofs[x0 + 128] & ~3), rows [yofs[y0], yofs[y0 + 32]) of the
// level l+1 tap tables) and then produces its level l+1 tile from LDS: level l is written once and never read back
// (levels 2..7 in three launches; HBM bytes per frame 3.15 -> 2.39 MB at 720p).  Same arithmetic as k_pyr_down.
#define P2_SW 240    // staged level l-1 row bytes (>= 1.21*176 + 2 + 15, multiple of 16)
#define P2_SH 57     // staged level l-1 rows      (>= 1.21*44 + 3)
struct Pyr2Level { int64_t off; int stride, w, h; const int *xofs, *xc1, *yofs, *yc1; };
__global__ __launch_bounds__(256) void k_pyr_two(uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes, int64_t s_off,
                                                 int s_stride, Pyr2Level M, Pyr2Level D, int tiles_x, int tiles_y,
                                                 int tx_magic, int nframes) {
  __shared__ uint32_t tileS[P2_SH * P2_SW / 4];
  __shared__ uint32_t tileM[GP_SH * GP_SW / 4];
  __shared__ int x1o[GP_SW]; __shared__ int x1c[GP_SW]; __shared__ int y1o[GP_SH]; __shared__ int y1c[GP_SH];
  __shared__ int xo_s[PD_W]; __shared__ int xc_s[PD_W]; __shared__ int yo_s[PD_H]; __shared__ int yc_s[PD_H];
  int f, bt;
  xcd_order(bt, f);
  if (bt >= tiles_x * tiles_y || f >= nframes) return;  // grid padding (workgroup-uniform)
  const int ty = div_magic20(bt, tx_magic), tx = bt - ty * tiles_x;
  const int x0 = tx * PD_W, y0 = ty * PD_H;
  const int x1 = min(x0 + PD_W, D.w) - 1, y1 = min(y0 + PD_H, D.h) - 1;
  // level-l region of this workgroup: what it owns and what its level l+1 tile reads
  const int rx0 = D.xofs[x0] & ~3, ry0 = D.yofs[y0];
  const int own_x1 = (tx == tiles_x - 1) ? ((M.w + 3) & ~3) : (D.xofs[x0 + PD_W] & ~3);
  const int own_y1 = (ty == tiles_y - 1) ? M.h : D.yofs[y0 + PD_H];
  const int rx1 = max(own_x1, min(D.xofs[x1] + 2, M.w)), ry1 = max(own_y1, min(D.yofs[y1] + 2, M.h));   // exclusive
  const int nqx = (rx1 - rx0 + 3) >> 2, nr = ry1 - ry0;                                                 // <= 44, <= 44
  // level l-1 footprint of that region (a right / bottom edge tap is (size - 2, weight 256): ofs + 1 stays inside)
  const int cx1 = min(rx1, M.w) - 1;
  const int sx0 = M.xofs[rx0] & ~15, sy0 = M.yofs[ry0];
  const int ex = M.xofs[cx1] + 1, ey = M.yofs[ry1 - 1] + 1;
  const int ncol16 = (ex - sx0) / 16 + 1, nrow = ey - sy0 + 1;     // <= P2_SW/16 = 15, <= P2_SH
  uint8_t* base = pyr + (int64_t)f * pyr_frame_bytes;
  {
    // 16 threads per staged row (<= 15 move a 16-byte cell); a thread's four cells (rows 16 apart) are all requested
    // before the first is stored: clamped addresses, the stores carry the bounds
    const int c16 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    const uint4* colc = reinterpret_cast<const uint4*>(base + s_off + sx0) + min(c16, ncol16 - 1);
    const int stride16 = s_stride >> 4;
    static_assert((P2_SH + 15) / 16 == 4, "four staged rows per thread");
#define P2_LD(k) colc[mad24((uint32_t)(sy0 + min(r0 + 16 * (k), nrow - 1)), (uint32_t)stride16, 0u)]
    const uint4 v0 = P2_LD(0), v1 = P2_LD(1), v2 = P2_LD(2), v3 = P2_LD(3);
#undef P2_LD
    // the tap tables of both levels while the loads are in flight
    for (int t = threadIdx.x; t < GP_SW + GP_SH + PD_W + PD_H; t += 256) {
      if (t < GP_SW) {
        const int xi = min(rx0 + t, M.w - 1);
        x1o[t] = M.xofs[xi] - sx0; x1c[t] = M.xc1[xi];
      } else if (t < GP_SW + GP_SH) {
        const int r = t - GP_SW, yi = min(ry0 + r, M.h - 1);
        y1o[r] = M.yofs[yi] - sy0; y1c[r] = M.yc1[yi];
      } else if (t < GP_SW + GP_SH + PD_W) {
        const int c = t - GP_SW - GP_SH, xi = min(x0 + c, D.w - 1);
        xo_s[c] = D.xofs[xi] - rx0; xc_s[c] = D.xc1[xi];
      } else {
        const int r = t - GP_SW - GP_SH - PD_W, yi = min(y0 + r, D.h - 1);
        yo_s[r] = D.yofs[yi] - ry0; yc_s[r] = D.yc1[yi];
      }
    }
    if (c16 < ncol16) {
      uint4* d = reinterpret_cast<uint4*>(&tileS[r0 * (P2_SW / 4) + c16 * 4]);
      if (r0 < nrow) d[0] = v0;
      if (r0 + 16 < nrow) d[16 * (P2_SW / 16)] = v1;
      if (r0 + 32 < nrow) d[32 * (P2_SW / 16)] = v2;
      if (r0 + 48 < nrow) d[48 * (P2_SW / 16)] = v3;
    }
  }
  __syncthreads();
  {
    // level l: every quad of the region from the staged level l-1 bytes; owned quads also go to HBM
    const uint8_t* ts = reinterpret_cast<const uint8_t*>(tileS);
    uint8_t* mimg = base + M.off;
    const float inv = 1.0f / (float)nqx;
    for (int q = threadIdx.x; q < nqx * nr; q += 256) {
      const int r = (int)(((float)q + 0.5f) * inv);     // exact: q + 0.5 is at least 0.5 away from a multiple of nqx
      const int qx = q - (int)mad24((uint32_t)r, (uint32_t)nqx, 0u);
      const int x = rx0 + 4 * qx, y = ry0 + r;
      if (x >= M.w) continue;                            // padding quad: never read by a tap, never stored
      const uint8_t* r0p = ts + y1o[r] * P2_SW;
      const uint8_t* r1p = r0p + P2_SW;
      const uint32_t m1 = (uint32_t)y1c[r], m0 = 256u - m1;
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int o = x1o[qx * 4 + i];
        const uint32_t c1 = (uint32_t)x1c[qx * 4 + i], c0 = 256u - c1;
        const uint32_t a0 = r0p[o], b0 = r0p[o + 1], a1 = r1p[o], b1 = r1p[o + 1];
        const uint32_t h0 = mad24(a0, c0, mad24(b0, c1, 0u));
        const uint32_t h1 = mad24(a1, c0, mad24(b1, c1, 0u));
        v[i] = mad24(h0, m0, mad24(h1, m1, 32768u));
      }
      const uint32_t out = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020C0Cu);
      tileM[r * (GP_SW / 4) + qx] = out;
      if (x < own_x1 && y < own_y1)
        *reinterpret_cast<uint32_t*>(mimg + mad24((uint32_t)y, (uint32_t)M.stride, (uint32_t)x)) = out;
    }
  }
  __syncthreads();
  uint8_t* dimg = base + D.off;
  const uint8_t* tile = reinterpret_cast<const uint8_t*>(tileM);
  const int qx = threadIdx.x & 31, qy = threadIdx.x >> 5;
  const int x = x0 + qx * 4;
  if (x >= D.w) return;
#pragma unroll
  for (int rr = 0; rr < PD_H / 8; rr++) {
    const int y = y0 + qy * (PD_H / 8) + rr;
    if (y >= D.h) break;
    const uint8_t* r0 = tile + yo_s[qy * (PD_H / 8) + rr] * GP_SW;
    const uint8_t* r1 = r0 + GP_SW;
    const int m1 = yc_s[qy * (PD_H / 8) + rr];
    const uint32_t m0 = 256u - (uint32_t)m1;      // same arithmetic as k_pyr_down
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int o = xo_s[qx * 4 + i];
      const uint32_t c1 = (uint32_t)xc_s[qx * 4 + i], c0 = 256u - c1;
      const uint32_t a0 = r0[o], b0 = r0[o + 1], a1 = r1[o], b1 = r1[o + 1];
      const uint32_t h0 = mad24(a0, c0, mad24(b0, c1, 0u));
      const uint32_t h1 = mad24(a1, c0, mad24(b1, c1, 0u));
      v[i] = mad24(h0, m0, mad24(h1, (uint32_t)m1, 32768u));
    }
    const uint32_t out = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020C0Cu);
    reinterpret_cast<uint32_t*>(dimg)[mad24((uint32_t)y, (uint32_t)(D.stride >> 2), (uint32_t)(x >> 2))] = out;
  }
}

// ------------------------------------------------------------------------------------------------------------
// K3: FAST-9/16 + corner score + 3x3 NMS + 31-px border filter, all pyramid levels of all frames in one launch.
// Workgroup = 64x32 output tile; the tile plus a 4-pixel halo is staged in LDS (coalesced dword loads), the
// corner score of tile+1 halo is computed into an LDS score plane, NMS + emission read that plane.
// score = max over the 16 arcs of 9 contiguous ring pixels of min(+-(centre - ring)) - 1; corner iff that max
// exceeds the threshold (equivalent to the ">= 9 contiguous strictly brighter/darker" definition).
struct FastArgs {
  EvhLevel lv[EVH_NLEVELS];
  uint8_t* pyr; int64_t pyr_frame_bytes;
  uint32_t* cand; int64_t cand_frame_entries;
  int* cand_count;
  // threshold lifting (k_fast_lift): per (frame, level) score threshold, sampled score histogram, redo flags
  int* thr;            // [F][8]
  unsigned* shist;     // [F][8][256]
  int* redo;           // [1 + F*8]: count, then the (frame * 8 + level) entries to redo densely
  uint32_t* tdesc;     // reference order: [F][total_tiles][8] per-tile burst descriptor (offset in the level's list, 28 row counts)
  int total_tiles;
  int lift_base;       // 1: k_fast_main scores through the pre-test + queue machinery at the base threshold too (reference order)
  int samp_start[EVH_NLEVELS], samp_mod[EVH_NLEVELS];   // sampling lattice of k_fast_sample
  // consecutive frames of one video look alike: with share_group = F > 0 the frames of a call form groups of F
  // consecutive frames and a frame at an odd position of its group takes the sampled score histogram of the frame
  // before it instead of sampling itself (any threshold is exact; a wrong guess only costs the dense redo)
  int share_group;
  // threshold hint carried from the previous detect call of this context (per level: the lower-quartile lifted
  // threshold over the frames of that call, 0 = none): the sample pass then scores its tiles with the lifted machinery at 5/8 of the hint instead
  // of densely -- the histogram is exact above that floor, which is where the new threshold will lie
  const int* hint_in; int* hint_out;
  unsigned* hint_hist;   // [8][256] votes of this call: the lifted threshold of every frame that sampled (bin 0: a failed level)
};

#define FT_W 128                 // output tile width (pixels)
#define FT_H 28                  // output tile height (30 score rows x 34 quads = 1020 quads = 4 full passes of 256)
#define FR_DW ((FT_W + 16) / 4)  // staged raw row: x0-8 .. x0+135, 36 dwords
#define FR_H (FT_H + 8)          // staged raw rows: y0-4 .. y0+FT_H+3
#define FS_DW ((FT_W + 8) / 4)   // score row: x0-4 .. x0+131, 34 quads (dwords of 4 byte scores)
#define FS_H (FT_H + 2)          // score rows: y0-1 .. y0+FT_H
#ifdef EVH_FAST_TWO_LEVEL
#define FSC_CAP (FT_W * FT_H / 2) // the A/B variant parks its first-level queue (<= 1020 quad indices) in the same words
#else
#define FSC_CAP 512              // scored-pixel list of the lifted path
#endif

// byte B (relative to the quad's own dword M; -4..-1 = left neighbour dword, 4..7 = right neighbour dword)
template <int B>
__device__ __forceinline__ int rbyte(uint32_t L, uint32_t M, uint32_t R) {
  if constexpr (B < 0) return (int)((L >> (8 * (4 + B))) & 0xFFu);
  else if constexpr (B < 4) return (int)((M >> (8 * B)) & 0xFFu);
  else return (int)((R >> (8 * (B - 4))) & 0xFFu);
}
// corner score of pixel J (0..3) of a quad; L/M/R[0..6] = the three dwords of rows y-3 .. y+3. Branch-free.
template <int J>
__device__ __forceinline__ void fast_diffs_px(const uint32_t (&L)[7], const uint32_t (&M)[7], const uint32_t (&R)[7],
                                              short (&d)[16]) {
  const int v = rbyte<J>(L[3], M[3], R[3]);
  d[0] = (short)(v - rbyte<J>(L[6], M[6], R[6]));       d[1] = (short)(v - rbyte<J + 1>(L[6], M[6], R[6]));
  d[2] = (short)(v - rbyte<J + 2>(L[5], M[5], R[5]));   d[3] = (short)(v - rbyte<J + 3>(L[4], M[4], R[4]));
  d[4] = (short)(v - rbyte<J + 3>(L[3], M[3], R[3]));   d[5] = (short)(v - rbyte<J + 3>(L[2], M[2], R[2]));
  d[6] = (short)(v - rbyte<J + 2>(L[1], M[1], R[1]));   d[7] = (short)(v - rbyte<J + 1>(L[0], M[0], R[0]));
  d[8] = (short)(v - rbyte<J>(L[0], M[0], R[0]));       d[9] = (short)(v - rbyte<J - 1>(L[0], M[0], R[0]));
  d[10] = (short)(v - rbyte<J - 2>(L[1], M[1], R[1]));  d[11] = (short)(v - rbyte<J - 3>(L[2], M[2], R[2]));
  d[12] = (short)(v - rbyte<J - 3>(L[3], M[3], R[3]));  d[13] = (short)(v - rbyte<J - 3>(L[4], M[4], R[4]));
  d[14] = (short)(v - rbyte<J - 2>(L[5], M[5], R[5]));  d[15] = (short)(v - rbyte<J - 1>(L[6], M[6], R[6]));
}

// 16-bit VALU min/max issue at full rate on gfx950 (measured: 2 cycles per wave64 instruction), the 32-bit and
// 3-input forms at half rate; the differences centre-ring fit int16, so the score trees run on the low halves.
typedef short i16;
__device__ __forceinline__ i16 mn16(i16 a, i16 b) { return a < b ? a : b; }
__device__ __forceinline__ i16 mx16(i16 a, i16 b) { return a > b ? a : b; }
__device__ __forceinline__ int sext16(i16 a) { return (int)a; }

// Exact corner score from the 16 differences d[k] = centre - ring[k] (low 16 bits significant):
// max over the 16 circular arcs of 9 of min(d) (darker) and of min(-d) (brighter).  Sliding minimum by block
// prefix/suffix scans (blocks 0..7 and 8..15; arc k = 8b+r is suffix_b[r] joined with prefix_{b^1}[r]).
__device__ __forceinline__ int fast_score_from_d(const i16 (&d)[16]) {
  i16 Pn[2][8], Sn[2][8], Px[2][8], Sx[2][8];
#pragma unroll
  for (int b = 0; b < 2; b++) {
    Pn[b][0] = d[8 * b]; Px[b][0] = d[8 * b];
    Sn[b][7] = d[8 * b + 7]; Sx[b][7] = d[8 * b + 7];
#pragma unroll
    for (int r = 1; r < 8; r++) {
      Pn[b][r] = mn16(Pn[b][r - 1], d[8 * b + r]);
      Px[b][r] = mx16(Px[b][r - 1], d[8 * b + r]);
      Sn[b][7 - r] = mn16(Sn[b][8 - r], d[8 * b + 7 - r]);
      Sx[b][7 - r] = mx16(Sx[b][8 - r], d[8 * b + 7 - r]);
    }
  }
  i16 a = mn16(Sn[0][0], Pn[1][0]);   // max over arcs of min(d)
  i16 m = mx16(Sx[0][0], Px[1][0]);   // min over arcs of max(d)
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (b == 0 && r == 0) continue;
      a = mx16(a, mn16(Sn[b][r], Pn[b ^ 1][r]));
      m = mn16(m, mx16(Sx[b][r], Px[b ^ 1][r]));
    }
  const int best = max(sext16(a), -sext16(m));
  return best > EVH_FAST_THR ? best - 1 : 0;
}

#define FQ_PITCH (FS_DW * 4)   // score plane pitch in bytes (136)


struct FastLds {
  alignas(16) uint32_t raw[FR_H * FR_DW];   // 36 rows x 36 dwords: rows y0-4.., columns x0-8.. (16-byte staging stores)
  alignas(16) uint32_t score[FS_H * FS_DW];   // 30 x 34 quads of byte scores: rows y0-1.., columns x0-4..
  // lst: NMS output, at most one corner per 2x2 block (896 entries).  The lifted path uses the same words first as
  // its queue of quads with a pixel that passes the pre-test (<= 1020 entries, quad index | pass bits << 16): the
  // queue is dead before NMS writes the list.
  uint32_t lst[FS_H * FS_DW + 4];
  // lifted path: pixels whose exact score reached T.  A few dozen per tile; a tile with more than FSC_CAP takes the
  // full-plane NMS instead (fast_nms_collect), so the list can be short: 17.0 -> 14.5 KB of LDS per workgroup lets 11
  // instead of 9 workgroups sit on a compute unit while some of them are down to their tail wave
  uint16_t scored[FSC_CAP];
  alignas(16) uint32_t sink[4];      // target of the second staging store of threads that have no second item
  int lcnt, gbase, qcnt, scnt, q1cnt;
  int wtot[4];                       // ordered collection: survivors per wave of the current pass
  uint32_t rowcnt[8];                // ordered collection: survivors per tile row, one byte each (FT_H = 28 rows)
};

// stage rows y0-4 .. y0+FT_H+3, columns x0-8 .. x0+135 with 16-byte loads (data outside the image reads as 0: it
// only feeds pixels whose centre is outside the testable range, which are never scored); clears the counters
__device__ __forceinline__ void fast_stage(FastLds& S, const uint8_t* img, const EvhLevel& L, int x0, int y0) {
  if (threadIdx.x == 0) { S.lcnt = 0; S.qcnt = 0; S.scnt = 0; S.q1cnt = 0; }
  if (threadIdx.x >= 8 && threadIdx.x < 16) S.rowcnt[threadIdx.x - 8] = 0;
  // 16-byte items (x0 - 8 = 16 + 128 tx is 16-byte aligned, a staged row is 9 of them): item i = (row i / 9,
  // column i % 9), 324 items = 2 per thread at most; +256 items = +28 rows +4 columns.  Rows are padded to 64 bytes,
  // so an item is wholly inside [0, stride) or wholly outside.
  static_assert(FR_DW % 4 == 0 && ((EVH_FAST_OX - 8) % 16) == 0 && (FT_W % 16) == 0, "16-byte staging");
  constexpr int C16 = FR_DW / 4;
  constexpr int NITEM = FR_H * C16;                     // 324 items: two per thread at most
  static_assert(NITEM > 256 && NITEM <= 512, "two staging items per thread");
  const int stride16 = L.stride >> 4;
  // workgroup-uniform: every staged byte exists (all tiles but those on the right / bottom edge of a level)
  const bool inside = y0 >= 4 && y0 + FT_H + 4 <= L.h && x0 >= 8 && x0 + FT_W + 8 <= L.stride;
  const int ra = (int)threadIdx.x / C16, ca = (int)threadIdx.x - ra * C16;
  int rb = ra + 256 / C16, cb = ca + 256 % C16;
  if (cb >= C16) { cb -= C16; rb++; }
  const bool has_b = (int)threadIdx.x + 256 < NITEM;
  if (!has_b) { rb = ra; cb = ca; }      // no second item: request the first one again (same line, no extra traffic)
  const uint4* img16 = reinterpret_cast<const uint4*>(img);
  // BOTH items are requested before either is stored (clamped addresses, unconditional loads: one memory round trip
  // per workgroup -- the predicated form compiled to load, wait, store, load, wait, store)
  const int ya = y0 - 4 + ra, xa = x0 - 8 + ca * 16, yb = y0 - 4 + rb, xb = x0 - 8 + cb * 16;
  const int xmax = L.stride - 16;
  const uint4 la = img16[mad24s(min(max(ya, 0), L.h - 1), stride16, min(max(xa, 0), xmax) >> 4)];
  const uint4 lb = img16[mad24s(min(max(yb, 0), L.h - 1), stride16, min(max(xb, 0), xmax) >> 4)];
  // straight-line stores (a thread without a second item writes it to a sink word): nothing between the two loads and
  // the two stores for the compiler to sink a load into
  uint4* da = reinterpret_cast<uint4*>(&S.raw[ra * FR_DW + ca * 4]);
  uint4* db = has_b ? reinterpret_cast<uint4*>(&S.raw[rb * FR_DW + cb * 4]) : reinterpret_cast<uint4*>(S.sink);
  *da = la;
  *db = lb;
  if (!inside) {                                          // edge tiles (workgroup-uniform): what lies outside the level reads as 0
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    if (!(xa >= 0 && xa < L.stride && ya >= 0 && ya < L.h)) *da = z;
    if (has_b && !(xb >= 0 && xb < L.stride && yb >= 0 && yb < L.h)) *db = z;
  }
}

// dense path: exact scores (threshold 20) of rows y0-1 .. y0+32, quads x0-4 .. x0+131; one thread = 4 adjacent
// pixels, ring bytes taken straight out of the row dwords (SDWA), branch-free
__device__ __forceinline__ void fast_dense_scores(FastLds& S, const EvhLevel& L, int x0, int y0) {
  for (int i = threadIdx.x; i < FS_H * FS_DW; i += 256) {
    const int sr = i / FS_DW, sq = i - sr * FS_DW;
    const int y = y0 - 1 + sr, xq = x0 - 4 + sq * 4;
    uint32_t out = 0;
    if (y >= 3 && y < L.h - 3 && xq + 3 >= 3 && xq < L.w - 3) {      // wave-divergent only at image borders
      uint32_t Lr[7], Mr[7], Rr[7];
      const uint32_t* p = S.raw + sr * FR_DW + sq;                    // row (y-3), dword of x = xq-4
#pragma unroll
      for (int r = 0; r < 7; r++) { Lr[r] = p[r * FR_DW]; Mr[r] = p[r * FR_DW + 1]; Rr[r] = p[r * FR_DW + 2]; }
      i16 d[16];
      fast_diffs_px<0>(Lr, Mr, Rr, d); int s0 = fast_score_from_d(d);
      fast_diffs_px<1>(Lr, Mr, Rr, d); int s1 = fast_score_from_d(d);
      fast_diffs_px<2>(Lr, Mr, Rr, d); int s2 = fast_score_from_d(d);
      fast_diffs_px<3>(Lr, Mr, Rr, d); int s3 = fast_score_from_d(d);
      if (xq < 3 || xq >= L.w - 3) s0 = 0;
      if (xq + 1 < 3 || xq + 1 >= L.w - 3) s1 = 0;
      if (xq + 2 < 3 || xq + 2 >= L.w - 3) s2 = 0;
      if (xq + 3 < 3 || xq + 3 >= L.w - 3) s3 = 0;
      out = (uint32_t)s0 | ((uint32_t)s1 << 8) | ((uint32_t)s2 << 16) | ((uint32_t)s3 << 24);
    }
    S.score[i] = out;
  }
}

// 4-point pre-test, byte-parallel (4 pixels per dword).  Bit 7 of each result byte is set where the pixel PASSES:
// centre - ring > T for two adjacent compass points (D), or ring - centre > T for two adjacent ones (B).
// Adjacent pairs of a 4-cycle: (D0&D4)|(D4&D8)|(D8&D12)|(D12&D0) == (D0|D8)&(D4|D12).  K4 = (T+1) * 0x01010101, T+1 <= 127.
// swar_ge: bit 7 of every byte = (a >= b), from the 7-bit difference t = (a|H) - (b&~H) which never borrows.
// Three-input boolean ops are spelled as v_bitop3_b32 explicitly: it issues at the full VALU rate on gfx950 while
// v_or3 / v_and_or (what the compiler picks for the same expressions) issue at half rate
// (profiles/r01_valu_issue_rates.txt).  Truth table = the expression evaluated on (0xF0, 0xCC, 0xAA).
template <class F>
constexpr uint32_t tt3(F f) { return f(0xF0u, 0xCCu, 0xAAu) & 0xFFu; }
#define BITOP3(a, b, c, EXPR) \
  __builtin_amdgcn_bitop3_b32((a), (b), (c), tt3([](uint32_t A, uint32_t B, uint32_t C) { return (EXPR); }))
__device__ __forceinline__ uint32_t swar_ge(uint32_t aH, uint32_t a, uint32_t b, uint32_t bL) {
  const uint32_t t = aH - bL;
  return BITOP3(a, b, t, (A & ~B) | (~(A ^ B) & C));
}
__device__ __forceinline__ uint32_t pretest_pass4(uint32_t c, uint32_t rd, uint32_t rr, uint32_t ru, uint32_t rl, uint32_t K4) {
  const uint32_t H = 0x80808080u, Lm = 0x7F7F7F7Fu;
  const uint32_t t = (c | H) - K4;                          // 128 + (c & 127) - K per byte
  const uint32_t cl = BITOP3(t, c, Lm, A & (B | C));        // c - K where c >= K;          bit 7 of (c | t): c >= K
  const uint32_t u = (c & Lm) + K4;                         // (c & 127) + K <= 254 per byte
  const uint32_t ch = BITOP3(u, c, H, A | (B & C));         // c + K where it fits a byte;  bit 7 of ~(c & u): it does
  const uint32_t clH = cl | H, chL = ch & Lm;
  const uint32_t D0 = swar_ge(clH, cl, rd, rd & Lm), D4 = swar_ge(clH, cl, rr, rr & Lm);
  const uint32_t D8 = swar_ge(clH, cl, ru, ru & Lm), D12 = swar_ge(clH, cl, rl, rl & Lm);
  const uint32_t B0 = swar_ge(rd | H, rd, ch, chL), B4 = swar_ge(rr | H, rr, ch, chL);
  const uint32_t B8 = swar_ge(ru | H, ru, ch, chL), B12 = swar_ge(rl | H, rl, ch, chL);
  const uint32_t Dx = D0 | D8, Bx = B0 | B8;
  const uint32_t Dy = BITOP3(D4, D12, Dx, (A | B) & C), By = BITOP3(B4, B12, Bx, (A | B) & C);
  const uint32_t Dm = BITOP3(Dy, c, t, A & (B | C));        // & (c >= K)
  const uint32_t Bm = BITOP3(By, c, u, A & ~(B & C));       // & (c + K <= 255)
  return BITOP3(Dm, Bm, H, (A | B) & C);
}

// the vertical ring pair alone (points 0 and 8): every 9-arc holds one of them, so "neither differs from the centre by
// more than T" rejects exactly.  First level of the two-level variant (EVH_FAST_TWO_LEVEL, an A/B build: see
// profiles/r03_fast_two_level_ab.txt); half the comparison network and no v_alignbyte.
__device__ __forceinline__ uint32_t vert_pass4(uint32_t c, uint32_t rd, uint32_t ru, uint32_t K4) {
  const uint32_t H = 0x80808080u, Lm = 0x7F7F7F7Fu;
  const uint32_t t = (c | H) - K4;
  const uint32_t cl = BITOP3(t, c, Lm, A & (B | C));
  const uint32_t u = (c & Lm) + K4;
  const uint32_t ch = BITOP3(u, c, H, A | (B & C));
  const uint32_t clH = cl | H, chL = ch & Lm;
  const uint32_t D0 = swar_ge(clH, cl, rd, rd & Lm), D8 = swar_ge(clH, cl, ru, ru & Lm);
  const uint32_t B0 = swar_ge(rd | H, rd, ch, chL), B8 = swar_ge(ru | H, ru, ch, chL);
  const uint32_t Dm = BITOP3(D0 | D8, c, t, A & (B | C));
  const uint32_t Bm = BITOP3(B0 | B8, c, u, A & ~(B & C));
  return BITOP3(Dm, Bm, H, (A | B) & C);
}

// The segment test itself, byte-parallel: bit 7 of byte j = pixel j of the quad IS a corner at threshold T (nine contiguous
// ring pixels all darker than centre - T or all brighter than centre + T), K4 = (T + 1) * 0x01010101.  p = the quad's centre row
// in the staged tile (p[0], p[1], p[2] = the dwords of x-4.., x.., x+4..); ring byte k of the four pixels = one dword, taken
// straight (dx = 0) or cut out of two neighbours with v_alignbyte.  Contiguity of 9 out of 16 (cyclic) with three-input ANDs:
// A3[k] = M[k] & M[k+1] & M[k+2], A9[k] = A3[k] & A3[k+3] & A3[k+6], any = OR_k A9[k] -- 40 v_bitop3 per polarity.
// Used where every corner at the base threshold is wanted (reference key-point order): the exact score is then computed for the
// corners only (~10 % of the pixels of a textured frame) instead of for every pixel.
__device__ __forceinline__ uint32_t corner16_pass4(const uint32_t* p, uint32_t K4) {
  const uint32_t H = 0x80808080u, Lm = 0x7F7F7F7Fu;
  const uint32_t c = p[1];
  const uint32_t t = (c | H) - K4;
  const uint32_t cl = BITOP3(t, c, Lm, A & (B | C));
  const uint32_t u = (c & Lm) + K4;
  const uint32_t ch = BITOP3(u, c, H, A | (B & C));
  const uint32_t clH = cl | H, chL = ch & Lm;
  uint32_t r[16];
  {
    const uint32_t* q = p + 3 * FR_DW;                       // row y + 3: ring 15, 0, 1
    const uint32_t l = q[0], m = q[1], rr = q[2];
    r[0] = m; r[1] = __builtin_amdgcn_alignbyte(rr, m, 1); r[15] = __builtin_amdgcn_alignbyte(m, l, 3);
  }
  {
    const uint32_t* q = p + 2 * FR_DW;                       // row y + 2: ring 14, 2
    r[2] = __builtin_amdgcn_alignbyte(q[2], q[1], 2); r[14] = __builtin_amdgcn_alignbyte(q[1], q[0], 2);
  }
  {
    const uint32_t* q = p + FR_DW;                           // row y + 1: ring 13, 3
    r[3] = __builtin_amdgcn_alignbyte(q[2], q[1], 3); r[13] = __builtin_amdgcn_alignbyte(q[1], q[0], 1);
  }
  r[4] = __builtin_amdgcn_alignbyte(p[2], c, 3); r[12] = __builtin_amdgcn_alignbyte(c, p[0], 1);   // row y: ring 12, 4
  {
    const uint32_t* q = p - FR_DW;                           // row y - 1: ring 11, 5
    r[5] = __builtin_amdgcn_alignbyte(q[2], q[1], 3); r[11] = __builtin_amdgcn_alignbyte(q[1], q[0], 1);
  }
  {
    const uint32_t* q = p - 2 * FR_DW;                       // row y - 2: ring 10, 6
    r[6] = __builtin_amdgcn_alignbyte(q[2], q[1], 2); r[10] = __builtin_amdgcn_alignbyte(q[1], q[0], 2);
  }
  {
    const uint32_t* q = p - 3 * FR_DW;                       // row y - 3: ring 9, 8, 7
    const uint32_t l = q[0], m = q[1], rr = q[2];
    r[8] = m; r[7] = __builtin_amdgcn_alignbyte(rr, m, 1); r[9] = __builtin_amdgcn_alignbyte(m, l, 3);
  }
  uint32_t D[16], Bq[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    D[k] = swar_ge(clH, cl, r[k], r[k] & Lm);                // centre - (T + 1) >= ring: darker
    Bq[k] = swar_ge(r[k] | H, r[k], ch, chL);                // ring >= centre + (T + 1): brighter
  }
  uint32_t d3[16], b3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    d3[k] = BITOP3(D[k], D[(k + 1) & 15], D[(k + 2) & 15], A & B & C);
    b3[k] = BITOP3(Bq[k], Bq[(k + 1) & 15], Bq[(k + 2) & 15], A & B & C);
  }
  uint32_t dany = 0, bany = 0;
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const uint32_t d9a = BITOP3(d3[k], d3[(k + 3) & 15], d3[(k + 6) & 15], A & B & C);
    const uint32_t d9b = BITOP3(d3[k + 1], d3[(k + 4) & 15], d3[(k + 7) & 15], A & B & C);
    dany = BITOP3(dany, d9a, d9b, A | B | C);
    const uint32_t b9a = BITOP3(b3[k], b3[(k + 3) & 15], b3[(k + 6) & 15], A & B & C);
    const uint32_t b9b = BITOP3(b3[k + 1], b3[(k + 4) & 15], b3[(k + 7) & 15], A & B & C);
    bany = BITOP3(bany, b9a, b9b, A | B | C);
  }
  const uint32_t Dm = BITOP3(dany, c, t, A & (B | C));       // & (centre >= T + 1): centre - (T + 1) did not wrap
  const uint32_t Bm = BITOP3(bany, c, u, A & ~(B & C));      // & (centre + T + 1 <= 255)
  return BITOP3(Dm, Bm, H, (A | B) & C);
}

// lifted path: only scores >= T are produced.  Phase A: 4-point pre-test at T (any 9-arc holds two adjacent
// compass points), four pixels per 32-bit operation; a quad with at least one passing pixel is queued
// (quad index | pass bits << 16).  Phase B: exact score of the queued pixels, 4 lanes per queued quad.
template <bool FULL16 = false>
__device__ __forceinline__ void fast_lift_scores(FastLds& S, const EvhLevel& L, int x0, int y0, int T) {
  const uint32_t K4 = (uint32_t)(T + 1) * 0x01010101u;
  // tiles whose whole score plane lies inside the testable range need no per-pixel range checks (wave-uniform)
  const bool interior = (y0 - 1 >= 3) && (y0 + FT_H < L.h - 3) && (x0 - 4 >= 3) && (x0 + FT_W + 3 < L.w - 3);
  // a thread takes quads tid, tid+256, tid+512, tid+768 of the 30 x 34 quad grid; the per-quad pass words (bit 7 of
  // byte j = pixel j passes) stay in registers and are queued once after the loop with one LDS atomic per wave
  uint32_t P[4];
  int sr = threadIdx.x / FS_DW, sq = threadIdx.x - sr * FS_DW;
  static_assert((FS_H * FS_DW) % 4 == 0 && FS_H * FS_DW <= 1024, "one 16-byte store per thread clears the score plane");
  if (threadIdx.x < FS_H * FS_DW / 4)                           // phase B overwrites the bytes that reach T
    reinterpret_cast<uint4*>(S.score)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
#ifdef EVH_FAST_TWO_LEVEL
  {
    // level 1: the vertical pair on every quad; survivors (quad index) -> q1, which lives in the words of S.scored (dead
    // until phase B)
    uint16_t* q1 = S.scored;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = threadIdx.x + 256 * k;
      P[k] = 0;
      if (i < FS_H * FS_DW) {
        const uint32_t* p = S.raw + mad24((uint32_t)(sr + 3), FR_DW, (uint32_t)sq);
        P[k] = vert_pass4(p[1], p[1 + 3 * FR_DW], p[1 - 3 * FR_DW], K4);
      }
      sr += 7; sq += 18;
      if (sq >= FS_DW) { sq -= FS_DW; sr++; }
    }
    {
      const unsigned long long m0 = __ballot(P[0] != 0), m1 = __ballot(P[1] != 0), m2 = __ballot(P[2] != 0),
                               m3 = __ballot(P[3] != 0);
      const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), tot = n0 + n1 + n2 + __popcll(m3);
      if (tot) {
        int base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&S.q1cnt, tot);
        base = __builtin_amdgcn_readfirstlane(base);
#define FQ1_PUSH(k, off, m)                                                                                             \
        if (P[k]) q1[base + (off) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32),                               \
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u))] =    \
            (uint16_t)(threadIdx.x + 256 * (k))
        FQ1_PUSH(0, 0, m0); FQ1_PUSH(1, n0, m1); FQ1_PUSH(2, n0 + n1, m2); FQ1_PUSH(3, n0 + n1 + n2, m3);
#undef FQ1_PUSH
      }
    }
    __syncthreads();
    // level 2: the full four-point test on the survivors, then the quad queue of phase B as before
    const int n1 = S.q1cnt;
    for (int e0 = 0; e0 < n1; e0 += 256) {             // workgroup-uniform
      const int e = e0 + (int)threadIdx.x;
      uint32_t Pq = 0; int qi = 0;
      if (e < n1) {
        qi = q1[e];
        const int sr2 = qi / FS_DW, sq2 = qi - sr2 * FS_DW;
        uint32_t cmask = 0x80808080u;
        if (!interior) {
          const int y = y0 - 1 + sr2, xq = x0 - 4 + sq2 * 4;
          const bool rowok = y >= 3 && y < L.h - 3;
          const int lo = min(max(3 - xq, 0), 4), hi = max(min(L.w - 3 - xq, 4), 0);
          cmask = (rowok && lo < hi) ? (0x80808080u << (8 * lo)) & (0x80808080u >> (8 * (4 - hi))) : 0u;
        }
        const uint32_t* p = S.raw + mad24((uint32_t)(sr2 + 3), FR_DW, (uint32_t)sq2);
        const uint32_t Lc = p[0], Mc = p[1], Rc = p[2], Mu = p[1 - 3 * FR_DW], Md = p[1 + 3 * FR_DW];
        Pq = pretest_pass4(Mc, Md, __builtin_amdgcn_alignbyte(Rc, Mc, 3), Mu, __builtin_amdgcn_alignbyte(Mc, Lc, 1), K4) & cmask;
      }
      const unsigned long long m = __ballot(Pq != 0);
      if (m) {
        int base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&S.qcnt, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (Pq) S.lst[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
            (uint32_t)qi | (((Pq >> 7) | (Pq >> 22)) << 16);
      }
    }
  }
#else
  // two copies of the loop: the interior one (most tiles) is straight-line code, so the LDS reads of its four quads
  // can be issued together instead of each behind its own range test
  auto quads = [&](auto interior_tag) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = threadIdx.x + 256 * k;
      P[k] = 0;
      if (i < FS_H * FS_DW) {
        uint32_t cmask = 0x80808080u;
        bool rowok = true;
        if constexpr (!INTERIOR) {
          const int y = y0 - 1 + sr, xq = x0 - 4 + sq * 4;
          rowok = y >= 3 && y < L.h - 3;
          const int lo = min(max(3 - xq, 0), 4), hi = max(min(L.w - 3 - xq, 4), 0);   // valid pixels j in [lo, hi)
          cmask = lo < hi ? (0x80808080u << (8 * lo)) & (0x80808080u >> (8 * (4 - hi))) : 0u;
        }
        if (INTERIOR || (rowok && cmask)) {
          const uint32_t* p = S.raw + mad24((uint32_t)(sr + 3), FR_DW, (uint32_t)sq);    // centre row, dword of x = xq-4
          if constexpr (FULL16) {
            P[k] = corner16_pass4(p, K4) & cmask;
          } else {
            const uint32_t Lc = p[0], Mc = p[1], Rc = p[2], Mu = p[1 - 3 * FR_DW], Md = p[1 + 3 * FR_DW];
            P[k] = pretest_pass4(Mc, Md, __builtin_amdgcn_alignbyte(Rc, Mc, 3), Mu, __builtin_amdgcn_alignbyte(Mc, Lc, 1), K4) &
                   cmask;
          }
        }
      }
      sr += 7; sq += 18;                                        // +256 quads = +7 rows +18 quads
      if (sq >= FS_DW) { sq -= FS_DW; sr++; }
    }
  };
  if (interior) quads(std::true_type{}); else quads(std::false_type{});
  const uint8_t* rawb = reinterpret_cast<const uint8_t*>(S.raw);
  uint8_t* scoreb = reinterpret_cast<uint8_t*>(S.score);
  // exact score of the pixel `j` of quad `qi`; writes the score byte when it reaches T
  auto score_pixel = [&](int qi, int j, bool list) {
    const int sr2 = qi / FS_DW, sq2 = qi - sr2 * FS_DW;
    const int pos = sr2 * FQ_PITCH + sq2 * 4 + j;
    const uint8_t* p = rawb + (sr2 + 3) * (FR_DW * 4) + sq2 * 4 + j + 4;
    const int W = FR_DW * 4;
    const int v = p[0];
    i16 d[16];
    d[0] = (i16)(v - p[3 * W]);       d[1] = (i16)(v - p[3 * W + 1]);   d[2] = (i16)(v - p[2 * W + 2]);
    d[3] = (i16)(v - p[W + 3]);       d[4] = (i16)(v - p[3]);           d[5] = (i16)(v - p[-W + 3]);
    d[6] = (i16)(v - p[-2 * W + 2]);  d[7] = (i16)(v - p[-3 * W + 1]);  d[8] = (i16)(v - p[-3 * W]);
    d[9] = (i16)(v - p[-3 * W - 1]);  d[10] = (i16)(v - p[-2 * W - 2]); d[11] = (i16)(v - p[-W - 3]);
    d[12] = (i16)(v - p[-3]);         d[13] = (i16)(v - p[W - 3]);      d[14] = (i16)(v - p[2 * W - 2]);
    d[15] = (i16)(v - p[3 * W - 1]);
    const int sc = fast_score_from_d(d);
    if (sc >= T) {
      scoreb[pos] = (uint8_t)sc;
      if (list) {
        const int k = atomicAdd(&S.scnt, 1);
        if (k < FSC_CAP) S.scored[k] = (uint16_t)pos;
      }
    }
  };
  if constexpr (FULL16) {
    // every passing pixel IS a corner (a fifth of the pixels of a textured frame): one lane per corner.  Queue of 16-bit
    // entries quad << 2 | pixel in the words of S.lst (2048 entries); a tile with more corners than that is scored densely.
    uint16_t* pq = reinterpret_cast<uint16_t*>(S.lst);
    constexpr int PQ_CAP = 2 * (FS_H * FS_DW);
    const int mine = __popc(P[0] & 0x80808080u) + __popc(P[1] & 0x80808080u) + __popc(P[2] & 0x80808080u) + __popc(P[3] & 0x80808080u);
    int at = mine ? atomicAdd(&S.qcnt, mine) : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t m = P[k] & 0x80808080u;
      while (m) {
        const int b = __ffs(m) - 1;                  // bit 7, 15, 23 or 31
        m &= m - 1;
        if (at < PQ_CAP) pq[at] = (uint16_t)(((threadIdx.x + 256 * k) << 2) | (b >> 3));
        at++;
      }
    }
    __syncthreads();
    const int np = S.qcnt;
    if (np > PQ_CAP) {                               // workgroup-uniform
      fast_dense_scores(S, L, x0, y0);
      return;
    }
    for (int e = threadIdx.x; e < np; e += 256) {
      const int ent = pq[e];
      score_pixel(ent >> 2, ent & 3, false);
    }
    return;
  }
  {
    const unsigned long long m0 = __ballot(P[0] != 0), m1 = __ballot(P[1] != 0), m2 = __ballot(P[2] != 0),
                             m3 = __ballot(P[3] != 0);
    const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), tot = n0 + n1 + n2 + __popcll(m3);
    if (tot) {                                                  // wave-uniform
      int base = 0;
      if ((threadIdx.x & 63) == 0) base = atomicAdd(&S.qcnt, tot);
      base = __builtin_amdgcn_readfirstlane(base);
      // entry = quad index | bits {16: px0, 24: px1, 17: px2, 25: px3}
#define FQ_PUSH(k, off, m)                                                                                              \
      if (P[k]) S.lst[base + (off) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32),                             \
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u))] =  \
          (uint32_t)(threadIdx.x + 256 * (k)) | (((P[k] >> 7) | (P[k] >> 22)) << 16)
      FQ_PUSH(0, 0, m0); FQ_PUSH(1, n0, m1); FQ_PUSH(2, n0 + n1, m2); FQ_PUSH(3, n0 + n1 + n2, m3);
#undef FQ_PUSH
    }
  }
#endif
  __syncthreads();
  const int nq = S.qcnt;
  // four lanes per queued quad.  (A pixel-granular list -- fewer busy waves -- was measured at +1.0 ms when a few dozen pixels
  // per tile pass: the four lanes of a quad read neighbouring bytes of the same LDS words, scattered pixels conflict on the
  // banks.  With the full segment test a fifth of the pixels pass and the pixel list wins: see FULL16 above.)
  for (int e = threadIdx.x; e < nq * 4; e += 256) {
    const uint32_t ent = S.lst[e >> 2];
    const int j = e & 3;
    if (!((ent >> (16 + 8 * (j & 1) + (j >> 1))) & 1u)) continue;
    score_pixel((int)(ent & 0xFFFu), j, true);
  }
}

// lifted path NMS: only the (few) pixels that reached T are visited
// (run by wave 0 alone: the list holds a few dozen pixels)
__device__ __forceinline__ void fast_nms_scored(FastLds& S, const EvhLevel& L, int x0, int y0) {
  if (!((L.w > 2 * EVH_EDGE) && (L.h > 2 * EVH_EDGE))) return;
  const uint8_t* sc = reinterpret_cast<const uint8_t*>(S.score);
  const int n = min(S.scnt, FSC_CAP);
  for (int i = threadIdx.x; i < n; i += 64) {
    const int pos = S.scored[i];
    const int sr = pos / FQ_PITCH, sx = pos - sr * FQ_PITCH;
    if (sr < 1 || sr > FT_H || sx < 4 || sx >= 4 + FT_W) continue;     // halo pixels belong to neighbouring tiles
    const int x = x0 - 4 + sx, y = y0 - 1 + sr;
    if (x < EVH_EDGE || x >= L.w - EVH_EDGE || y < EVH_EDGE || y >= L.h - EVH_EDGE) continue;
    const uint8_t* c = sc + pos;
    const int s = c[0];
    if (s > c[-1] && s > c[1] && s > c[-FQ_PITCH - 1] && s > c[-FQ_PITCH] && s > c[-FQ_PITCH + 1] &&
        s > c[FQ_PITCH - 1] && s > c[FQ_PITCH] && s > c[FQ_PITCH + 1]) {
      const int slot = atomicAdd(&S.lcnt, 1);
      S.lst[slot] = ((uint32_t)s << 24) | ((uint32_t)y << 12) | (uint32_t)x;
    }
  }
}

// 3x3 non-max suppression (strict '>' against all 8 neighbours) + 31-px border filter -> S.lst / S.lcnt
__device__ __forceinline__ void fast_nms_collect(FastLds& S, const EvhLevel& L, int x0, int y0) {
  if (!((L.w > 2 * EVH_EDGE) && (L.h > 2 * EVH_EDGE))) return;
#pragma unroll 1
  for (int i = threadIdx.x; i < (FT_W / 4) * FT_H; i += 256) {
    const int qr = i / (FT_W / 4), qc = i - qr * (FT_W / 4);
    const int y = y0 + qr, xq = x0 + qc * 4;
    const uint32_t* p = S.score + (qr + 1) * FS_DW + (qc + 1);       // this quad, row y
    const uint32_t m = p[0];
    if (m == 0 || y < EVH_EDGE || y >= L.h - EVH_EDGE) continue;
    const uint32_t lft = p[-1], rgt = p[1];
    const uint32_t um = p[-FS_DW], ul = p[-FS_DW - 1], ur = p[-FS_DW + 1];
    const uint32_t dm = p[FS_DW], dl = p[FS_DW - 1], dr = p[FS_DW + 1];
    // 6-byte windows (x-1 .. x+4) of the three rows
    const uint64_t wu = ((uint64_t)ur << 40) | ((uint64_t)um << 8) | (ul >> 24);
    const uint64_t wm = ((uint64_t)rgt << 40) | ((uint64_t)m << 8) | (lft >> 24);
    const uint64_t wd = ((uint64_t)dr << 40) | ((uint64_t)dm << 8) | (dl >> 24);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int s = (int)((wm >> (8 * (j + 1))) & 0xFF);
      const int x = xq + j;
      if (s == 0 || x < EVH_EDGE || x >= L.w - EVH_EDGE) continue;
      const int n0 = (int)((wm >> (8 * j)) & 0xFF), n1 = (int)((wm >> (8 * (j + 2))) & 0xFF);
      const int u0 = (int)((wu >> (8 * j)) & 0xFF), u1 = (int)((wu >> (8 * (j + 1))) & 0xFF), u2 = (int)((wu >> (8 * (j + 2))) & 0xFF);
      const int d0 = (int)((wd >> (8 * j)) & 0xFF), d1 = (int)((wd >> (8 * (j + 1))) & 0xFF), d2 = (int)((wd >> (8 * (j + 2))) & 0xFF);
      if (s > n0 && s > n1 && s > u0 && s > u1 && s > u2 && s > d0 && s > d1 && s > d2) {
        const int slot = atomicAdd(&S.lcnt, 1);
        S.lst[slot] = ((uint32_t)s << 24) | ((uint32_t)y << 12) | (uint32_t)x;
      }
    }
  }
}

// The same suppression with the survivors left in ROW-MAJOR order (reference key-point order: FAST hands its corners over
// row by row, and k_select_cv rebuilds a level's row-major list from the tiles' ordered bursts).  Wave w owns tile rows
// 7w .. 7w+6 and walks them two rows (64 quads) a step, so its survivors come out in order from ballots alone -- no
// barrier; they go to the wave's quarter of S.lst (at most one survivor per 2x2 block: <= 256 per wave).  S.wtot receives
// the survivors per wave, S.rowcnt the survivors per tile row (a byte each).
__device__ __forceinline__ void fast_nms_collect_ordered(FastLds& S, const EvhLevel& L, int x0, int y0) {
  const bool live = (L.w > 2 * EVH_EDGE) && (L.h > 2 * EVH_EDGE);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  static_assert(FT_H == 28 && FT_W == 128, "four waves x seven rows of 32 quads");
  uint32_t* mine = S.lst + 256 * wv;
  int running = 0;
  uint32_t rc_lo = 0, rc_hi = 0;        // survivors of this wave's rows 0..3 / 4..6, a byte each (lane 0 keeps them)
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    const int local = 64 * k + lane;    // quad index inside the wave's 7 x 32 block
    uint32_t v0 = 0, v1 = 0;
    int cnt = 0;
    if (live && local < 7 * 32) {
      const int qr = 7 * wv + (local >> 5), qc = local & 31;
      const int y = y0 + qr, xq = x0 + qc * 4;
      const uint32_t* p = S.score + (qr + 1) * FS_DW + (qc + 1);
      const uint32_t m = p[0];
      if (m != 0 && y >= EVH_EDGE && y < L.h - EVH_EDGE) {
        // byte-parallel 3 x 3 maximum test (round 4; the per-pixel form cost ~130 instructions per quad, a quarter of the
        // kernel): the eight neighbour bytes of the quad's four pixels as eight dwords (two straight, six cut out of two words
        // with v_alignbyte), bit 7 of a byte of swar_ge(n, c) = neighbour >= centre, a survivor = a nonzero centre byte that no
        // neighbour reaches.  Strict maxima in a 3 x 3 window: at most two of four neighbouring pixels survive.
        const uint32_t H = 0x80808080u, Lm = 0x7F7F7F7Fu;
        const uint32_t lft = p[-1], rgt = p[1];
        const uint32_t um = p[-FS_DW], ul = p[-FS_DW - 1], ur = p[-FS_DW + 1];
        const uint32_t dm = p[FS_DW], dl = p[FS_DW - 1], dr = p[FS_DW + 1];
        const uint32_t nl = __builtin_amdgcn_alignbyte(m, lft, 3), nr = __builtin_amdgcn_alignbyte(rgt, m, 1);
        const uint32_t nul = __builtin_amdgcn_alignbyte(um, ul, 3), nur = __builtin_amdgcn_alignbyte(ur, um, 1);
        const uint32_t ndl = __builtin_amdgcn_alignbyte(dm, dl, 3), ndr = __builtin_amdgcn_alignbyte(dr, dm, 1);
        const uint32_t cL = m & Lm;
        const uint32_t g0 = swar_ge(nl | H, nl, m, cL), g1 = swar_ge(nr | H, nr, m, cL), g2 = swar_ge(um | H, um, m, cL);
        const uint32_t g3 = swar_ge(nul | H, nul, m, cL), g4 = swar_ge(nur | H, nur, m, cL), g5 = swar_ge(dm | H, dm, m, cL);
        const uint32_t g6 = swar_ge(ndl | H, ndl, m, cL), g7 = swar_ge(ndr | H, ndr, m, cL);
        const uint32_t ga = BITOP3(g0, g1, g2, A | B | C), gb = BITOP3(g3, g4, g5, A | B | C);
        const uint32_t any_ge = BITOP3(ga, gb, g6 | g7, A | B | C);
        const uint32_t nz = (cL + Lm) | m;                    // bit 7 of a byte: the centre byte is not zero
        uint32_t keep = BITOP3(nz, any_ge, H, A & ~B & C);
        {
          const int lo = min(max(EVH_EDGE - xq, 0), 4), hi = max(min(L.w - EVH_EDGE - xq, 4), 0);   // valid pixels j in [lo, hi)
          if (lo > 0 || hi < 4) keep &= lo < hi ? (H << (8 * lo)) & (H >> (8 * (4 - hi))) : 0u;
        }
        if (keep) {
          const int b0 = __ffs(keep) - 1;                     // bit 7, 15, 23 or 31 of the first survivor
          const int j0 = b0 >> 3;
          v0 = (((m >> (8 * j0)) & 0xFFu) << 24) | ((uint32_t)y << 12) | (uint32_t)(xq + j0);
          cnt = 1;
          const uint32_t rest = keep & (keep - 1u);
          if (rest) {
            const int j1 = (__ffs(rest) - 1) >> 3;
            v1 = (((m >> (8 * j1)) & 0xFFu) << 24) | ((uint32_t)y << 12) | (uint32_t)(xq + j1);
            cnt = 2;
          }
        }
      }
    }
    const unsigned long long m1 = __ballot(cnt >= 1), m2 = __ballot(cnt >= 2);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int pre = running + __popcll(m1 & lt) + __popcll(m2 & lt);
    if (cnt >= 1) mine[pre] = v0;
    if (cnt >= 2) mine[pre + 1] = v1;
    running += __popcll(m1) + __popcll(m2);
    const uint32_t ra = (uint32_t)(__popcll(m1 & 0xFFFFFFFFull) + __popcll(m2 & 0xFFFFFFFFull));
    const uint32_t rb = (uint32_t)(__popcll(m1 >> 32) + __popcll(m2 >> 32));
    if (k < 2) rc_lo |= (ra << (16 * k)) | (rb << (16 * k + 8));
    else rc_hi |= (ra << (16 * (k - 2))) | (rb << (16 * (k - 2) + 8));
  }
  if (lane == 0) {
    S.wtot[wv] = running;
    // tile row 7w + i -> byte (7w + i) & 3 of word (7w + i) >> 2; the rows of different waves share words: LDS atomics
    for (int i = 0; i < 7; i++) {
      const uint32_t c = i < 4 ? (rc_lo >> (8 * i)) & 0xFFu : (rc_hi >> (8 * (i - 4))) & 0xFFu;
      const int row = 7 * wv + i;
      if (c) atomicOr(&S.rowcnt[row >> 2], c << (8 * (row & 3)));
    }
  }
}

// ordered bursts: the four waves' quarters of S.lst one after another, then the tile's descriptor for k_select_cv
// (word 0 = offset of the burst in the level's candidate list, words 1..7 = survivors per tile row)
__device__ __forceinline__ void fast_emit_ordered(FastLds& S, const FastArgs& A, const EvhLevel& L, int f, int l, int tile) {
  __syncthreads();
  const int n0 = S.wtot[0], n1 = S.wtot[1], n2 = S.wtot[2], n = n0 + n1 + n2 + S.wtot[3];
  if (n > 0) {                                              // workgroup-uniform
    if (threadIdx.x == 0) S.gbase = atomicAdd(A.cand_count + f * EVH_NLEVELS + l, n);
    __syncthreads();
  }
  const int base = n > 0 ? S.gbase : 0;
  if (n > 0) {
    uint32_t* out = A.cand + (int64_t)f * A.cand_frame_entries + L.cand_off;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cnt = S.wtot[wv], off = wv == 0 ? 0 : wv == 1 ? n0 : wv == 2 ? n0 + n1 : n0 + n1 + n2;
    for (int i = lane; i < cnt; i += 64)
      if (base + off + i < L.cand_cap) out[base + off + i] = S.lst[256 * wv + i];
  }
  if (threadIdx.x < 8) {
    uint32_t* d = A.tdesc + ((int64_t)f * A.total_tiles + tile) * 8;
    d[threadIdx.x] = threadIdx.x == 0 ? (uint32_t)base : S.rowcnt[threadIdx.x - 1];
  }
}

// survivors -> the level's candidate list; ONE global atomic per workgroup reserves the slots (a returning global
// atomic per wave would serialise on its ~1-2 us latency)
__device__ __forceinline__ void fast_emit(FastLds& S, const FastArgs& A, const EvhLevel& L, int f, int l) {
  __syncthreads();
  const int n = S.lcnt;
  if (n > 0) {
    if (threadIdx.x == 0) S.gbase = atomicAdd(A.cand_count + f * EVH_NLEVELS + l, n);
    __syncthreads();
    uint32_t* out = A.cand + (int64_t)f * A.cand_frame_entries + L.cand_off;
    const int base = S.gbase;
    for (int i = threadIdx.x; i < n; i += 256)
      if (base + i < L.cand_cap) out[base + i] = S.lst[i];
  }
}

// the same by wave 0 alone (the other waves of the workgroup have left): wave-level ordering only, no barrier
__device__ __forceinline__ void fast_emit_wave0(FastLds& S, const FastArgs& A, const EvhLevel& L, int f, int l) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int n = S.lcnt;
  if (n > 0) {
    int base = 0;
    if (threadIdx.x == 0) base = atomicAdd(A.cand_count + f * EVH_NLEVELS + l, n);
    base = __builtin_amdgcn_readfirstlane(base);
    uint32_t* out = A.cand + (int64_t)f * A.cand_frame_entries + L.cand_off;
    for (int i = threadIdx.x; i < n; i += 64)
      if (base + i < L.cand_cap) out[base + i] = S.lst[i];
  }
}

__device__ __forceinline__ int fast_level_of_tile(const FastArgs& A, int& t) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < EVH_NLEVELS; i++)
    if (t >= A.lv[i].tile_start) l = i;
  t -= A.lv[l].tile_start;
  return l;
}

// K3 dense: every tile of every level at threshold 20
__global__ __launch_bounds__(256) void k_fast(FastArgs A) {
  __shared__ FastLds S;
  const int f = blockIdx.y;
  int t = blockIdx.x;
  const int l = fast_level_of_tile(A, t);
  const EvhLevel L = A.lv[l];
  const int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
  const int x0 = EVH_FAST_OX + tx * FT_W, y0 = EVH_FAST_OY + ty * FT_H;
  fast_stage(S, A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off, L, x0, y0);
  __syncthreads();
  fast_dense_scores(S, L, x0, y0);
  __syncthreads();
  if (A.tdesc) {                                   // reference key-point order (workgroup-uniform)
    fast_nms_collect_ordered(S, L, x0, y0);
    fast_emit_ordered(S, A, L, f, l, (int)blockIdx.x);
    return;
  }
  fast_nms_collect(S, L, x0, y0);
  fast_emit(S, A, L, f, l);
}

// ------------------------------------------------------------------------------------------------------------
// K3, threshold-lifted form.  ORB keeps only the 2*quota best-scoring FAST corners of a level (all ties at the
// cut), typically <1 % of the corners found at threshold 20.  A corner with score >= T is kept by NMS and by that
// selection exactly as before if every pixel with score < T is treated as "no corner": such neighbours cannot
// suppress it and cannot be selected.  So the exact score is only needed for pixels that can reach T.
//   k_fast_sample: threshold 20 on a sparse lattice of tiles (every 27th / 13th / 7th tile of a level), histogram
//                  of the NMS-surviving scores;
//   k_fast_thr:    T per (frame, level) such that ~4x the needed 2*quota corners are expected at or above it;
//   k_fast_main:   all tiles at T (pre-test + queued exact scores; the dense path where T stayed 20);
//   k_fast_verify: a lifted level that delivered fewer than 2*quota corners is reset ...
//   k_fast_redo:   ... and redone at threshold 20.  The result equals the dense kernel's by construction.
__global__ __launch_bounds__(256, 8) void k_fast_sample(FastArgs A) {
  __shared__ FastLds S;
  const int f = blockIdx.y;
  if (A.share_group > 0 && ((f % A.share_group) & 1)) return;   // shares the histogram of frame f - 1
  int s = blockIdx.x, l = 0;
#pragma unroll
  for (int i = 1; i < EVH_NLEVELS; i++)
    if (s >= A.samp_start[i]) l = i;
  s -= A.samp_start[l];
  const int mod = A.samp_mod[l];
  const EvhLevel L = A.lv[l];
  if (mod == 0) return;
  const int t = (f * 5 + l) % mod + s * mod;          // sampled tile index inside the level
  if (t >= L.tiles_x * L.tiles_y) return;
  const int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
  const int x0 = EVH_FAST_OX + tx * FT_W, y0 = EVH_FAST_OY + ty * FT_H;
  const int hint = A.hint_in[l];
  const int Tp = (hint > 36 && hint < 256) ? max(EVH_FAST_THR + 1, (hint * 7) >> 3) : 0;   // 0: dense sample
  fast_stage(S, A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off, L, x0, y0);
  __syncthreads();
  if (Tp) fast_lift_scores(S, L, x0, y0, Tp);          // exact scores >= Tp, zero elsewhere
  else fast_dense_scores(S, L, x0, y0);
  __syncthreads();
  if (Tp && S.scnt <= FSC_CAP) {               // workgroup-uniform: the short list of scored pixels is complete
    if (threadIdx.x < 64) fast_nms_scored(S, L, x0, y0);
  } else {
    fast_nms_collect(S, L, x0, y0);
  }
  __syncthreads();
  const int n = S.lcnt;
  unsigned* h = A.shist + (int64_t)(f * EVH_NLEVELS + l) * 256;
  if (threadIdx.x == 0) h[0] = (unsigned)Tp;            // bin 0 (never a score) carries the floor of this histogram
  // tile histogram in LDS first (the score plane is dead), then one global atomic per non-empty bin
  uint32_t* lh = S.score;
  lh[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) atomicAdd(&lh[S.lst[i] >> 24], 1u);
  __syncthreads();
  const uint32_t cnt = lh[threadIdx.x];
  if (cnt && threadIdx.x > EVH_FAST_THR) atomicAdd(&h[threadIdx.x], cnt);
}

__global__ void k_fast_thr(FastArgs A, int nframes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nframes * EVH_NLEVELS) return;
  const int l = i % EVH_NLEVELS;
  const EvhLevel L = A.lv[l];
  const int mod = A.samp_mod[l];
  int T = EVH_FAST_THR;
  if (mod > 0 && L.quota > 0) {
    const int need = max(24, (6 * L.quota + mod - 1) / mod);   // 3x the 2*quota corners the level must deliver (8: +3 % FAST time)
    const int f = i / EVH_NLEVELS;
    const int src = (A.share_group > 0 && ((f % A.share_group) & 1)) ? i - EVH_NLEVELS : i;
    const unsigned* h = A.shist + (int64_t)src * 256;
    const int floor_t = (int)h[0];                       // 0: dense histogram, else exact only from floor_t up
    int acc = 0;
    T = floor_t ? floor_t : EVH_FAST_THR;                // not enough mass above the floor: take all of it (verify decides)
    for (int s = 255; s > max(EVH_FAST_THR, floor_t - 1); s--) {
      acc += (int)h[s];
      if (acc >= need) { T = s; break; }
    }
    if (src == i) atomicAdd(&A.hint_hist[l * 256 + (T > EVH_FAST_THR ? min(T, 255) : 0)], 1u);   // vote for the next call's hint
  }
  A.thr[i] = min(T, 126);   // the byte-parallel pre-test needs T + 1 <= 127; any T in (20, score range] is exact
  if (i == 0) A.redo[0] = 0;                // work list of k_fast_redo: [0] = count, [1..] = frame * 8 + level
}

__global__ __launch_bounds__(256, 8) void k_fast_main(FastArgs A) {
  __shared__ FastLds S;
  const int f = blockIdx.y;      // (the XCD order of the gray / pyramid kernels measured no gain here: VALU-bound)
  int t = blockIdx.x;
  const int l = fast_level_of_tile(A, t);
  const EvhLevel L = A.lv[l];
  const int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
  const int x0 = EVH_FAST_OX + tx * FT_W, y0 = EVH_FAST_OY + ty * FT_H;
  const int T = A.lift_base ? EVH_FAST_THR : A.thr[f * EVH_NLEVELS + l];
  fast_stage(S, A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off, L, x0, y0);
  __syncthreads();
  if (A.lift_base) {
    // reference key-point order: every corner at the base threshold, found by the full segment test, scored exactly, handed
    // over as a row-major burst (fast_nms_collect_ordered walks the score plane, which is complete: zero where no corner is)
    fast_lift_scores<true>(S, L, x0, y0, EVH_FAST_THR);
    __syncthreads();
    fast_nms_collect_ordered(S, L, x0, y0);
    fast_emit_ordered(S, A, L, f, l, (int)blockIdx.x);
    return;
  }
  if (T > EVH_FAST_THR) {
    fast_lift_scores(S, L, x0, y0, T);
    __syncthreads();
    if (S.scnt <= FSC_CAP) {                      // workgroup-uniform.  What is left is a few dozen scored pixels:
      if (threadIdx.x >= 64) return;              // waves 1-3 are done (no barrier follows on this path), wave 0
      fast_nms_scored(S, L, x0, y0);              // runs NMS and the emission on its own
      fast_emit_wave0(S, A, L, f, l);
      return;
    }
    fast_nms_collect(S, L, x0, y0);               // the short list overflowed: the score plane itself is complete
  } else {
    fast_dense_scores(S, L, x0, y0);
    __syncthreads();
    fast_nms_collect(S, L, x0, y0);
  }
  fast_emit(S, A, L, f, l);
}

__global__ void k_fast_verify(FastArgs A, int nframes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nframes * EVH_NLEVELS) return;
  const int l = i % EVH_NLEVELS;
  if (A.thr[i] > EVH_FAST_THR && A.cand_count[i] < 2 * A.lv[l].quota) {
    A.cand_count[i] = 0;
    A.thr[i] = EVH_FAST_THR;
    A.redo[1 + atomicAdd(&A.redo[0], 1)] = i;
    atomicAdd(&A.hint_hist[l * 256], 1u);               // a vote for 'no hint': the level came up short
  }
}

// next call's hint per level: the lower quartile of this call's votes (robust against a few odd frames; a frame whose
// own threshold lies below 7/8 of it merely takes everything above that floor, and verify/redo stays the safety net)
__global__ void k_fast_hint(FastArgs A) {
  const int l = threadIdx.x;
  if (l >= EVH_NLEVELS) return;
  const unsigned* h = A.hint_hist + l * 256;
  unsigned total = 0;
  for (int s = 0; s < 256; s++) total += h[s];
  int hint = 0;
  if (total) {
    unsigned acc = 0;
    for (int s = 0; s < 256; s++) { acc += h[s]; if (4 * acc >= total) { hint = s; break; } }
  }
  A.hint_out[l] = hint;
}

// dense rescoring of the (frame, level) entries k_fast_verify listed: blockIdx.x = tile of the level, blockIdx.y
// walks the list, so a flagged level is redone by all its tiles in parallel; with an empty list every workgroup leaves
// at once
__global__ __launch_bounds__(256) void k_fast_redo(FastArgs A) {
  __shared__ FastLds S;
  const int count = A.redo[0];
  for (int e = blockIdx.y; e < count; e += gridDim.y) {      // workgroup-uniform bounds
    const int i = A.redo[1 + e];
    const int f = i / EVH_NLEVELS, l = i - f * EVH_NLEVELS;
    const EvhLevel L = A.lv[l];
    const int t = blockIdx.x;
    if (t < L.tiles_x * L.tiles_y) {
      const int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
      const int x0 = EVH_FAST_OX + tx * FT_W, y0 = EVH_FAST_OY + ty * FT_H;
      fast_stage(S, A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off, L, x0, y0);
      __syncthreads();
      fast_dense_scores(S, L, x0, y0);
      __syncthreads();
      fast_nms_collect(S, L, x0, y0);
      fast_emit(S, A, L, f, l);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// K4: per frame, per level: retainBest(2*quota) by FAST score (all ties with the cut kept), Harris response,
// retainBest(quota) by Harris (ties kept), canonical order (y, x); writes keypoint records.
struct SelectArgs {
  EvhLevel lv[EVH_NLEVELS];
  const uint8_t* pyr; int64_t pyr_frame_bytes;
  const uint32_t* cand; int64_t cand_frame_entries;
  const int* cand_count;
  float* kp_xy; uint32_t* kp_meta; float* kp_resp; int* kp_count; int* frame_flags;
  uint32_t* tmp_meta; float* tmp_resp; int* lvl_count;   // per-level staging (segments at EvhLevel.kp_base)
  int kcap;
  int k1cap, k2cap;   // LDS capacities of k_select (stage-1 / stage-2 survivors of one level)
  int nframes;        // real frame count (the grid of k_select is padded, xcd_grid)
};

__device__ __forceinline__ float harris_response(const uint8_t* img, int stride, int x0, int y0) {
  // the 9 x 9 window (x0-4 .. x0+4, y0-4 .. y0+4) as three aligned dwords per row, realigned in registers:
  // 27 dword loads per key point instead of ~190 scattered byte loads (the texture-address path was the limit).
  // Integer sums are order-independent, the float tail below is unchanged.
  const int xa = (x0 - 4) & ~3;
  const uint32_t sh = (uint32_t)((x0 - 4) - xa);
  const uint32_t* base = reinterpret_cast<const uint32_t*>(img + (int64_t)(y0 - 4) * stride + xa);
  const int sd = stride >> 2;
  uint32_t w0[9], w1[9], w2[9];
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const uint32_t d0 = base[r * sd], d1 = base[r * sd + 1], d2 = base[r * sd + 2];
    w0[r] = __builtin_amdgcn_alignbyte(d1, d0, sh);      // bytes x0-4 .. x0-1
    w1[r] = __builtin_amdgcn_alignbyte(d2, d1, sh);      // bytes x0 .. x0+3
    w2[r] = d2 >> (8 * sh);                              // byte x0+4 in bits 0..7
  }
#define HB(r, c) ((c) < 4 ? (int)((w0[r] >> (8 * (c))) & 0xFFu) : (c) < 8 ? (int)((w1[r] >> (8 * ((c) - 4))) & 0xFFu) : (int)(w2[r] & 0xFFu))
  int a = 0, b = 0, c = 0;
#pragma unroll
  for (int i = 1; i <= 7; i++) {          // window row i = y0 - 4 + i
#pragma unroll
    for (int j = 1; j <= 7; j++) {        // window column j = x0 - 4 + j
      const int Ix = (HB(i, j + 1) - HB(i, j - 1)) * 2 + (HB(i - 1, j + 1) - HB(i - 1, j - 1)) + (HB(i + 1, j + 1) - HB(i + 1, j - 1));
      const int Iy = (HB(i + 1, j) - HB(i - 1, j)) * 2 + (HB(i + 1, j - 1) - HB(i - 1, j - 1)) + (HB(i + 1, j + 1) - HB(i - 1, j + 1));
      a = mad24s(Ix, Ix, a); b = mad24s(Iy, Iy, b); c = mad24s(Ix, Iy, c);   // |Ix|, |Iy| <= 1020
    }
  }
#undef HB
  const float scale = 1.f / (4 * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  float fa = (float)a, fb = (float)b, fc = (float)c;
  float t1 = fa * fb;
  float t2 = fc * fc;
  float s = fa + fb;
  float t3 = (0.04f * s) * s;
  return ((t1 - t2) - t3) * scale_sq_sq;
}

__device__ __forceinline__ uint32_t f32_order_key(float v) {
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// All 256 threads: the largest bin d whose inclusive suffix sum (bins d..255) reaches `target`, and the sum of the bins
// above d.  Equals the serial scan "from 255 down, stop at the first bin where the running sum reaches target".
// hist must be complete (barrier before the call); the caller guarantees that the total reaches target.  sh: int[12].
__device__ __forceinline__ int block_suffix_cut(const uint32_t* hist, int target, int* sh, int& above) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int v = (int)hist[tid];
  int s = v;                                    // inclusive suffix sum inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(s, o); if (lane + o < 64) s += t; }
  if (lane == 0) sh[wv] = s;
  __syncthreads();
  int hi = 0;
  for (int w = wv + 1; w < 4; w++) hi += sh[w];
  const int suf = s + hi;
  const unsigned long long m = __ballot(suf >= target);
  if (lane == 0) sh[4 + wv] = m ? (wv * 64 + 63 - (int)__clzll(m)) : -1;
  __syncthreads();
  const int d = max(max(sh[4], sh[5]), max(max(sh[6], sh[7]), 0));
  if (tid == d) sh[8] = suf - v;
  __syncthreads();
  above = sh[8];
  return d;
}

// one workgroup per (level, frame): both retainBest stages + Harris + canonical order; results go to the level's
// segment of the frame's staging arrays, k_pack then concatenates the eight segments.
__global__ __launch_bounds__(256, 6) void k_select(SelectArgs A) {   // 75 VGPRs: 6 instead of 4 waves per SIMD, 0.42 -> 0.39 ms
  // dynamic LDS, sized by the launcher from the key-point budget: keys[k1cap] | resp[k1cap] | sel[k2cap] | selr[k2cap]
  extern __shared__ uint32_t sel_dyn[];
  const int K1CAP = A.k1cap, K2CAP = A.k2cap;
  uint32_t* keys = sel_dyn;
  float* resp = reinterpret_cast<float*>(keys + K1CAP);
  uint32_t* sel = reinterpret_cast<uint32_t*>(resp + K1CAP);
  float* selr = reinterpret_cast<float*>(sel + K2CAP);
  __shared__ uint32_t hist[256];
  __shared__ int sh_i[8];  // 1: k1, 2: k2
  __shared__ int sh_cut[12];
  int l, f;                      // the eight levels of a frame on one XCD: 0.54 -> 0.45 ms
  xcd_order(l, f);
  if (f >= A.nframes) return;    // grid padding (workgroup-uniform)
  const int tid = threadIdx.x;
  const EvhLevel L = A.lv[l];
  const uint32_t* cand = A.cand + (int64_t)f * A.cand_frame_entries + L.cand_off;
  const int n_raw = A.cand_count[f * EVH_NLEVELS + l];
  bool overflow = n_raw > L.cand_cap;
  const int n = min(n_raw, L.cand_cap);
  const int q = L.quota;
  int k2 = 0;
  if (n > 0 && q > 0) {
    // ---- stage 1: cut on the integer FAST score through a 256-bin histogram
    hist[tid] = 0;
    if (tid < 8) sh_i[tid] = 0;
    __syncthreads();
    if (n > 2 * q)
      for (int i = tid; i < n; i += 256) atomicAdd(&hist[cand[i] >> 24], 1u);
    __syncthreads();
    uint32_t cut = 0;
    if (n > 2 * q) { int above; cut = (uint32_t)block_suffix_cut(hist, 2 * q, sh_cut, above); }   // workgroup-uniform branch
    for (int i = tid; i < n; i += 256) {
      uint32_t c = cand[i];
      if ((c >> 24) >= cut) {
        int slot = atomicAdd(&sh_i[1], 1);
        if (slot < K1CAP) keys[slot] = c;
      }
    }
    __syncthreads();
    const int k1 = sh_i[1];
    // retainBest keeps EVERY tie at the cut, so k1 has no bound but n (saturated / binary content ties massively on the
    // integer score).  More survivors than the LDS list holds: spill path -- nothing is stored, the survivors are
    // re-read from the candidate list and their Harris responses recomputed in each pass (same values, same cut).
    const bool spill = k1 > K1CAP;                       // workgroup-uniform
    const uint8_t* img = A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off;
    // ---- Harris response of every stage-1 survivor
    if (!spill) {
      for (int j = tid; j < k1; j += 256) {
        uint32_t c = keys[j];
        resp[j] = harris_response(img, L.stride, (int)(c & 0xFFFu), (int)((c >> 12) & 0xFFFu));
      }
    }
    __syncthreads();
    // ---- stage 2: value of the q-th largest response by a 4 x 8-bit radix select on order-preserving keys
    float cutf = -INFINITY;
    if (k1 > q) {
      uint32_t prefix = 0;
      int rem = q;
      for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        if (!spill) {
          for (int j = tid; j < k1; j += 256) {
            uint32_t u = f32_order_key(resp[j]);
            bool in = pass == 0 ? true : ((u >> (shift + 8)) == (prefix >> (shift + 8)));
            if (in) atomicAdd(&hist[(u >> shift) & 0xFFu], 1u);
          }
        } else {
          for (int i = tid; i < n; i += 256) {
            const uint32_t c = cand[i];
            if ((c >> 24) < cut) continue;
            uint32_t u = f32_order_key(harris_response(img, L.stride, (int)(c & 0xFFFu), (int)((c >> 12) & 0xFFFu)));
            bool in = pass == 0 ? true : ((u >> (shift + 8)) == (prefix >> (shift + 8)));
            if (in) atomicAdd(&hist[(u >> shift) & 0xFFu], 1u);
          }
        }
        __syncthreads();
        int above;
        const int d = block_suffix_cut(hist, rem, sh_cut, above);
        rem -= above;
        prefix |= (uint32_t)d << shift;
      }
      uint32_t u = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
      cutf = __uint_as_float(u);
    }
    if (!spill) {
      for (int j = tid; j < k1; j += 256)
        if (resp[j] >= cutf) {
          int slot = atomicAdd(&sh_i[2], 1);
          if (slot < K2CAP) { sel[slot] = keys[j]; selr[slot] = resp[j]; }
        }
    } else {
      for (int i = tid; i < n; i += 256) {
        const uint32_t c = cand[i];
        if ((c >> 24) < cut) continue;
        const float r = harris_response(img, L.stride, (int)(c & 0xFFFu), (int)((c >> 12) & 0xFFFu));
        if (r >= cutf) {
          int slot = atomicAdd(&sh_i[2], 1);
          if (slot < K2CAP) { sel[slot] = c; selr[slot] = r; }
        }
      }
    }
    __syncthreads();
    k2 = sh_i[2];
    // the one hard bound left: a level cannot deliver more key points than a frame's slot holds (K2CAP == kcap);
    // such a frame is flagged (EVH_PAIR_CAPACITY for its pairs), never truncated silently
    if (k2 > K2CAP) { overflow = true; k2 = K2CAP; }
    // ---- canonical order inside the level: ascending (y, x) by rank counting
    for (int j = tid; j < k2; j += 256) {
      uint32_t kj = sel[j] & 0xFFFFFFu;
      int pos = 0;
      for (int i = 0; i < k2; i++) pos += ((sel[i] & 0xFFFFFFu) < kj) ? 1 : 0;
      int64_t o = ((int64_t)f * EVH_NLEVELS + l) * A.kcap + pos;
      A.tmp_meta[o] = ((uint32_t)l << 24) | kj;
      A.tmp_resp[o] = selr[j];
    }
  }
  if (tid == 0) {
    A.lvl_count[f * EVH_NLEVELS + l] = k2;
    if (overflow) atomicOr(&A.frame_flags[f], 1);
  }
}

// concatenates the per-level segments of one frame (canonical order = level, y, x) and derives kp.pt; a frame whose
// levels together hold more than kcap key points is flagged (its slot keeps the first kcap)
__global__ __launch_bounds__(256) void k_pack(SelectArgs A) {
  const int f = blockIdx.x, tid = threadIdx.x;
  int base = 0;
  for (int l = 0; l < EVH_NLEVELS; l++) {
    const EvhLevel L = A.lv[l];
    const int n = min(A.lvl_count[f * EVH_NLEVELS + l], A.kcap - base);
    if (n < A.lvl_count[f * EVH_NLEVELS + l] && tid == 0) atomicOr(&A.frame_flags[f], 1);
    for (int j = tid; j < n; j += 256) {
      const int64_t si = ((int64_t)f * EVH_NLEVELS + l) * A.kcap + j, o = (int64_t)f * A.kcap + base + j;
      const uint32_t m = A.tmp_meta[si];
      A.kp_meta[o] = m;
      A.kp_resp[o] = A.tmp_resp[si];
      A.kp_xy[2 * o] = (float)(int)(m & 0xFFFu) * L.scale;          // keypoint.pt *= layerScale
      A.kp_xy[2 * o + 1] = (float)(int)((m >> 12) & 0xFFFu) * L.scale;
    }
    base += n;
  }
  if (tid == 0) A.kp_count[f] = base;
}

// ------------------------------------------------------------------------------------------------------------
// K4, reference order (EVH_ORDER_OPENCV, the default).  KeyPointsFilter::retainBest (features2d/src/keypoint.cpp) is
//     std::nth_element(begin, begin + n, end, response-greater); amb = kp[n - 1].response;
//     new_end = std::partition(begin + n, end, response >= amb); resize(new_end - begin)
// and both calls PERMUTE the vector: the order ORB hands its key points over in -- hence the order of the matches, of the
// rows given to RANSAC, hence which minimal samples its random generator draws -- is the order libstdc++'s introselect and
// partition leave behind, and with ties at the cut even the surviving SET depends on it (position n - 1 holds an arbitrary
// member of the best n).  The reference's own recorded run agrees with this order and with no other
// (tests/test_capture_golden.py), so the order is part of the operator.  k_select_cv runs the same algorithms on the same
// sequence (FAST corners of a level in row-major order), with every pass over the data done by the whole workgroup:
//   * Hoare's unguarded partition = pair the k-th element from the left that is not "before" the pivot with the k-th from
//     the right that is not "after" it while the former lies left of the latter; the pairs are independent, so the two
//     stopper lists are built by a scan, the number of pairs by a search, the swaps in parallel; the cut follows from the
//     first unpaired stoppers.  std::partition is the same with a predicate.
//   * the row-major sequence comes from a bit plane of the corners: rank = set bits before the corner.
// What stays sequential is what libstdc++ does per round in O(1): the median-of-three pivot and the final insertion sort.
struct SelCvArgs {
  SelectArgs s;
  unsigned long long* seq;   // [nframes][cand_frame_entries]  stage 2: Harris key << 32 | packed candidate
  uint32_t* seq32;           // [nframes][cand_frame_entries]  stage 1: the candidates themselves (key = FAST score = top byte)
  uint32_t* lpos;            // [nframes][cand_frame_entries]  left-stopper positions, ascending
  uint32_t* rpos;            // [nframes][cand_frame_entries]  right-stopper positions, ascending
  uint32_t* mask;            // [nframes][2 * mask_frame_words] corner bit plane, then its running popcount
  int64_t mask_frame_words;
  int mask_off[EVH_NLEVELS];
  const uint32_t* tdesc;     // [nframes][total_tiles][8] tile burst descriptors written by k_fast
  int total_tiles;
  int heap_cap;              // entries of the dynamic LDS heap (>= 2 * largest quota + 1)
  int level0, nlev;          // this launch handles levels level0 .. level0 + nlev - 1
  int phase_limit;           // profiling aid: 1 = stop after the row-major sequence, 2 = after the first retainBest, 0 = all
};

struct CvLds {
  int wsumL[16], wsumR[16];   // up to 16 waves per workgroup (k_select_cv runs with 256 or 1024 threads)
  int bc[16];
};

#define CV_NOPOS 0x7FFFFFFF

// exclusive prefix of (a, b) over the threads of the workgroup (blockDim.x = 64 * NW); totals come back in ta / tb.  Two barriers.
__device__ __forceinline__ void cv_scan2(CvLds& S, int a, int b, int& ea, int& eb, int& ta, int& tb) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ua = __shfl_up(ia, o), ub = __shfl_up(ib, o);
    if (lane >= o) { ia += ua; ib += ub; }
  }
  if (lane == 63) { S.wsumL[wv] = ia; S.wsumR[wv] = ib; }
  __syncthreads();
  int ba = 0, bb = 0;
  ta = 0; tb = 0;
  const int nw = (int)blockDim.x >> 6;
  for (int w = 0; w < nw; w++) {
    const int l_ = S.wsumL[w], r_ = S.wsumR[w];
    if (w < wv) { ba += l_; bb += r_; }
    ta += l_; tb += r_;
  }
  ea = ba + ia - a;
  eb = bb + ib - b;
  __syncthreads();
}

// Partition pass over a[lo, hi).  MODE 0: Hoare around the pivot key p (left stoppers key <= p, right stoppers key >= p),
// returns the cut.  MODE 1: std::partition with the predicate key >= p (left stoppers !pred, right stoppers pred), returns
// the position of the first element of the false group.  EP / LP: element and position-list pointers (global memory with
// 32-bit positions, or the LDS copy of a short range with 16-bit positions).
// element = key << 32 | candidate (64-bit: the Harris stage) or the 32-bit candidate itself, whose top byte is the FAST score
__device__ __forceinline__ uint32_t cv_key(unsigned long long e) { return (uint32_t)(e >> 32); }
__device__ __forceinline__ uint32_t cv_key(uint32_t e) { return e >> 24; }
template <class E>
__device__ __forceinline__ bool cv_gt(E x, E y) { return cv_key(x) > cv_key(y); }

template <int MODE, class EP, class LP>
__device__ int cv_partition(EP a, int lo, int hi, uint32_t p, LP lpos, LP rpos, CvLds& S) {
  typedef typename std::remove_pointer<LP>::type PT;
  typedef typename std::remove_pointer<EP>::type E;
  const int tid = threadIdx.x, NT = (int)blockDim.x, NW = NT >> 6;
  constexpr int EPT = 4;      // elements per thread and step (8 for the 32-bit elements measured slower: 5.5 against 4.9 ms)
  int cntL = 0, cntR = 0;
  for (int base = lo; base < hi; base += EPT * NT) {
    const int i0 = base + EPT * tid;
    uint32_t fl = 0, fr = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      if (i0 + e < hi) {
        const uint32_t key = cv_key(a[i0 + e]);
        const bool le = MODE == 0 ? key <= p : key < p;
        const bool ge = key >= p;
        fl |= (le ? 1u : 0u) << e;
        fr |= (ge ? 1u : 0u) << e;
      }
    }
    int el, er, tl, tr;
    cv_scan2(S, __popc(fl), __popc(fr), el, er, tl, tr);
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      if ((fl >> e) & 1u) lpos[cntL + el++] = (PT)(i0 + e);
      if ((fr >> e) & 1u) rpos[cntR + er++] = (PT)(i0 + e);
    }
    cntL += tl;
    cntR += tr;
  }
  __syncthreads();   // lists complete
  // number of pairs: the largest m with L[k] < R[k] for all k < m (monotone), by NT-way search; R[k] = rpos[cntR - 1 - k]
  const int K = min(cntL, cntR);
  int lo_k = 0, hi_k = K;   // invariant: pairs [0, lo_k) swap, pairs [hi_k, K) do not
  while (hi_k > lo_k) {
    const int span = hi_k - lo_k;
    const int step = (span + NT - 1) / NT;
    const int k = lo_k + tid * step;
    const bool ok = k < hi_k && (int)lpos[k] < (int)rpos[cntR - 1 - k];
    const unsigned long long bal = __ballot(ok);
    if ((tid & 63) == 0) S.bc[tid >> 6] = __popcll(bal);
    __syncthreads();
    int good = 0;                                             // probes are monotone: the first `good` probes hold
    for (int w = 0; w < NW; w++) good += S.bc[w];
    __syncthreads();
    if (good == 0) { hi_k = lo_k; break; }
    const int last_good = lo_k + (good - 1) * step;
    lo_k = last_good + 1;
    hi_k = min(hi_k, last_good + step);
  }
  const int m = lo_k;
  for (int k = tid; k < m; k += NT) {
    const int i = (int)lpos[k], j = (int)rpos[cntR - 1 - k];
    const E t = a[i];
    a[i] = a[j];
    a[j] = t;
  }
  int ret;
  if (MODE == 0) {
    const int Lm = m < cntL ? (int)lpos[m] : CV_NOPOS;
    const int Rm1 = m > 0 ? (int)rpos[cntR - m] : CV_NOPOS;
    ret = min(Lm, Rm1);
  } else {
    ret = lo + cntR;
  }
  __syncthreads();   // swaps visible, lists free
  return ret;
}

// ---- libstdc++ heap primitives on an LDS array (one thread): __adjust_heap (with its __push_heap tail), __make_heap
template <class E>
__device__ void cv_adjust_heap(E* hp, int hole, int len, E value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (cv_gt(hp[child], hp[child - 1])) child--;
    hp[hole] = hp[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    hp[hole] = hp[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && cv_gt(hp[parent], value)) {
    hp[hole] = hp[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  hp[hole] = value;
}

// std::__heap_select(a + first, a + middle, a + last, greater-by-key), introselect's fall-back when its depth limit is
// reached: the heap [first, middle) lives in LDS while the tail is scanned; the scan is the workgroup's (256 elements per
// step, the next element that beats the heap's top found by ballot), the heap operations are one thread's.
template <class EP>
__device__ void cv_heap_select(EP a, int first, int middle, int last, typename std::remove_pointer<EP>::type* hp, CvLds& S) {
  typedef typename std::remove_pointer<EP>::type E;
  const int tid = threadIdx.x, len = middle - first, NT = (int)blockDim.x, NW = NT >> 6;
  for (int i = tid; i < len; i += NT) hp[i] = a[first + i];
  __syncthreads();
  if (tid == 0 && len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      const E value = hp[parent];
      cv_adjust_heap(hp, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  __syncthreads();
  for (int base = middle; base < last; base += NT) {
    const int idx = base + tid;
    const E mine = idx < last ? a[idx] : (E)0;
    int done = base;   // elements of this chunk below `done` have been handled
    for (;;) {
      const E top = hp[0];
      const bool hit = idx < last && idx >= done && cv_gt(mine, top);
      const unsigned long long bal = __ballot(hit);
      if ((tid & 63) == 0) S.bc[tid >> 6] = bal ? (int)(tid + __ffsll((long long)bal) - 1) : 1 << 20;
      __syncthreads();
      int j = 1 << 20;                                                   // thread index of the first hit
      for (int w = 0; w < NW; w++) j = min(j, S.bc[w]);
      __syncthreads();
      if (j >= NT) break;
      if (tid == j) {
        // __pop_heap(first, middle, result = a + idx)
        a[idx] = top;
        cv_adjust_heap(hp, 0, len, mine);
      }
      done = base + j + 1;
      __syncthreads();
    }
  }
  for (int i = tid; i < len; i += NT) a[first + i] = hp[i];
  __syncthreads();
}

// a range of at most CV_SMALL elements is worked on in LDS: a round then costs LDS latencies instead of a chain of
// dependent global accesses (pivot, cut, lists), which is what the small pyramid levels and the last rounds of the large
// ones consist of
#ifndef CV_SMALL
#define CV_SMALL 2048
#endif
#ifndef CV_MINW
#define CV_MINW 1
#endif
#ifndef CV_MAXT
#define CV_MAXT 1024
#endif
struct CvSmall {
  unsigned long long a[CV_SMALL];
  uint16_t l[CV_SMALL], r[CV_SMALL];
};

// libstdc++ __introselect on a[first, last) with `depth` rounds left; false = the heap of the depth-limit fall-back does
// not fit the LDS array (cannot happen for nth <= 2 * quota: the launcher sizes it so)
template <class EP, class LP>
__device__ bool cv_introselect_loop(EP a, int first, int nth, int last, int depth, LP lpos, LP rpos, CvLds& S,
                                    unsigned long long* hp_raw, int hp_cap, CvSmall* sm) {
  typedef typename std::remove_pointer<EP>::type E;
  E* hp = reinterpret_cast<E*>(hp_raw);
  while (last - first > 3) {
    if constexpr (std::is_same<LP, uint32_t*>::value) if (sm && last - first <= CV_SMALL) {   // (the LDS instantiation never stages)
      const int len = last - first;
      E* la = reinterpret_cast<E*>(sm->a);
      for (int i = threadIdx.x; i < len; i += (int)blockDim.x) la[i] = a[first + i];
      __syncthreads();
      const bool ok = cv_introselect_loop<E*, uint16_t*>(la, 0, nth - first, len, depth, sm->l, sm->r, S, hp_raw, hp_cap, nullptr);
      for (int i = threadIdx.x; i < len; i += (int)blockDim.x) a[first + i] = la[i];
      __syncthreads();
      return ok;
    }
    if (depth == 0) {
      if (nth + 1 - first > hp_cap) return false;
      cv_heap_select(a, first, nth + 1, last, hp, S);
      if (threadIdx.x == 0) {
        const E t = a[first];
        a[first] = a[nth];
        a[nth] = t;
      }
      __syncthreads();
      return true;
    }
    --depth;
    if (threadIdx.x == 0) {
      // __move_median_to_first(first, first + 1, mid, last - 1)
      const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
      const E va = a[ia], vb = a[ib], vc = a[ic];
      int pick;
      if (cv_gt(va, vb)) pick = cv_gt(vb, vc) ? ib : (cv_gt(va, vc) ? ic : ia);
      else pick = cv_gt(va, vc) ? ia : (cv_gt(vb, vc) ? ic : ib);
      const E t = a[first];
      a[first] = a[pick];
      a[pick] = t;
    }
    __syncthreads();
    const uint32_t p = cv_key(a[first]);
    const int cut = cv_partition<0>(a, first + 1, last, p, lpos, rpos, S);
    if (cut <= nth) first = cut; else last = cut;
  }
  if (threadIdx.x == 0) {
    // __insertion_sort(first, last)
    for (int i = first + 1; i < last; i++) {
      const E val = a[i];
      if (cv_gt(val, a[first])) {
        for (int j = i; j > first; j--) a[j] = a[j - 1];
        a[first] = val;
      } else {
        int j = i;
        while (cv_gt(val, a[j - 1])) { a[j] = a[j - 1]; --j; }
        a[j] = val;
      }
    }
  }
  __syncthreads();
  return true;
}

// KeyPointsFilter::retainBest on a[0, n): std::nth_element(a, a + npoints, a + n), then std::partition of the tail by
// "response >= a[npoints - 1].response".  Returns the new size, -1 when the fall-back heap does not fit.
template <class E>
__device__ int cv_retain_best(E* a, int n, int npoints, uint32_t* lpos, uint32_t* rpos, CvLds& S,
                              unsigned long long* hp, int hp_cap, CvSmall* sm) {
  if (npoints < 0 || n <= npoints) return n;
  if (npoints == 0) return 0;
  const int depth = 2 * (31 - __clz(n));
  if (n <= CV_SMALL) {                       // everything in LDS, the survivors copied back
    E* la = reinterpret_cast<E*>(sm->a);
    for (int i = threadIdx.x; i < n; i += (int)blockDim.x) la[i] = a[i];
    __syncthreads();
    if (!cv_introselect_loop<E*, uint16_t*>(la, 0, npoints, n, depth, sm->l, sm->r, S, hp, hp_cap, nullptr)) return -1;
    const uint32_t amb = cv_key(la[npoints - 1]);
    const int k = cv_partition<1>(la, npoints, n, amb, sm->l, sm->r, S);
    for (int i = threadIdx.x; i < k; i += (int)blockDim.x) a[i] = la[i];
    __syncthreads();
    return k;
  }
  if (!cv_introselect_loop<E*, uint32_t*>(a, 0, npoints, n, depth, lpos, rpos, S, hp, hp_cap, sm)) return -1;
  const uint32_t amb = cv_key(a[npoints - 1]);
  return cv_partition<1>(a, npoints, n, amb, lpos, rpos, S);
}

__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Launched twice: the large pyramid levels with 1024 threads per workgroup (a partition pass is a chain of steps, each a
// load + a workgroup scan + stores: four times the threads = a quarter of the steps), the small ones with 256.
__global__ __launch_bounds__(CV_MAXT, CV_MINW) void k_select_cv(SelCvArgs B) {
  const SelectArgs& A = B.s;
  __shared__ CvLds S;
  __shared__ CvSmall SM;
  extern __shared__ unsigned long long cv_heap[];   // 2 * quota(level 0) + 2 entries
  int l, f;
  xcd_order(l, f);
  if (f >= A.nframes || l >= B.nlev) return;
  l += B.level0;
  const int tid = threadIdx.x, NT = (int)blockDim.x, NW = NT >> 6;
  const EvhLevel L = A.lv[l];
  const uint32_t* cand = A.cand + (int64_t)f * A.cand_frame_entries + L.cand_off;
  const int n_raw = A.cand_count[f * EVH_NLEVELS + l];
  bool overflow = n_raw > L.cand_cap;
  const int n = min(n_raw, L.cand_cap);
  const int q = L.quota;
  unsigned long long* a = B.seq + (int64_t)f * A.cand_frame_entries + L.cand_off;
  uint32_t* a32 = B.seq32 + (int64_t)f * A.cand_frame_entries + L.cand_off;
  uint32_t* lpos = B.lpos + (int64_t)f * A.cand_frame_entries + L.cand_off;
  uint32_t* rpos = B.rpos + (int64_t)f * A.cand_frame_entries + L.cand_off;
  int k2 = 0;
  bool unsupported = false;
  if (n > 0 && q > 0) {
    // ---- the corners of the level in row-major order (as cv::FAST hands them over).  Every FAST tile left its corners as one
    // row-major burst in the level's list, with a descriptor (offset, corners per tile row): the place of a corner is
    // (corners in earlier rows of the level) + (corners of its row in tiles to the left) + (its rank in its tile's row).
    const int TX = L.tiles_x, TY = L.tiles_y, NE = TY * FT_H * TX;
    uint32_t* P = B.mask + (int64_t)f * 2 * B.mask_frame_words + B.mask_off[l];   // exclusive prefix in (tile row, row, tile column) order
    const uint32_t* td = B.tdesc + ((int64_t)f * B.total_tiles + L.tile_start) * 8;
    {
      const int per = (NE + NT - 1) / NT, e0 = tid * per, e1 = min(NE, e0 + per);
      int sum = 0;
      for (int e = e0; e < e1; e++) {
        const int tx = e % TX, rr = e / TX, r = rr % FT_H, ty = rr / FT_H;
        sum += (int)((td[(ty * TX + tx) * 8 + 1 + (r >> 2)] >> (8 * (r & 3))) & 0xFFu);
      }
      int ex, d0, tot, d1;
      cv_scan2(S, sum, 0, ex, d0, tot, d1);
      for (int e = e0; e < e1; e++) {
        const int tx = e % TX, rr = e / TX, r = rr % FT_H, ty = rr / FT_H;
        P[e] = (uint32_t)ex;
        ex += (int)((td[(ty * TX + tx) * 8 + 1 + (r >> 2)] >> (8 * (r & 3))) & 0xFFu);
      }
      if (tot != n) overflow = true;    // cannot happen: the descriptors and the list come from the same tiles
    }
    __syncthreads();
    {
      const int lane = tid & 63, wv = tid >> 6;
      for (int t = wv; t < TX * TY; t += NW) {
        const int ty = t / TX, tx = t - ty * TX;
        const uint32_t wd = lane < 8 ? td[t * 8 + lane] : 0u;
        const int base = (int)__shfl(wd, 0);
        // corners per tile row in lanes 0..27, their exclusive prefix = first burst index of the row
        const uint32_t cw = __shfl(wd, 1 + (min(lane, FT_H - 1) >> 2));
        const int c = lane < FT_H ? (int)((cw >> (8 * (lane & 3))) & 0xFFu) : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        const int start = inc - c;
        const int total = __shfl(inc, FT_H - 1);
        for (int j = lane; j < ((total + 63) & ~63); j += 64) {
          uint32_t cnd = 0;
          int r = 0;
          if (j < total) {
            cnd = cand[base + j];
            r = (int)((cnd >> 12) & 0xFFFu) - (EVH_FAST_OY + ty * FT_H);
          }
          const int st = __shfl(start, r);
          if (j < total) {
            const int pos = (int)P[(ty * FT_H + r) * TX + tx] + (j - st);
            a32[pos] = cnd;
          }
        }
      }
    }
    __syncthreads();
    if (B.phase_limit == 1) return;
    // ---- retainBest(2 * quota) by FAST score
    int k1 = cv_retain_best(a32, n, 2 * q, lpos, rpos, S, cv_heap, B.heap_cap, &SM);
    if (k1 < 0) { unsupported = true; k1 = 0; }
    if (B.phase_limit == 2) return;
    // ---- Harris response of the survivors, in place
    const uint8_t* img = A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off;
    for (int j = tid; j < k1; j += NT) {
      const uint32_t c = a32[j];
      const float r = harris_response(img, L.stride, (int)(c & 0xFFFu), (int)((c >> 12) & 0xFFFu));
      a[j] = ((unsigned long long)f32_order_key(r) << 32) | c;
    }
    __syncthreads();
    // ---- retainBest(quota) by Harris response
    k2 = cv_retain_best(a, k1, q, lpos, rpos, S, cv_heap, B.heap_cap, &SM);
    if (k2 < 0) { unsupported = true; k2 = 0; }
    if (k2 > A.kcap) { overflow = true; k2 = A.kcap; }
    for (int j = tid; j < k2; j += NT) {
      const unsigned long long e = a[j];
      const int64_t o = ((int64_t)f * EVH_NLEVELS + l) * A.kcap + j;
      A.tmp_meta[o] = ((uint32_t)l << 24) | ((uint32_t)e & 0xFFFFFFu);
      A.tmp_resp[o] = f32_from_order_key((uint32_t)(e >> 32));
    }
  }
  if (tid == 0) {
    A.lvl_count[f * EVH_NLEVELS + l] = k2;
    if (overflow) atomicOr(&A.frame_flags[f], 1);
    if (unsupported) atomicOr(&A.frame_flags[f], 2);
  }
}

// ------------------------------------------------------------------------------------------------------------
// K5 + K6: one wavefront per keypoint.  The 45x45 raw neighbourhood is staged in LDS once (16-byte loads) and serves the
// intensity-centroid orientation (radius-15 disc), the 7x7 sigma-2 fixed-point Gaussian (only the 39x39 region
// the steered taps can reach) and the 256 rotated BRIEF tests (4 x 64-lane ballots = the 32 descriptor bytes).
struct DescribeArgs {
  EvhLevel lv[EVH_NLEVELS];
  const uint8_t* pyr; int64_t pyr_frame_bytes;
  const float* kp_xy; const uint32_t* kp_meta; const int* kp_count;
  float* kp_angle; uint8_t* desc;
  int kcap;
};

__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
// byte masks of the radius-15 disc: c_omask[|v|][d] selects the bytes c = 4d..4d+3 of patch row v with |c - 22| <= umax[|v|]
__constant__ uint32_t c_omask[16][10] = {
  {0x00000000u, 0xFF000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x0000FFFFu},
  {0x00000000u, 0xFF000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x0000FFFFu},
  {0x00000000u, 0xFF000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x0000FFFFu},
  {0x00000000u, 0xFF000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x0000FFFFu},
  {0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x000000FFu},
  {0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x000000FFu},
  {0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x000000FFu},
  {0x00000000u, 0x00000000u, 0xFFFFFF00u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u},
  {0x00000000u, 0x00000000u, 0xFFFFFF00u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u},
  {0x00000000u, 0x00000000u, 0xFFFF0000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00FFFFFFu, 0x00000000u},
  {0x00000000u, 0x00000000u, 0xFF000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x0000FFFFu, 0x00000000u},
  {0x00000000u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x000000FFu, 0x00000000u},
  {0x00000000u, 0x00000000u, 0x00000000u, 0xFFFFFF00u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0x00000000u},
  {0x00000000u, 0x00000000u, 0x00000000u, 0xFFFF0000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00FFFFFFu, 0x00000000u, 0x00000000u},
  {0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x000000FFu, 0x00000000u, 0x00000000u},
  {0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0xFF000000u, 0xFFFFFFFFu, 0x0000FFFFu, 0x00000000u, 0x00000000u, 0x00000000u}};


#define DP_R 22                 // raw neighbourhood radius
#define DP_N (2 * DP_R + 1)     // 45
#define DP_STRIDE4 17           // dwords per staged row: 16 loaded + 1 (odd dword stride: row-per-lane reads are conflict-free)
#define DB_R 19                 // blurred radius reachable by steered taps
#define DB_N (2 * DB_R + 1)     // 39
#define DH_STRIDE 41            // u16 per row of the horizontal-pass buffer (odd: conflict-free row-per-lane writes)
#define DW_PER_BLOCK 4

// 7-tap sigma-2 kernel, symmetric: 18 34 49 55 49 34 18 (byte / 16-bit dot products in k_describe)
__global__ __launch_bounds__(64 * DW_PER_BLOCK, 8) void k_describe(DescribeArgs A) {
  // ONE LDS region per wave (3.7 KB), used in turn as the raw patch (45 x 68 B), the horizontal-pass buffer
  // (45 x 41 u16) and the blurred patch (39 x 39 B): every pass first loads all it needs into registers, a
  // wave-level fence follows, only then does it store the next form over the same words.  14.8 KB per workgroup.
  __shared__ uint32_t patch32[DW_PER_BLOCK][(DP_N * DH_STRIDE + 2) / 2 + 1];
  static_assert(DP_N * DP_STRIDE4 <= (DP_N * DH_STRIDE + 2) / 2 + 1 && DB_N * DB_N <= 4 * ((DP_N * DH_STRIDE + 2) / 2 + 1), "forms share one region");
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* raw = patch32[wv];
  uint16_t* hb = reinterpret_cast<uint16_t*>(patch32[wv]);
  uint8_t* blurp = reinterpret_cast<uint8_t*>(patch32[wv]);
  const int f = blockIdx.y;      // (XCD order measured 2 % slower here)
  const int k = blockIdx.x * DW_PER_BLOCK + wv;
  if (k >= A.kp_count[f]) return;  // whole wave exits; only wave-level synchronisation is used below
  const int64_t o = (int64_t)f * A.kcap + k;
  const uint32_t meta = A.kp_meta[o];
  const int l = (int)(meta >> 24);
  const EvhLevel L = A.lv[l];
  // centre exactly as computeOrbDescriptors recovers it from kp.pt
  const float inv = 1.f / L.scale;
  const int cx = (int)rintf(A.kp_xy[2 * o] * inv), cy = (int)rintf(A.kp_xy[2 * o + 1] * inv);
  const uint8_t* img = A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off;
  // the 45-byte patch rows lie inside the 64 bytes from the 16-byte boundary below their first pixel: each row is
  // four aligned 16-byte loads (180 per patch = 3 per lane, no division), stored with a 17-dword row stride
  // (odd: the row-per-lane reads below are conflict-free).  A key point is >= 31 px from every border of its level
  // and rows are padded to 64 bytes, so the window never leaves the level's rows.
  const int xs = cx - DP_R, xa = xs & ~15, shq = (xs - xa) >> 2;
  const uint32_t sh = (uint32_t)(xs & 3);
  {
    // 45 rows x 4 cells = 180 cells, lane = (row & 15, cell) three times over; the third round covers rows 32..47: its
    // loads are clamped to row 44 and its stores of rows 45..47 land in the part of the wave's region the raw form does
    // not use -- no predicate, so all three requests are in flight before the first store (the predicated form
    // compiled to two loads, wait, third load, wait)
    static_assert(47 * DP_STRIDE4 + 16 <= (DP_N * DH_STRIDE + 2) / 2 + 1, "rows 45..47 fit behind the raw patch");
    const uint4* src = reinterpret_cast<const uint4*>(img + xa);
    const int stride16 = L.stride >> 4;
    const int r0 = lane >> 2, c = lane & 3;
    uint4 v[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
      v[k] = src[mad24((uint32_t)(cy - DP_R + min(r0 + 16 * k, DP_N - 1)), (uint32_t)stride16, (uint32_t)c)];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      uint32_t* d = raw + (r0 + 16 * k) * DP_STRIDE4 + c * 4;
      d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
    }
  }
#define WAVE_LDS_SYNC()                                    \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
  __builtin_amdgcn_wave_barrier();                         \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")
  WAVE_LDS_SYNC();
  // ---- one lane per patch row: realign the row to the patch origin, then (a) orientation moments over the
  //      radius-15 disc and (b) the horizontal 7-tap pass with a sliding window
  int m10 = 0, m01 = 0;
  uint32_t w[12];
  if (lane < DP_N) {
    const uint32_t* rp = raw + lane * DP_STRIDE4 + shq;
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = __builtin_amdgcn_alignbyte(rp[j + 1], rp[j], sh);
  }
  WAVE_LDS_SYNC();      // every row is in registers: the region may now take the horizontal-pass values
  if (lane < DP_N) {
    const int v = lane - DP_R;
    // intensity-centroid moments over the disc as byte dot products: row bytes masked by the disc's extent in this
    // row, s0 = sum I, s1 = sum (u + 15) I - 15 s0 with u = c - 22 (weights 0..30 fit a byte); same integers as the
    // per-pixel sums
    int s0 = 0, s1 = 0;
    if (abs(v) <= 15) {
      const uint32_t* mk = c_omask[abs(v)];
      uint32_t a0 = 0, a1 = 0;
#pragma unroll
      for (int d = 1; d <= 9; d++) {               // bytes 4 .. 39 cover c = 7 .. 37
        const uint32_t x = w[d] & mk[d];
        const int u0 = 4 * d - DP_R + 15;          // weight of the dword's first byte; bytes outside 0..30 are masked off
        const uint32_t wt = ((uint32_t)(u0 & 0xFF)) | ((uint32_t)((u0 + 1) & 0xFF) << 8) | ((uint32_t)((u0 + 2) & 0xFF) << 16) |
                            ((uint32_t)((u0 + 3) & 0xFF) << 24);
        a0 = __builtin_amdgcn_udot4(x, 0x01010101u, a0, false);
        a1 = __builtin_amdgcn_udot4(x, wt, a1, false);
      }
      s0 = (int)a0; s1 = (int)a1 - 15 * (int)a0;
    }
    m10 = s1; m01 = v * s0;
    // horizontal 7-tap pass as byte dot products: X(c) = the dword of bytes c..c+3 of the realigned row (every fourth
    // one is a register as it stands, the others one v_alignbyte), h(c) = dot4(X(c), {18,34,49,55}) +
    // dot4(X(c+4), {49,34,18,0}) -- the same integer as gauss7 on the seven bytes
    uint16_t* hrow = hb + lane * DH_STRIDE;
    uint32_t X[DB_N + 4];
#pragma unroll
    for (int c = 0; c < DB_N + 4; c++)
      X[c] = (c & 3) == 0 ? w[c >> 2] : __builtin_amdgcn_alignbyte(w[(c >> 2) + 1], w[c >> 2], (uint32_t)(c & 3));
    const uint32_t W0 = 18u | (34u << 8) | (49u << 16) | (55u << 24), W1 = 49u | (34u << 8) | (18u << 16);
#pragma unroll
    for (int c = 0; c < DB_N; c++)
      hrow[c] = (uint16_t)__builtin_amdgcn_udot4(X[c + 4], W1, __builtin_amdgcn_udot4(X[c], W0, 0u, false), false);
  }
  for (int s = 32; s > 0; s >>= 1) { m10 += __shfl_xor(m10, s); m01 += __shfl_xor(m01, s); }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  WAVE_LDS_SYNC();
  // ---- one lane per blurred column: vertical 7-tap pass down the 45 rows.  The column is held as PAIRS of
  //      consecutive rows (h[2j] | h[2j+1] << 16: the second ds_read_u16 of a pair lands in the high half of the same
  //      register), an output row is then four v_dot2_u32_u16 with the tap pairs (18,34)(49,55)(49,34)(18,0) or
  //      (0,18)(34,49)(55,49)(34,18) -- the same integer as gauss7 on the seven values, 4 instead of 9 operations
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 P[(DP_N + 1) / 2];
  if (lane < DB_N) {
    const uint16_t* hc = hb + lane;
#pragma unroll
    for (int j = 0; j < DP_N / 2; j++) { P[j].x = hc[(2 * j) * DH_STRIDE]; P[j].y = hc[(2 * j + 1) * DH_STRIDE]; }
    P[DP_N / 2].x = hc[(DP_N - 1) * DH_STRIDE]; P[DP_N / 2].y = 0;       // row 45 does not exist (its tap weight is 0)
  }
  WAVE_LDS_SYNC();      // every column is in registers: the region may now take the blurred patch
  if (lane < DB_N) {
    const u16x2 WE[4] = {{18, 34}, {49, 55}, {49, 34}, {18, 0}}, WO[4] = {{0, 18}, {34, 49}, {55, 49}, {34, 18}};
#pragma unroll
    for (int r = 0; r < DB_N; r++) {              // output row r = taps on rows r .. r + 6 of the 45
      uint32_t sum = 32768u;
#pragma unroll
      for (int q = 0; q < 4; q++) sum = __builtin_amdgcn_udot2(P[(r >> 1) + q], (r & 1) ? WO[q] : WE[q], sum, false);
      blurp[r * DB_N + lane] = (uint8_t)(sum >> 16);
    }
  }
  WAVE_LDS_SYNC();
  // ---- steered BRIEF
  const float ang = angle * (float)(3.14159265358979323846 / 180.f);
  double sd, cd;
  det_sincos((double)ang, &sd, &cd);
  const float a = (float)cd, b = (float)sd;
  unsigned long long bits[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int8_t* p = c_pattern + (lane + 64 * m) * 4;
    float px0 = (float)p[0], py0 = (float)p[1], px1 = (float)p[2], py1 = (float)p[3];
    float fx0 = px0 * a - py0 * b, fy0 = px0 * b + py0 * a;
    float fx1 = px1 * a - py1 * b, fy1 = px1 * b + py1 * a;
    int t0 = blurp[((int)rintf(fy0) + DB_R) * DB_N + (int)rintf(fx0) + DB_R];
    int t1 = blurp[((int)rintf(fy1) + DB_R) * DB_N + (int)rintf(fx1) + DB_R];
    bits[m] = __ballot(t0 < t1);
  }
  if (lane < 4) reinterpret_cast<unsigned long long*>(A.desc + o * 32)[lane] = bits[lane];
  if (lane == 0) A.kp_angle[o] = angle;
#undef WAVE_LDS_SYNC
}

// host side of xcd_order / div_magic20
dim3 xcd_grid(int tiles, int frames) { return dim3((unsigned)((tiles + 7) & ~7), (unsigned)((frames + 7) & ~7)); }
int magic20(int d) { return (1 << 20) / d + 1; }   // exact for v * d < 2^20: at most 4095 / 128 x 4095 / 32 tiles
}  // namespace

// ------------------------------------------------------------------------------------------------------------
int evh_launch_gray_level0(evh_ctx* c, const uint8_t* d_frames, int nframes, int channels, int64_t row_stride,
                           int64_t frame_stride) {
  const EvhLevel& L = c->g.lv[0];
  const EvhLevel& D = c->g.lv[1];
  const int aligned4 = (((uintptr_t)d_frames | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 3) == 0;
  // level 1 is produced by the same launch whenever its scale keeps the footprint of a tile inside the LDS tile
  c->level1_fused = (int64_t)L.w * 100 <= (int64_t)D.w * 121 && (int64_t)L.h * 100 <= (int64_t)D.h * 121 && D.w >= 2 &&
                    D.h >= 2 && row_stride < (1 << 24);
  if (c->level1_fused) {
    const int tiles_x = (D.w + PD_W - 1) / PD_W, tiles_y = (D.h + PD_H - 1) / PD_H;
    const int* t = c->d_tabs + D.tab_off;
    if (channels == 3 && aligned4 && (L.w & 3) == 0)
      hipLaunchKernelGGL(k_gray_pyr1<true>, xcd_grid(tiles_x * tiles_y, nframes), dim3(256), 0, c->stream, d_frames, channels,
                       row_stride, frame_stride, aligned4, c->d_pyr, c->g.pyr_frame_bytes, L.stride, L.w, L.h, D.off,
                       D.stride, D.w, D.h, tiles_x, tiles_y, magic20(tiles_x), nframes, t, t + D.w, t + 2 * D.w,
                       t + 2 * D.w + D.h);
    else
      hipLaunchKernelGGL(k_gray_pyr1<false>, xcd_grid(tiles_x * tiles_y, nframes), dim3(256), 0, c->stream, d_frames, channels,
                       row_stride, frame_stride, aligned4, c->d_pyr, c->g.pyr_frame_bytes, L.stride, L.w, L.h, D.off,
                       D.stride, D.w, D.h, tiles_x, tiles_y, magic20(tiles_x), nframes, t, t + D.w, t + 2 * D.w,
                       t + 2 * D.w + D.h);
  } else {
    int quads = ((L.w + 3) / 4) * L.h;
    dim3 grid((quads + 255) / 256, nframes);
    hipLaunchKernelGGL(k_gray_level0, grid, dim3(256), 0, c->stream, d_frames, channels, row_stride, frame_stride,
                       c->d_pyr, c->g.pyr_frame_bytes, L.w, L.h, L.stride, aligned4);
  }
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_pyramid(evh_ctx* c, int nframes) {
  // A/B aids, read at every launch (tests switch them): the round-2 kernel (k_pyr_down, 128 x 64 tiles) and two levels per
  // launch (k_pyr_two: 24 % fewer HBM bytes, 70 % slower -- the pyramid is bound by its LDS byte reads and multiply-adds,
  // profiles/r03_pyramid_two_ab.txt)
  const bool old_form = getenv("EVH_PYR_OLD") != nullptr;
  const bool two_form = getenv("EVH_PYR_TWO") != nullptr;
  auto shrink_ok = [](const EvhLevel& S, const EvhLevel& D) {          // <= 1.21 x per axis: what the staged footprints are sized for
    return (int64_t)S.w * 100 <= (int64_t)D.w * 121 && (int64_t)S.h * 100 <= (int64_t)D.h * 121 && D.w >= 2 && D.h >= 2;
  };
  for (int l = c->level1_fused ? 2 : 1; l < EVH_NLEVELS; l++) {
    const EvhLevel& S = c->g.lv[l - 1];
    const EvhLevel& D = c->g.lv[l];
    const int* t = c->d_tabs + D.tab_off;   // xofs | xc1 | yofs | yc1 (linear_exact_tab in evh_api.hip)
    if (two_form && l + 1 < EVH_NLEVELS && shrink_ok(S, D) && shrink_ok(D, c->g.lv[l + 1])) {
      const EvhLevel& E = c->g.lv[l + 1];
      const int* t2 = c->d_tabs + E.tab_off;
      const Pyr2Level PM{D.off, D.stride, D.w, D.h, t, t + D.w, t + 2 * D.w, t + 2 * D.w + D.h};
      const Pyr2Level PD{E.off, E.stride, E.w, E.h, t2, t2 + E.w, t2 + 2 * E.w, t2 + 2 * E.w + E.h};
      const int tiles_x = (E.w + PD_W - 1) / PD_W, tiles_y = (E.h + PD_H - 1) / PD_H;
      hipLaunchKernelGGL(k_pyr_two, xcd_grid(tiles_x * tiles_y, nframes), dim3(256), 0, c->stream, c->d_pyr,
                         c->g.pyr_frame_bytes, S.off, S.stride, PM, PD, tiles_x, tiles_y, magic20(tiles_x), nframes);
      EVH_HIP(c, hipGetLastError());
      l++;
      continue;
    }
    // the row-walking kernel stages <= 1.21 x its tile: ORB's levels shrink by 1.2 (a first level built from a
    // resized frame never comes here: level 0 -> 1 has the same ratio)
    const bool walk = !old_form && shrink_ok(S, D);
    if (walk) {
      const int tiles_x = (D.w + PW_W - 1) / PW_W, tiles_y = (D.h + PW_H - 1) / PW_H;
      hipLaunchKernelGGL(k_pyr_walk, xcd_grid(tiles_x * tiles_y, nframes), dim3(256), 0, c->stream, c->d_pyr,
                         c->g.pyr_frame_bytes, S.off, S.stride, D.off, D.stride, D.w, D.h, tiles_x,
                         magic20(tiles_x), tiles_x * tiles_y, nframes, t, t + D.w, t + 2 * D.w, t + 2 * D.w + D.h);
    } else {
      const int tiles_x = (D.w + PD_W - 1) / PD_W, tiles_y = (D.h + PDN_H - 1) / PDN_H;
      hipLaunchKernelGGL(k_pyr_down, xcd_grid(tiles_x * tiles_y, nframes), dim3(256), 0, c->stream, c->d_pyr,
                         c->g.pyr_frame_bytes, S.off, S.stride, D.off, D.stride, D.w, D.h, tiles_x,
                         magic20(tiles_x), tiles_x * tiles_y, nframes, t, t + D.w, t + 2 * D.w, t + 2 * D.w + D.h);
    }
    EVH_HIP(c, hipGetLastError());
  }
  return EVH_SUCCESS;
}

int evh_launch_fast(evh_ctx* c, int nframes, int share_group) {
  EVH_HIP(c, hipMemsetAsync(c->d_cand_count, 0, sizeof(int) * EVH_NLEVELS * (size_t)nframes, c->stream));
  FastArgs A;
  for (int l = 0; l < EVH_NLEVELS; l++) A.lv[l] = c->g.lv[l];
  A.pyr = c->d_pyr; A.pyr_frame_bytes = c->g.pyr_frame_bytes;
  A.cand = c->d_cand; A.cand_frame_entries = c->g.cand_frame_entries;
  A.cand_count = c->d_cand_count;
  A.thr = c->d_fast_thr; A.shist = c->d_fast_hist; A.redo = c->d_fast_redo;
  A.share_group = share_group > 1 ? share_group : 0;
  A.hint_in = c->fast_hint ? c->d_fast_hint + 8 * c->fast_hint_idx : c->d_fast_hint + 16 + 8 * 256;   // off: all zero
  A.hint_out = c->d_fast_hint + 8 * (c->fast_hint_idx ^ 1);
  A.hint_hist = reinterpret_cast<unsigned*>(c->d_fast_hint + 16);
  c->fast_hint_idx ^= 1;
  int nsamp = 0;
  for (int l = 0; l < EVH_NLEVELS; l++) {
    const int tiles = A.lv[l].tiles_x * A.lv[l].tiles_y;
    A.samp_mod[l] = tiles >= 128 ? 27 : tiles >= 32 ? 13 : tiles >= 14 ? 7 : 0;   // 0: too small to sample, T stays 20
    A.samp_start[l] = nsamp;
    if (A.samp_mod[l]) nsamp += (tiles + A.samp_mod[l] - 1) / A.samp_mod[l];
  }
  const dim3 grid(c->g.total_tiles, nframes);
  A.lift_base = 0;
  A.tdesc = c->order_mode == EVH_ORDER_OPENCV ? c->d_cv_tdesc : nullptr;
  A.total_tiles = c->g.total_tiles;
  // the reference's key-point order is a function of EVERY corner at threshold 20 (k_select_cv): the threshold cannot be
  // lifted there, but the exact score is still only needed where the 4-point pre-test at 20 passes
  // reference order + lifting allowed: the full 16-point segment test decides which pixels get an exact score (k_fast_main with
  // lift_base; the 4-point pre-test alone was measured and dropped here: 22.8 ms against 19.8 ms dense on the 720p texture of
  // SURVEY 8d, where it passes most quads).  EVH_FAST_DENSE=1 or evh_set_fast_lift(0): the dense kernel.
  if (c->order_mode == EVH_ORDER_OPENCV && c->fast_lift && !getenv("EVH_FAST_DENSE")) {
    A.lift_base = 1;
    hipLaunchKernelGGL(k_fast_main, grid, dim3(256), 0, c->stream, A);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  if (!c->fast_lift || nsamp == 0 || c->order_mode == EVH_ORDER_OPENCV) {
    hipLaunchKernelGGL(k_fast, grid, dim3(256), 0, c->stream, A);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  const int nfl = nframes * EVH_NLEVELS;
  EVH_HIP(c, hipMemsetAsync(c->d_fast_hist, 0, sizeof(unsigned) * 256 * (size_t)nfl, c->stream));
  EVH_HIP(c, hipMemsetAsync(A.hint_hist, 0, sizeof(unsigned) * 8 * 256, c->stream));
  hipLaunchKernelGGL(k_fast_sample, dim3(nsamp, nframes), dim3(256), 0, c->stream, A);
  hipLaunchKernelGGL(k_fast_thr, dim3((nfl + 255) / 256), dim3(256), 0, c->stream, A, nframes);
  hipLaunchKernelGGL(k_fast_main, grid, dim3(256), 0, c->stream, A);
  hipLaunchKernelGGL(k_fast_verify, dim3((nfl + 255) / 256), dim3(256), 0, c->stream, A, nframes);
  hipLaunchKernelGGL(k_fast_hint, dim3(1), dim3(64), 0, c->stream, A);
  hipLaunchKernelGGL(k_fast_redo, dim3(A.lv[0].tiles_x * A.lv[0].tiles_y, std::min(nfl, 256)), dim3(256), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_select(evh_ctx* c, int nframes) {
  SelectArgs A;
  for (int l = 0; l < EVH_NLEVELS; l++) A.lv[l] = c->g.lv[l];
  A.pyr = c->d_pyr; A.pyr_frame_bytes = c->g.pyr_frame_bytes;
  A.cand = c->d_cand; A.cand_frame_entries = c->g.cand_frame_entries; A.cand_count = c->d_cand_count;
  A.kp_xy = c->d_kp_xy; A.kp_meta = c->d_kp_meta; A.kp_resp = c->d_kp_resp; A.kp_count = c->d_kp_count;
  A.frame_flags = c->d_frame_flags; A.kcap = c->kcap;
  A.tmp_meta = c->d_tmp_meta; A.tmp_resp = c->d_tmp_resp; A.lvl_count = c->d_lvl_count;
  EVH_HIP(c, hipMemsetAsync(c->d_frame_flags, 0, sizeof(int) * (size_t)nframes, c->stream));
  // LDS capacities: stage 1 keeps 2*q0 + score ties in LDS up to k1cap and spills beyond it (exact either way);
  // stage 2 can hold a whole frame slot (kcap), the only hard bound left
  int q0 = 0;
  for (int l = 0; l < EVH_NLEVELS; l++) q0 = std::max(q0, A.lv[l].quota);
  A.k1cap = std::min(EVH_K1CAP, std::max(1024, (4 * q0 + 63) / 64 * 64));
  A.k2cap = c->kcap;
  const size_t lds = sizeof(uint32_t) * 2 * ((size_t)A.k1cap + A.k2cap);
  A.nframes = nframes;
  if (c->order_mode == EVH_ORDER_OPENCV) {
    SelCvArgs B;
    B.s = A;
    B.seq = c->d_cv_seq; B.seq32 = c->d_cv_seq32; B.lpos = c->d_cv_lpos; B.rpos = c->d_cv_rpos; B.mask = c->d_cv_mask;
    B.mask_frame_words = c->cv_mask_frame_words;
    int mo = 0;
    for (int l = 0; l < EVH_NLEVELS; l++) { B.mask_off[l] = mo; mo += ((A.lv[l].w + 31) / 32) * A.lv[l].h; }
    if (mo > c->cv_mask_frame_words) return evh_fail(c, EVH_ERR_CAPACITY, "evh_launch_select: corner bit plane larger than the context's");
    B.heap_cap = 2 * q0 + 2;
    B.tdesc = c->d_cv_tdesc; B.total_tiles = c->g.total_tiles;
    { const char* e = getenv("EVH_CV_PHASE"); B.phase_limit = e ? atoi(e) : 0; }
    // levels with many corners (>= ~8000 expected: area above 0.3 Mpx) on 1024 threads, the rest on 256
    // (measured: 1024 threads for the large levels 10.6 ms against 6.4 ms with 256 everywhere -- a barrier over 16 waves costs
    // more than the steps it saves; EVH_CV_SPLIT / EVH_CV_NT keep the experiment reachable)
    int split = 0, nt_small = 256;
    { const char* e = getenv("EVH_CV_SPLIT"); if (e) split = std::min(EVH_NLEVELS, std::max(0, atoi(e))); }
    { const char* e = getenv("EVH_CV_NT"); if (e) nt_small = std::min(CV_MAXT, std::max(64, atoi(e) & ~63)); }
    if (CV_MAXT < 1024) split = 0;
    const size_t heap_bytes = sizeof(unsigned long long) * (size_t)B.heap_cap;
    if (split > 0) {
      B.level0 = 0; B.nlev = split;
      hipLaunchKernelGGL(k_select_cv, xcd_grid(split, nframes), dim3(1024), heap_bytes, c->stream, B);
    }
    if (split < EVH_NLEVELS) {
      B.level0 = split; B.nlev = EVH_NLEVELS - split;
      hipLaunchKernelGGL(k_select_cv, xcd_grid(EVH_NLEVELS - split, nframes), dim3(nt_small), heap_bytes, c->stream, B);
    }
    EVH_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(k_pack, dim3(nframes), dim3(256), 0, c->stream, A);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  if (lds > 48 * 1024)   // from ~3000 key points on (74.7 KB at 4000): opt in to more dynamic LDS than the default
    EVH_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_select), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
  hipLaunchKernelGGL(k_select, xcd_grid(EVH_NLEVELS, nframes), dim3(256), lds, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(k_pack, dim3(nframes), dim3(256), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_describe(evh_ctx* c, int nframes) {
  DescribeArgs A;
  for (int l = 0; l < EVH_NLEVELS; l++) A.lv[l] = c->g.lv[l];
  A.pyr = c->d_pyr; A.pyr_frame_bytes = c->g.pyr_frame_bytes;
  A.kp_xy = c->d_kp_xy; A.kp_meta = c->d_kp_meta; A.kp_count = c->d_kp_count;
  A.kp_angle = c->d_kp_angle; A.desc = c->d_desc; A.kcap = c->kcap;
  hipLaunchKernelGGL(k_describe, dim3((c->kcap + DW_PER_BLOCK - 1) / DW_PER_BLOCK, nframes), dim3(64 * DW_PER_BLOCK), 0,
                     c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
