// evh_detect.hip -- ORB detect + describe for gfx950 (MI355X): gray conversion, 8-level pyramid, FAST-9/16 with
// corner score + 3x3 NMS, per-level selection (FAST score, then Harris), orientation, steered BRIEF.
// Replaces cv2.ORB_create().detectAndCompute (reference: evenvizion/processing/frame_processing.py:59-61).
// Integer stages are exact; float stages use one IEEE operation at a time (-ffp-contract=off).
#include "evh_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// K1: BGR -> gray (Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14) or gray copy, into pyramid level 0.
// One thread = 4 output pixels (one dword store); grid.y = frame.
__global__ void k_gray_level0(const uint8_t* __restrict__ src, int channels, int64_t row_stride, int64_t frame_stride,
                              uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes, int w, int h, int dst_stride) {
  int f = blockIdx.y;
  int qpr = (w + 3) >> 2;
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= qpr * h) return;
  int y = q / qpr, x = (q - y * qpr) * 4;
  const uint8_t* s = src + (int64_t)f * frame_stride + (int64_t)y * row_stride + (int64_t)x * channels;
  uint8_t* d = pyr + (int64_t)f * pyr_frame_bytes + (int64_t)y * dst_stride + x;
  uint32_t out = 0;
  int n = min(4, w - x);
  if (channels == 1) {
    for (int i = 0; i < n; i++) out |= (uint32_t)s[i] << (8 * i);
  } else {
    for (int i = 0; i < n; i++) {
      uint32_t b = s[3 * i], g = s[3 * i + 1], r = s[3 * i + 2];
      out |= ((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14) << (8 * i);
    }
  }
  *reinterpret_cast<uint32_t*>(d) = out;  // rows are 64-byte aligned and padded, a full dword is always in range
}

// ------------------------------------------------------------------------------------------------------------
// K2: pyramid level l from level l-1, resize(INTER_LINEAR_EXACT): 8.8 fixed-point weights per axis,
// out = ((c0*s00 + c1*s01)*m0 + (c0*s10 + c1*s11)*m1 + 32768) >> 16.  Tables (host-computed): per dst column
// (xofs, xc1), per dst row (yofs, yc1); edge replication is encoded in the tables.
__global__ void k_pyr_down(uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes, int64_t src_off, int src_stride,
                           int64_t dst_off, int dst_stride, int dw, int dh, const int* __restrict__ xofs,
                           const int* __restrict__ xc1, const int* __restrict__ yofs, const int* __restrict__ yc1) {
  int f = blockIdx.y;
  int qpr = (dw + 3) >> 2;
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= qpr * dh) return;
  int y = q / qpr, x = (q - y * qpr) * 4;
  const uint8_t* base = pyr + (int64_t)f * pyr_frame_bytes;
  const uint8_t* r0 = base + src_off + (int64_t)yofs[y] * src_stride;
  const uint8_t* r1 = r0 + src_stride;
  uint32_t m1 = (uint32_t)yc1[y], m0 = 256u - m1;
  uint32_t out = 0;
  int n = min(4, dw - x);
  for (int i = 0; i < n; i++) {
    int o = xofs[x + i];
    uint32_t c1 = (uint32_t)xc1[x + i], c0 = 256u - c1;
    uint32_t h0 = c0 * r0[o] + c1 * r0[o + 1];
    uint32_t h1 = c0 * r1[o] + c1 * r1[o + 1];
    out |= ((h0 * m0 + h1 * m1 + 32768u) >> 16) << (8 * i);
  }
  *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(base) + dst_off + (int64_t)y * dst_stride + x) = out;
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
};

#define FT_W 64
#define FT_H 32
#define FT_LW (FT_W + 8)   // 72 bytes per staged row
#define FT_LH (FT_H + 8)   // 40 rows
#define FS_W (FT_W + 2)    // score plane 66 x 34
#define FS_H (FT_H + 2)

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }

__global__ __launch_bounds__(256) void k_fast(FastArgs A) {
  __shared__ uint32_t tile32[FT_LW * FT_LH / 4];
  __shared__ uint8_t score[FS_W * FS_H + 2];
  __shared__ uint32_t lst[FT_W * FT_H / 4];  // NMS keeps at most one corner per 2x2 block
  __shared__ int lcnt, gbase;
  uint8_t* tile = reinterpret_cast<uint8_t*>(tile32);
  if (threadIdx.x == 0) lcnt = 0;
  int f = blockIdx.y;
  int t = blockIdx.x;
  int l = 0;
#pragma unroll
  for (int i = 1; i < EVH_NLEVELS; i++)
    if (t >= A.lv[i].tile_start) l = i;
  const EvhLevel L = A.lv[l];
  t -= L.tile_start;
  int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
  int x0 = tx * FT_W, y0 = ty * FT_H;  // tile origin in level coordinates
  const uint8_t* img = A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off;
  // stage rows y0-4 .. y0+35, columns x0-4 .. x0+67 (clamped; clamped pixels never influence a tested centre)
  for (int i = threadIdx.x; i < FT_LH * (FT_LW / 4); i += 256) {
    int r = i / (FT_LW / 4), c4 = i - r * (FT_LW / 4);
    int y = min(max(y0 - 4 + r, 0), L.h - 1);
    int x = x0 - 4 + c4 * 4;
    uint32_t v;
    if (x >= 0 && x + 3 < L.stride) v = *reinterpret_cast<const uint32_t*>(img + (int64_t)y * L.stride + x);
    else {
      v = 0;
      for (int k = 0; k < 4; k++) v |= (uint32_t)img[(int64_t)y * L.stride + min(max(x + k, 0), L.w - 1)] << (8 * k);
    }
    tile32[i] = v;
  }
  __syncthreads();
  const int thr = EVH_FAST_THR;
  for (int i = threadIdx.x; i < FS_W * FS_H; i += 256) {
    int sy = i / FS_W, sx = i - sy * FS_W;
    int x = x0 - 1 + sx, y = y0 - 1 + sy;
    int s = 0;
    if (x >= 3 && x < L.w - 3 && y >= 3 && y < L.h - 3) {
      const uint8_t* p = tile + (sy + 3) * FT_LW + (sx + 3);
      int v = p[0];
      int d[16];
      d[0] = v - p[3 * FT_LW];           d[1] = v - p[3 * FT_LW + 1];   d[2] = v - p[2 * FT_LW + 2];
      d[3] = v - p[FT_LW + 3];           d[4] = v - p[3];               d[5] = v - p[-FT_LW + 3];
      d[6] = v - p[-2 * FT_LW + 2];      d[7] = v - p[-3 * FT_LW + 1];  d[8] = v - p[-3 * FT_LW];
      d[9] = v - p[-3 * FT_LW - 1];      d[10] = v - p[-2 * FT_LW - 2]; d[11] = v - p[-FT_LW - 3];
      d[12] = v - p[-3];                 d[13] = v - p[FT_LW - 3];      d[14] = v - p[2 * FT_LW - 2];
      d[15] = v - p[3 * FT_LW - 1];
      // quick reject: any 9-arc holds two adjacent compass points (0,4,8,12)
      bool dk = (d[0] > thr && d[4] > thr) || (d[4] > thr && d[8] > thr) || (d[8] > thr && d[12] > thr) ||
                (d[12] > thr && d[0] > thr);
      bool br = (d[0] < -thr && d[4] < -thr) || (d[4] < -thr && d[8] < -thr) || (d[8] < -thr && d[12] < -thr) ||
                (d[12] < -thr && d[0] < -thr);
      if (dk || br) {
        int lo3[16], hi3[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
          hi3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        }
        int best = -256;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          int mn = min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);    // min of d over arc k..k+8
          int mx = max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);    // max of d over arc k..k+8
          best = max3i(best, mn, -mx);
        }
        if (best > thr) s = best - 1;
      }
    }
    score[i] = (uint8_t)s;
  }
  __syncthreads();
  const bool level_ok = (L.w > 2 * EVH_EDGE) && (L.h > 2 * EVH_EDGE);
  // NMS + border filter; survivors are collected in LDS, then ONE global atomic per workgroup reserves their slots
  // (a returning global atomic per wave-iteration serialises on its ~1-2 us latency).
#pragma unroll 1
  for (int k = 0; k < FT_W * FT_H / 256; k++) {
    int i = threadIdx.x + k * 256;
    int py = i / FT_W, px = i - py * FT_W;
    int x = x0 + px, y = y0 + py;
    const uint8_t* c = score + (py + 1) * FS_W + (px + 1);
    int s = c[0];
    if (s && level_ok && x >= EVH_EDGE && x < L.w - EVH_EDGE && y >= EVH_EDGE && y < L.h - EVH_EDGE) {
      bool keep = s > c[-1] && s > c[1] && s > c[-FS_W - 1] && s > c[-FS_W] && s > c[-FS_W + 1] && s > c[FS_W - 1] &&
                  s > c[FS_W] && s > c[FS_W + 1];
      if (keep) {
        int slot = atomicAdd(&lcnt, 1);
        lst[slot] = ((uint32_t)s << 24) | ((uint32_t)y << 12) | (uint32_t)x;
      }
    }
  }
  __syncthreads();
  const int n = lcnt;
  if (n == 0) return;
  if (threadIdx.x == 0) gbase = atomicAdd(A.cand_count + f * EVH_NLEVELS + l, n);
  __syncthreads();
  uint32_t* out = A.cand + (int64_t)f * A.cand_frame_entries + L.cand_off;
  const int base = gbase;
  for (int i = threadIdx.x; i < n; i += 256)
    if (base + i < L.cand_cap) out[base + i] = lst[i];
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
  int kcap;
};

__device__ __forceinline__ float harris_response(const uint8_t* img, int stride, int x0, int y0) {
  int a = 0, b = 0, c = 0;
  for (int i = -3; i <= 3; i++) {
    const uint8_t* pm = img + (int64_t)(y0 + i - 1) * stride + x0;
    const uint8_t* p0 = pm + stride;
    const uint8_t* pp = p0 + stride;
#pragma unroll
    for (int j = -3; j <= 3; j++) {
      int Ix = (p0[j + 1] - p0[j - 1]) * 2 + (pm[j + 1] - pm[j - 1]) + (pp[j + 1] - pp[j - 1]);
      int Iy = (pp[j] - pm[j]) * 2 + (pp[j - 1] - pm[j - 1]) + (pp[j + 1] - pm[j + 1]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
  }
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

__global__ __launch_bounds__(256) void k_select(SelectArgs A) {
  __shared__ uint32_t keys[EVH_K1CAP];
  __shared__ float resp[EVH_K1CAP];
  __shared__ uint32_t sel[EVH_K2CAP];
  __shared__ float selr[EVH_K2CAP];
  __shared__ uint32_t hist[256];
  __shared__ int sh_i[8];  // 0: cut / prefix, 1: k1, 2: k2, 3: remaining, 4: overflow
  const int f = blockIdx.x, tid = threadIdx.x;
  int base = 0;
  bool overflow = false;
  for (int l = 0; l < EVH_NLEVELS; l++) {
    const EvhLevel L = A.lv[l];
    const uint32_t* cand = A.cand + (int64_t)f * A.cand_frame_entries + L.cand_off;
    int n_raw = A.cand_count[f * EVH_NLEVELS + l];
    if (n_raw > L.cand_cap) overflow = true;
    const int n = min(n_raw, L.cand_cap);
    const int q = L.quota;
    if (n == 0 || q == 0) { __syncthreads(); continue; }
    // ---- stage 1: cut on the integer FAST score through a 256-bin histogram
    hist[tid] = 0;
    if (tid < 8) sh_i[tid] = 0;
    __syncthreads();
    if (n > 2 * q)
      for (int i = tid; i < n; i += 256) atomicAdd(&hist[cand[i] >> 24], 1u);
    __syncthreads();
    if (tid == 0) {
      int cut = 0;
      if (n > 2 * q) {
        int acc = 0;
        for (int s = 255; s >= 0; s--) { acc += (int)hist[s]; if (acc >= 2 * q) { cut = s; break; } }
      }
      sh_i[0] = cut;
    }
    __syncthreads();
    const uint32_t cut = (uint32_t)sh_i[0];
    for (int i = tid; i < n; i += 256) {
      uint32_t c = cand[i];
      if ((c >> 24) >= cut) {
        int slot = atomicAdd(&sh_i[1], 1);
        if (slot < EVH_K1CAP) keys[slot] = c;
      }
    }
    __syncthreads();
    int k1 = sh_i[1];
    if (k1 > EVH_K1CAP) { overflow = true; k1 = EVH_K1CAP; }
    // ---- Harris response of every stage-1 survivor
    const uint8_t* img = A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off;
    for (int j = tid; j < k1; j += 256) {
      uint32_t c = keys[j];
      resp[j] = harris_response(img, L.stride, (int)(c & 0xFFFu), (int)((c >> 12) & 0xFFFu));
    }
    __syncthreads();
    // ---- stage 2: value of the q-th largest response by a 4 x 8-bit radix select on order-preserving keys
    float cutf = -INFINITY;
    if (k1 > q) {
      uint32_t prefix = 0;
      if (tid == 0) sh_i[3] = q;
      for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (int j = tid; j < k1; j += 256) {
          uint32_t u = f32_order_key(resp[j]);
          bool in = pass == 0 ? true : ((u >> (shift + 8)) == (prefix >> (shift + 8)));
          if (in) atomicAdd(&hist[(u >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          int rem = sh_i[3], acc = 0, d = 255;
          for (; d > 0; d--) { if (acc + (int)hist[d] >= rem) break; acc += (int)hist[d]; }
          sh_i[3] = rem - acc;
          sh_i[0] = d;
        }
        __syncthreads();
        prefix |= (uint32_t)sh_i[0] << shift;
        __syncthreads();
      }
      uint32_t u = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
      cutf = __uint_as_float(u);
    }
    for (int j = tid; j < k1; j += 256)
      if (resp[j] >= cutf) {
        int slot = atomicAdd(&sh_i[2], 1);
        if (slot < EVH_K2CAP) { sel[slot] = keys[j]; selr[slot] = resp[j]; }
      }
    __syncthreads();
    int k2 = sh_i[2];
    if (k2 > EVH_K2CAP) { overflow = true; k2 = EVH_K2CAP; }
    if (base + k2 > A.kcap) { overflow = true; k2 = max(A.kcap - base, 0); }
    // ---- canonical order inside the level: ascending (y, x) by rank counting
    for (int j = tid; j < k2; j += 256) {
      uint32_t kj = sel[j] & 0xFFFFFFu;
      int pos = 0;
      for (int i = 0; i < k2; i++) pos += ((sel[i] & 0xFFFFFFu) < kj) ? 1 : 0;
      int x = (int)(kj & 0xFFFu), y = (int)(kj >> 12);
      int64_t o = (int64_t)f * A.kcap + base + pos;
      A.kp_meta[o] = ((uint32_t)l << 24) | kj;
      A.kp_xy[2 * o] = (float)x * L.scale;
      A.kp_xy[2 * o + 1] = (float)y * L.scale;
      A.kp_resp[o] = selr[j];
    }
    base += k2;
    __syncthreads();
  }
  if (tid == 0) {
    A.kp_count[f] = base;
    A.frame_flags[f] = overflow ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------------------
// K5 + K6: one wavefront per keypoint.  The 45x45 raw neighbourhood is staged in LDS once and serves the
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
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
__constant__ int c_gauss[7] = {18, 34, 49, 55, 49, 34, 18};  // cvRound(256 * normalised exp(-x^2/8)), x=-3..3

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  float ax = fabsf(x), ay = fabsf(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// sin/cos for x in [0, 2*pi]: Cody-Waite reduction by pi/2 + fixed-order polynomial kernels (bit-reproducible)
__device__ __forceinline__ void det_sincos(double x, double* so, double* co) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00;
  const double pio2_lo = 6.07710050650619224932e-11;
  double fn = __builtin_rint(x * two_over_pi);
  int n = (int)fn;
  double r = (x - fn * pio2_hi) - fn * pio2_lo;
  double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  double ks = r + (z * r) * (S1 + z * ps);
  double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double kc = 1.0 - (0.5 * z - z * pc);
  double s, c;
  switch (n & 3) {
    case 0: s = ks; c = kc; break;
    case 1: s = kc; c = -ks; break;
    case 2: s = -ks; c = -kc; break;
    default: s = -kc; c = ks; break;
  }
  *so = s; *co = c;
}

#define DP_R 22                 // raw neighbourhood radius
#define DP_N (2 * DP_R + 1)     // 45
#define DP_STRIDE 52            // 13 dwords per staged row
#define DB_R 19                 // blurred radius reachable by steered taps
#define DB_N (2 * DB_R + 1)     // 39
#define DW_PER_BLOCK 4

__global__ __launch_bounds__(64 * DW_PER_BLOCK) void k_describe(DescribeArgs A) {
  __shared__ uint32_t raw32[DW_PER_BLOCK][DP_N * DP_STRIDE / 4];
  __shared__ uint16_t hbuf[DW_PER_BLOCK][DP_N * DB_N];
  __shared__ uint8_t blur[DW_PER_BLOCK][DB_N * DB_N + 3];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int f = blockIdx.y;
  const int k = blockIdx.x * DW_PER_BLOCK + wv;
  if (k >= A.kp_count[f]) return;  // whole wave exits; no block-level barrier is used below
  const int64_t o = (int64_t)f * A.kcap + k;
  const uint32_t meta = A.kp_meta[o];
  const int l = (int)(meta >> 24);
  const EvhLevel L = A.lv[l];
  // centre exactly as computeOrbDescriptors recovers it from kp.pt
  const float inv = 1.f / L.scale;
  const int cx = (int)rintf(A.kp_xy[2 * o] * inv), cy = (int)rintf(A.kp_xy[2 * o + 1] * inv);
  const uint8_t* img = A.pyr + (int64_t)f * A.pyr_frame_bytes + L.off;
  const int xs = cx - DP_R, sh = xs & 3, xa = xs - sh;
  uint8_t* raw = reinterpret_cast<uint8_t*>(raw32[wv]);
  for (int i = lane; i < DP_N * (DP_STRIDE / 4); i += 64) {
    int r = i / (DP_STRIDE / 4), c4 = i - r * (DP_STRIDE / 4);
    raw32[wv][i] = *reinterpret_cast<const uint32_t*>(img + (int64_t)(cy - DP_R + r) * L.stride + xa + c4 * 4);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#define RAW(r, c) raw[(r) * DP_STRIDE + (c) + sh]
  // ---- orientation: m10 = sum u*I, m01 = sum v*I over the radius-15 disc
  int m10 = 0, m01 = 0;
  for (int i = lane; i < 31 * 31; i += 64) {
    int v = i / 31 - 15, u = i - (v + 15) * 31 - 15;
    if (abs(u) <= c_umax[abs(v)]) {
      int I = RAW(DP_R + v, DP_R + u);
      m10 += u * I; m01 += v * I;
    }
  }
  for (int s = 32; s > 0; s >>= 1) { m10 += __shfl_xor(m10, s); m01 += __shfl_xor(m01, s); }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // ---- 7x7 Gaussian, horizontal then vertical, integer
  for (int i = lane; i < DP_N * DB_N; i += 64) {
    int r = i / DB_N, c = i - r * DB_N + (DP_R - DB_R);
    int s = 0;
#pragma unroll
    for (int t = -3; t <= 3; t++) s += c_gauss[t + 3] * RAW(r, c + t);
    hbuf[wv][i] = (uint16_t)s;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int i = lane; i < DB_N * DB_N; i += 64) {
    int r = i / DB_N, c = i - r * DB_N;
    int s = 0;
#pragma unroll
    for (int t = -3; t <= 3; t++) s += c_gauss[t + 3] * (int)hbuf[wv][(r + (DP_R - DB_R) + t) * DB_N + c];
    blur[wv][i] = (uint8_t)((s + 32768) >> 16);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
    int t0 = blur[wv][((int)rintf(fy0) + DB_R) * DB_N + (int)rintf(fx0) + DB_R];
    int t1 = blur[wv][((int)rintf(fy1) + DB_R) * DB_N + (int)rintf(fx1) + DB_R];
    bits[m] = __ballot(t0 < t1);
  }
  if (lane < 4) reinterpret_cast<unsigned long long*>(A.desc + o * 32)[lane] = bits[lane];
  if (lane == 0) A.kp_angle[o] = angle;
#undef RAW
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
int evh_launch_gray_level0(evh_ctx* c, const uint8_t* d_frames, int nframes, int channels, int64_t row_stride,
                           int64_t frame_stride) {
  const EvhLevel& L = c->g.lv[0];
  int quads = ((L.w + 3) / 4) * L.h;
  dim3 grid((quads + 255) / 256, nframes);
  hipLaunchKernelGGL(k_gray_level0, grid, dim3(256), 0, c->stream, d_frames, channels, row_stride, frame_stride, c->d_pyr,
                     c->g.pyr_frame_bytes, L.w, L.h, L.stride);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_pyramid(evh_ctx* c, int nframes) {
  for (int l = 1; l < EVH_NLEVELS; l++) {
    const EvhLevel& S = c->g.lv[l - 1];
    const EvhLevel& D = c->g.lv[l];
    int quads = ((D.w + 3) / 4) * D.h;
    dim3 grid((quads + 255) / 256, nframes);
    const int* t = c->d_tabs + D.tab_off;
    hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, c->stream, c->d_pyr, c->g.pyr_frame_bytes, S.off, S.stride, D.off,
                       D.stride, D.w, D.h, t, t + D.w, t + 2 * D.w, t + 2 * D.w + D.h);
    EVH_HIP(c, hipGetLastError());
  }
  return EVH_SUCCESS;
}

int evh_launch_fast(evh_ctx* c, int nframes) {
  EVH_HIP(c, hipMemsetAsync(c->d_cand_count, 0, sizeof(int) * EVH_NLEVELS * (size_t)nframes, c->stream));
  FastArgs A;
  for (int l = 0; l < EVH_NLEVELS; l++) A.lv[l] = c->g.lv[l];
  A.pyr = c->d_pyr; A.pyr_frame_bytes = c->g.pyr_frame_bytes;
  A.cand = c->d_cand; A.cand_frame_entries = c->g.cand_frame_entries;
  A.cand_count = c->d_cand_count;
  hipLaunchKernelGGL(k_fast, dim3(c->g.total_tiles, nframes), dim3(256), 0, c->stream, A);
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
  hipLaunchKernelGGL(k_select, dim3(nframes), dim3(256), 0, c->stream, A);
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
