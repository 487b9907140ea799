// evh_sift.hip -- N4: SIFT detectAndCompute on the GPU, the MI355X counterpart of
//   cv2.xfeatures2d.SIFT_create().detectAndCompute(frame, None)      evenvizion/processing/frame_processing.py:62-64
// (opencv-contrib 3.4.2 defaults: nOctaveLayers 3, contrastThreshold 0.04, edgeThreshold 10, sigma 1.6, every key point
// kept).  Float scale space in HBM, 6 Gaussian layers per octave, octave 0 = the frame doubled; the difference-of-
// Gaussians is never stored: D_l = G_{l+1} - G_l is one exact subtraction where it is read.
//
// Everything that decides a key point (threshold / extremum tests, the sub-pixel fit, the edge test) and every
// histogram accumulation follows the operator's own summation order -- each orientation bin and each of the 4x4x8
// descriptor bins is accumulated by ONE thread walking the samples in raster order -- so that the results can be
// compared bit for bit with the CPU oracle (the library is built with -ffp-contract=off, IEEE divide / sqrt).
//
// Kernels: k_sift_upsample (frame x2, INTER_LINEAR), k_sift_blur_row / k_sift_blur_col (separable Gaussian, LDS-staged
// rows / column tiles, reflect-101), k_sift_down (next octave: nearest /2), k_sift_extrema (26-neighbour test on the
// DoG), k_sift_refine (one wavefront per extremum: sub-pixel fit, 36-bin orientation histogram, one key point per
// peak), k_sift_rank + k_sift_dedup (KeyPointsFilter::removeDuplicatedSorted: the operator's own output order),
// k_sift_desc (one workgroup per key point: 4x4x8 histogram, normalise / clip 0.2 / x512 / round to uint8).
#include "evh_internal.h"
#include "evh_devmath.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int SL = 3;             // nOctaveLayers
constexpr int SNG = SL + 3;       // Gaussian layers per octave
constexpr int SBORDER = 5;        // SIFT_IMG_BORDER
constexpr int SMAXTAPS = 27;      // largest Gaussian kernel (sigma 3.09 -> 27 taps)
constexpr int SORI_R = 16;        // largest orientation-window radius: round(4.5 * 1.6 * 2^(3.5/3)) = 16
constexpr int SORI_N = (2 * SORI_R + 1) * (2 * SORI_R + 1);

struct SiftTaps { float k[SMAXTAPS]; int n; };

struct SiftArgs {
  EvhSiftGeom g;
  float* pyr; int64_t pyr_frame_floats;      // [group][frame_floats]
  float* tmp; int64_t tmp_frame_floats;      // [group][octave-0 layer]
  uint32_t* cand; int* ncand; int cand_cap;  // [F][cand_cap], [F]
  float* raw; int* nraw;                     // [F][cap][8], [F]
  float* srt;                                // [F][cap][8] sorted
  float* kp;                                 // [F][cap][8] final: x, y, size, angle, response, octave bits
  float* xy; uint8_t* desc; int* count; int* flags;
  int cap;
};

__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
  return p;
}

__device__ __forceinline__ const float* layer_ptr(const SiftArgs& A, int gf, int o, int l) {
  return A.pyr + (int64_t)gf * A.pyr_frame_floats + A.g.ooff[o] + (int64_t)l * A.g.os[o] * A.g.oh[o];
}

// ---- octave 0, layer "pre-blur": resize(gray, 2w x 2h, INTER_LINEAR) on float ------------------------------------
__global__ __launch_bounds__(256) void k_sift_upsample(const uint8_t* __restrict__ gray, int64_t gray_frame_bytes, int gstride,
                                                       int sw, int sh, float* __restrict__ dst, int64_t dst_frame_floats,
                                                       int dstride, int f0) {
  const int f = blockIdx.z, dy = blockIdx.y;
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dw = 2 * sw;
  if (dx >= dw) return;
  const uint8_t* S = gray + (int64_t)(f0 + f) * gray_frame_bytes;
  // fx = (float)((dx + 0.5) * scale - 0.5), scale = 1 / (dw / sw) = 0.5
  float fx = (float)(((double)dx + 0.5) * 0.5 - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  bool edge = false;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx + 1 >= sw) { edge = true; if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; } }
  float fy = (float)(((double)dy + 0.5) * 0.5 - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  const uint8_t* r0 = S + (int64_t)y0 * gstride;
  const uint8_t* r1 = S + (int64_t)y1 * gstride;
  const int sx1 = min(sx + 1, sw - 1);
  float d0, d1;
  if (!edge) {
    d0 = (float)r0[sx] * a0 + (float)r0[sx1] * a1;
    d1 = (float)r1[sx] * a0 + (float)r1[sx1] * a1;
  } else {
    d0 = (float)r0[sx] * 1.f;
    d1 = (float)r1[sx] * 1.f;
  }
  dst[(int64_t)f * dst_frame_floats + (int64_t)dy * dstride + dx] = d0 * b0 + d1 * b1;
}

// ---- GaussianBlur on float: row pass  s = k0*S0; s = fma(kj, Sj, s)  (left to right) --------------------------------------
// The reference's OpenCV build runs its 8-lane FMA row body over columns [0, w & ~7) and the plain scalar loop (multiply, then
// add) over the rest; the column pass likewise fuses [0, w & ~15) only.  The reference's recorded run is reproduced to the last
// bit with exactly this split (oracle/evz_sift.cpp gaussian_blur, tests/test_capture_golden.py), so it is part of the result.
__global__ __launch_bounds__(256) void k_sift_blur_row(const float* __restrict__ src, int64_t src_frame, float* __restrict__ dst,
                                                       int64_t dst_frame, int w, int h, int stride, SiftTaps T) {
  __shared__ float L[256 + SMAXTAPS];
  const int f = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 256, r = T.n >> 1;
  const float* S = src + (int64_t)f * src_frame + (int64_t)y * stride;
  for (int i = threadIdx.x; i < 256 + 2 * r; i += 256) L[i] = S[reflect101(x0 - r + i, w)];
  __syncthreads();
  const int x = x0 + threadIdx.x;
  if (x >= w) return;
  float s = T.k[0] * L[threadIdx.x];
  if (x < (w & ~7)) {
    for (int j = 1; j < T.n; j++) s = fmaf(T.k[j], L[threadIdx.x + j], s);
  } else {
    for (int j = 1; j < T.n; j++) s = __fadd_rn(s, __fmul_rn(T.k[j], L[threadIdx.x + j]));
  }
  dst[(int64_t)f * dst_frame + (int64_t)y * stride + x] = s;
}

// ---- column pass, the symmetric form  s = k_mid*S0; s = fma(k_j, S[+j] + S[-j], s) ---------------------------------------
#define SC_W 64
#define SC_H 32
__global__ __launch_bounds__(256) void k_sift_blur_col(const float* __restrict__ src, int64_t src_frame, float* __restrict__ dst,
                                                       int64_t dst_frame, int w, int h, int stride, SiftTaps T) {
  __shared__ float L[(SC_H + SMAXTAPS) * SC_W];
  const int f = blockIdx.z, x0 = blockIdx.x * SC_W, y0 = blockIdx.y * SC_H, r = T.n >> 1;
  const float* S = src + (int64_t)f * src_frame;
  const int tx = threadIdx.x & (SC_W - 1), ty = threadIdx.x / SC_W;     // 64 x 4
  const int x = x0 + tx;
  for (int i = ty; i < SC_H + 2 * r; i += 4)
    L[i * SC_W + tx] = x < w ? S[(int64_t)reflect101(y0 - r + i, h) * stride + x] : 0.f;
  __syncthreads();
  if (x >= w) return;
  for (int yy = ty; yy < SC_H; yy += 4) {
    const int y = y0 + yy;
    if (y >= h) break;
    const float* c = L + (yy + r) * SC_W + tx;
    float s = T.k[r] * c[0];
    if (x < (w & ~15)) {
      for (int j = 1; j <= r; j++) s = fmaf(T.k[r + j], c[j * SC_W] + c[-j * SC_W], s);
    } else {
      for (int j = 1; j <= r; j++) s = __fadd_rn(s, __fmul_rn(T.k[r + j], c[j * SC_W] + c[-j * SC_W]));
    }
    dst[(int64_t)f * dst_frame + (int64_t)y * stride + x] = s;
  }
}

// ---- next octave: resize(layer nOctaveLayers, (w/2, h/2), INTER_NEAREST) ---------------------------------------------
__global__ __launch_bounds__(256) void k_sift_down(const float* __restrict__ src, float* __restrict__ dst, int64_t frame_floats,
                                                   int sw, int sh, int sstride, int dw, int dh, int dstride, double ifx,
                                                   double ify) {
  const int f = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= dw) return;
  const int sy = min((int)floor((double)y * ify), sh - 1), sx = min((int)floor((double)x * ifx), sw - 1);
  dst[(int64_t)f * frame_floats + (int64_t)y * dstride + x] = src[(int64_t)f * frame_floats + (int64_t)sy * sstride + sx];
}

// ---- scale-space extrema: |D| > threshold and D >= (or <=) all 26 neighbours -----------------------------------------
__global__ __launch_bounds__(256) void k_sift_extrema(SiftArgs A, int o, int f0, int threshold) {
  const int z = blockIdx.z, layer = 1 + z % SL, gf = z / SL;
  const int w = A.g.ow[o], h = A.g.oh[o], st = A.g.os[o];
  const int c = SBORDER + blockIdx.x * 64 + (threadIdx.x & 63), r = SBORDER + blockIdx.y * 4 + (threadIdx.x >> 6);
  bool hit = false;
  if (c < w - SBORDER && r < h - SBORDER) {
    const float* g0 = layer_ptr(A, gf, o, layer - 1);
    const int64_t ls = (int64_t)st * h;
    const float* p = g0 + (int64_t)r * st + c;
    const float val = p[2 * ls] - p[ls];                       // D_layer = G_{layer+1} - G_layer
    if (fabsf(val) > (float)threshold) {
      bool ismax = val > 0, ismin = val < 0;
#pragma unroll
      for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++) {
          const float* q = p + dr * st + dc;
          const float g_0 = q[0], g_1 = q[ls], g_2 = q[2 * ls], g_3 = q[3 * ls];
          const float dprev = g_1 - g_0, dcur = g_2 - g_1, dnext = g_3 - g_2;
          if (!(val >= dcur && val >= dprev && val >= dnext)) ismax = false;
          if (!(val <= dcur && val <= dprev && val <= dnext)) ismin = false;
        }
      hit = ismax || ismin;
    }
  }
  const unsigned long long m = __ballot(hit);
  if (m) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(A.ncand + f0 + gf, __popcll(m));
    base = __shfl(base, 0);
    if (hit) {
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < A.cand_cap)
        A.cand[(int64_t)(f0 + gf) * A.cand_cap + slot] = ((uint32_t)o << 28) | ((uint32_t)layer << 26) | ((uint32_t)r << 13) | (uint32_t)c;
    }
  }
}

// ---- cv::hal::exp32f (the float-polynomial body) and 2^x for the key-point size ----------------------------------------
__constant__ float c_exptab[64] = {
#include "sift_exptab.inc"
};
__device__ __forceinline__ float sift_exp32f(float x) {
  const double A0 = .9670371139572337719125840413672004409288e-2;
  const float A4 = (float)(1.000000000000002438532970795181890933776 / A0),
              A3 = (float)(.6931471805521448196800669615864773144641 / A0),
              A2 = (float)(.2402265109513301490103372422686535526573 / A0),
              A1 = (float)(.5550339366753125211915322047004666939128e-1 / A0);
  const double exp_prescale = 1.4426950408889634073599246810019 * 64, exp_postscale = 1. / 64, exp_max_val = 3000. * 64;
  const float minval = (float)(-exp_max_val / exp_prescale), maxval = (float)(exp_max_val / exp_prescale);
  const float xc = fminf(fmaxf(x, minval), maxval);
  const double xd = (double)xc * exp_prescale;
  const int xi = (int)__builtin_rint(xd);
  const float xf = (float)(xd - (double)xi) * (float)exp_postscale;
  int e = (xi >> 6) + 127;
  e = e < 0 ? 0 : e > 255 ? 255 : e;
  const float yf = c_exptab[xi & 63] * __uint_as_float((uint32_t)e << 23);
  float zf = xf + A1;
  zf = zf * xf + A2; zf = zf * xf + A3; zf = zf * xf + A4;
  return zf * yf;
}
__device__ __forceinline__ float sift_exp2(float x) {
  const double xd = (double)x;
  const double fn = __builtin_rint(xd);
  const double r = (xd - fn) * 0.6931471805599453094;
  double p = 1.0 / 6227020800.0;
  const double inv[13] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                          1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
#pragma unroll
  for (int i = 0; i < 13; i++) p = p * r + inv[i];
  return (float)ldexp(p, (int)fn);
}

// D(l, r, c) of octave o: one exact subtraction of two Gaussian layers
struct DogView {
  const float* g; int st; int64_t ls;
  __device__ __forceinline__ float at(int l, int r, int c) const {
    const float* p = g + (int64_t)l * ls + (int64_t)r * st + c;
    return p[ls] - p[0];
  }
};

// ---- one wavefront per extremum: adjustLocalExtrema, calcOrientationHist, one key point per histogram peak ------------
#define SR_WAVES 4
__global__ __launch_bounds__(64 * SR_WAVES) void k_sift_refine(SiftArgs A, int f0) {
  __shared__ float s_val[SR_WAVES][SORI_N];
  // per wave and orientation bin: which samples of the window fall on it (bits set by the samples with LDS atomic ORs, walked
  // in ascending order = raster order by the bin's lane; 36 lanes each testing every sample cost 6 instructions per
  // sample and lane)
  constexpr int SORI_W = (SORI_N + 31) / 32;
  __shared__ unsigned s_map[SR_WAVES][36 * SORI_W];
  __shared__ float s_hist[SR_WAVES][40];
  const int gf = blockIdx.y, f = f0 + gf, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = lane; i < 36 * SORI_W; i += 64) s_map[wv][i] = 0u;
  const int ncand = min(A.ncand[f], A.cand_cap);
  const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale,
              cross_deriv_scale = img_scale * 0.25f;
  for (int ci = blockIdx.x * SR_WAVES + wv; ci < ncand; ci += gridDim.x * SR_WAVES) {      // wave-uniform
    const uint32_t pk = A.cand[(int64_t)f * A.cand_cap + ci];
    const int octv = (int)(pk >> 28);
    int layer = (int)((pk >> 26) & 3u), r = (int)((pk >> 13) & 0x1FFFu), c = (int)(pk & 0x1FFFu);
    const int w = A.g.ow[octv], h = A.g.oh[octv], st = A.g.os[octv];
    DogView D{layer_ptr(A, gf, octv, 0), st, (int64_t)st * h};
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int it = 0;
    bool ok = true;
    for (; it < 5; it++) {                                  // SIFT_MAX_INTERP_STEPS; every lane runs the same scalar fit
      const float vc = D.at(layer, r, c);
      const float dD0 = (D.at(layer, r, c + 1) - D.at(layer, r, c - 1)) * deriv_scale;
      const float dD1 = (D.at(layer, r + 1, c) - D.at(layer, r - 1, c)) * deriv_scale;
      const float dD2 = (D.at(layer + 1, r, c) - D.at(layer - 1, r, c)) * deriv_scale;
      const float v2 = vc * 2;
      const float dxx = (D.at(layer, r, c + 1) + D.at(layer, r, c - 1) - v2) * second_deriv_scale;
      const float dyy = (D.at(layer, r + 1, c) + D.at(layer, r - 1, c) - v2) * second_deriv_scale;
      const float dss = (D.at(layer + 1, r, c) + D.at(layer - 1, r, c) - v2) * second_deriv_scale;
      const float dxy = (D.at(layer, r + 1, c + 1) - D.at(layer, r + 1, c - 1) - D.at(layer, r - 1, c + 1) + D.at(layer, r - 1, c - 1)) * cross_deriv_scale;
      const float dxs = (D.at(layer + 1, r, c + 1) - D.at(layer + 1, r, c - 1) - D.at(layer - 1, r, c + 1) + D.at(layer - 1, r, c - 1)) * cross_deriv_scale;
      const float dys = (D.at(layer + 1, r + 1, c) - D.at(layer + 1, r - 1, c) - D.at(layer - 1, r + 1, c) + D.at(layer - 1, r - 1, c)) * cross_deriv_scale;
      // Matx33f::solve(dD, DECOMP_LU) = Cramer's rule in float
      const float a00 = dxx, a01 = dxy, a02 = dxs, a10 = dxy, a11 = dyy, a12 = dys, a20 = dxs, a21 = dys, a22 = dss;
      float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
      float X0 = 0, X1 = 0, X2 = 0;
      if (d != 0) {
        d = 1 / d;
        X0 = d * (dD0 * (a11 * a22 - a12 * a21) - a01 * (dD1 * a22 - a12 * dD2) + a02 * (dD1 * a21 - a11 * dD2));
        X1 = d * (a00 * (dD1 * a22 - a12 * dD2) - dD0 * (a10 * a22 - a12 * a20) + a02 * (a10 * dD2 - dD1 * a20));
        X2 = d * (a00 * (a11 * dD2 - dD1 * a21) - a01 * (a10 * dD2 - dD1 * a20) + dD0 * (a10 * a21 - a11 * a20));
      }
      xi = -X2; xr = -X1; xc = -X0;
      if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
      const float big = (float)(2147483647 / 3);
      if (fabsf(xi) > big || fabsf(xr) > big || fabsf(xc) > big) { ok = false; break; }
      c += (int)rintf(xc); r += (int)rintf(xr); layer += (int)rintf(xi);
      if (layer < 1 || layer > SL || c < SBORDER || c >= w - SBORDER || r < SBORDER || r >= h - SBORDER) { ok = false; break; }
    }
    if (!ok || it >= 5) continue;
    {
      const float dD0 = (D.at(layer, r, c + 1) - D.at(layer, r, c - 1)) * deriv_scale;
      const float dD1 = (D.at(layer, r + 1, c) - D.at(layer, r - 1, c)) * deriv_scale;
      const float dD2 = (D.at(layer + 1, r, c) - D.at(layer - 1, r, c)) * deriv_scale;
      float t = 0;
      t += dD0 * xc; t += dD1 * xr; t += dD2 * xi;
      const float vc = D.at(layer, r, c);
      contr = vc * img_scale + t * 0.5f;
      if (fabsf(contr) * SL < 0.04f) continue;
      const float v2 = vc * 2.f;
      const float dxx = (D.at(layer, r, c + 1) + D.at(layer, r, c - 1) - v2) * second_deriv_scale;
      const float dyy = (D.at(layer, r + 1, c) + D.at(layer, r - 1, c) - v2) * second_deriv_scale;
      const float dxy = (D.at(layer, r + 1, c + 1) - D.at(layer, r + 1, c - 1) - D.at(layer, r - 1, c + 1) + D.at(layer, r - 1, c - 1)) * cross_deriv_scale;
      const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
      const float et = 10.f;
      if (det <= 0 || tr * tr * et >= (et + 1) * (et + 1) * det) continue;
    }
    const float kx = ((float)c + xc) * (float)(1 << octv);
    const float ky = ((float)r + xr) * (float)(1 << octv);
    const int koct = octv + (layer << 8) + ((int)__builtin_rint(((double)xi + 0.5) * 255) << 16);
    const float ksize = 1.6f * sift_exp2(((float)layer + xi) / (float)SL) * (float)(1 << octv) * 2;
    const float kresp = fabsf(contr);
    // ---- calcOrientationHist on Gaussian layer `layer`: 36 bins, each accumulated in raster order by one lane
    const float scl_octv = ksize * 0.5f / (float)(1 << octv);
    const int radius = min((int)rintf(4.5f * scl_octv), SORI_R);
    const float sigma = 1.5f * scl_octv;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    const float* G = layer_ptr(A, gf, octv, layer);
    const int side = 2 * radius + 1, len = side * side;
    for (int k = lane; k < len; k += 64) {
      const int i = k / side - radius, j = k % side - radius;
      const int y = r + i, x = c + j;
      int bin = -1; float v = 0.f;
      if (y > 0 && y < h - 1 && x > 0 && x < w - 1) {
        const float* p = G + (int64_t)y * st + x;
        const float dx = p[1] - p[-1];
        const float dy = p[-st] - p[st];
        const float W = sift_exp32f((float)(i * i + j * j) * expf_scale);
        const float Ori = fast_atan2_deg(dy, dx);
        const float Mag = sqrtf(dx * dx + dy * dy);
        bin = (int)rintf((36 / 360.f) * Ori);
        if (bin >= 36) bin -= 36;
        if (bin < 0) bin += 36;
        v = W * Mag;
      }
      s_val[wv][k] = v;
      if (bin >= 0) atomicOr(&s_map[wv][bin * SORI_W + (k >> 5)], 1u << (k & 31));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float th = 0.f;
    if (lane < 36) {
      const int nw = (len + 31) >> 5;
      for (int w = 0; w < nw; w++) {
        unsigned bits = s_map[wv][lane * SORI_W + w];
        if (bits) s_map[wv][lane * SORI_W + w] = 0u;        // clean for the next candidate
        while (bits) {
          const int k = w * 32 + (__ffs((int)bits) - 1);
          bits &= bits - 1u;
          th += s_val[wv][k];
        }
      }
    }
    // temphist[-2..37] with the circular padding, then the 1-4-6-4-1 smoothing
    if (lane < 36) s_hist[wv][lane + 2] = th;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float hv = -INFINITY, hl = 0, hr = 0;
    if (lane < 36) {
      auto TH = [&](int i) { return s_hist[wv][((i + 36) % 36) + 2]; };
      auto HS = [&](int i) {
        return (TH(i - 2) + TH(i + 2)) * (1.f / 16.f) + (TH(i - 1) + TH(i + 1)) * (4.f / 16.f) + TH(i) * (6.f / 16.f);
      };
      hv = HS(lane); hl = HS(lane > 0 ? lane - 1 : 35); hr = HS(lane < 35 ? lane + 1 : 0);
    }
    float omax = hv;
    for (int s = 32; s > 0; s >>= 1) omax = fmaxf(omax, __shfl_xor(omax, s));
    const float mag_thr = omax * 0.8f;
    const bool peak = lane < 36 && hv > hl && hv > hr && hv >= mag_thr;
    const unsigned long long pm = __ballot(peak);
    __builtin_amdgcn_wave_barrier();                         // s_map / s_val / s_hist are free for the next candidate
    if (pm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(A.nraw + f, __popcll(pm));
      base = __shfl(base, 0);
      if (peak) {
        float bin = (float)lane + 0.5f * (hl - hr) / (hl - 2 * hv + hr);
        bin = bin < 0 ? 36 + bin : bin >= 36 ? bin - 36 : bin;
        float ang = 360.f - (float)((360.f / 36) * bin);
        if (fabsf(ang - 360.f) < FLT_EPSILON) ang = 0.f;
        const int slot = base + __popcll(pm & ((1ull << lane) - 1ull));
        if (slot < A.cap) {
          float* o = A.raw + ((int64_t)f * A.cap + slot) * 8;
          o[0] = kx; o[1] = ky; o[2] = ksize; o[3] = ang; o[4] = kresp; o[5] = __int_as_float(koct); o[6] = 0.f; o[7] = 0.f;
        }
      }
    }
  }
}

// ---- KeyPointsFilter::removeDuplicatedSorted -----------------------------------------------------------------------------
// order: x, y, size (descending), angle, response (descending), octave (descending); rank by counting
struct KpRec { float x, y, size, angle, resp; int oct; };
__device__ __forceinline__ bool kp_less(const KpRec& a, const KpRec& b) {
  if (a.x != b.x) return a.x < b.x;
  if (a.y != b.y) return a.y < b.y;
  if (a.size != b.size) return a.size > b.size;
  if (a.angle != b.angle) return a.angle < b.angle;
  if (a.resp != b.resp) return a.resp > b.resp;
  if (a.oct != b.oct) return a.oct > b.oct;
  return false;
}
__global__ __launch_bounds__(256) void k_sift_rank(SiftArgs A, int f0) {
  // rank = number of key points that sort before this one.  x decides almost every comparison: a tile of x values goes
  // through LDS, the full record of the other key point is fetched only where x is equal (several orientations of one
  // extremum).  (Full records through LDS and two full comparisons per pair: 33.6 ms per 33 frames of 25 800 key points.)
  __shared__ float T[2048];
  const int f = f0 + blockIdx.y;
  const int n = A.nraw[f];
  if (n > A.cap || A.ncand[f] > A.cand_cap) return;        // flagged by k_sift_dedup
  if ((int)blockIdx.x * 256 >= n) return;
  const float* R = A.raw + (int64_t)f * A.cap * 8;
  const int i = blockIdx.x * 256 + threadIdx.x;
  KpRec me{0, 0, 0, 0, 0, 0};
  if (i < n) { const float* p = R + (int64_t)i * 8; me = KpRec{p[0], p[1], p[2], p[3], p[4], __float_as_int(p[5])}; }
  int rank = 0;
  auto tie = [&](int j) {                                  // equal x: the rest of the comparator, then the index
    const float* p = R + (int64_t)j * 8;
    const KpRec o{p[0], p[1], p[2], p[3], p[4], __float_as_int(p[5])};
    return kp_less(o, me) || (!kp_less(me, o) && j < i);
  };
  for (int t0 = 0; t0 < n; t0 += 2048) {
    const int tn = min(2048, n - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < tn; k += 256) T[k] = R[(int64_t)(t0 + k) * 8];
    __syncthreads();
    if (i < n) {
      int j = 0;
      for (; j + 8 <= tn; j += 8) {
        float ox[8];
#pragma unroll
        for (int u = 0; u < 8; u++) ox[u] = T[j + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (ox[u] < me.x) rank++;
          else if (ox[u] == me.x && tie(t0 + j + u)) rank++;
        }
      }
      for (; j < tn; j++) {
        const float ox = T[j];
        if (ox < me.x) rank++;
        else if (ox == me.x && tie(t0 + j)) rank++;
      }
    }
  }
  if (i < n) {
    float* o = A.srt + ((int64_t)f * A.cap + rank) * 8;
    o[0] = me.x; o[1] = me.y; o[2] = me.size; o[3] = me.angle; o[4] = me.resp; o[5] = __int_as_float(me.oct);
  }
}
// drop every key point equal to its predecessor in (x, y, size, angle); then the firstOctave = -1 rescale of
// detectAndCompute: octave field, pt *= 0.5, size *= 0.5
__global__ __launch_bounds__(1024) void k_sift_dedup(SiftArgs A, int f0) {
  __shared__ int wtot[16];
  __shared__ int s_base;
  const int f = f0 + blockIdx.x;
  const int n = A.nraw[f];
  if (n > A.cap || A.ncand[f] > A.cand_cap) {
    if (threadIdx.x == 0) { A.count[f] = 0; A.flags[f] = 1; }
    return;
  }
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  const float* S = A.srt + (int64_t)f * A.cap * 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    bool keep = false;
    if (i < n) {
      keep = true;
      if (i > 0) {
        const float* a = S + (int64_t)(i - 1) * 8; const float* b = S + (int64_t)i * 8;
        keep = a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3];
      }
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wtot[wv] = __popcll(m);
    __syncthreads();
    int off = s_base, tot = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) off += wtot[k]; tot += wtot[k]; }
    if (keep) {
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      const float* b = S + (int64_t)i * 8;
      int oct = __float_as_int(b[5]);
      oct = (oct & ~255) | ((oct + (-1)) & 255);
      const float scale = 1.f / (float)(1 << 1);
      float* o = A.kp + ((int64_t)f * A.cap + slot) * 8;
      const float x = b[0] * scale, y = b[1] * scale;
      o[0] = x; o[1] = y; o[2] = b[2] * scale; o[3] = b[3]; o[4] = b[4]; o[5] = __int_as_float(oct);
      A.xy[((int64_t)f * A.cap + slot) * 2] = x; A.xy[((int64_t)f * A.cap + slot) * 2 + 1] = y;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) { A.count[f] = s_base; A.flags[f] = 0; }
}

// ---- calcSIFTDescriptor: one workgroup per key point -------------------------------------------------------------------------
// bins (r, c, o), r, c in 1..4 of the padded 6 x 6 grid, o in 0..9: thread t < 160 owns one bin and adds, sample after
// sample in raster order, the one of the eight trilinear shares that falls on it
__global__ __launch_bounds__(256) void k_sift_desc(SiftArgs A, int f0) {
  __shared__ float s_list[256 * 8];             // the chunk's shares, grouped by bin, in sample order inside a bin
  __shared__ unsigned short s_pref[160 * 8];     // per bin and 32-sample word: shares of the bin in the words before
  __shared__ int s_binbase[160];
  __shared__ int s_wtot[4];
  __shared__ int s_rowlo[256], s_rowoff[257];    // per window row (at most 77 by the operator's constants; sized for 255):
                                                 // first j of its interval, samples in the rows before
  __shared__ float s_hist[160];
  __shared__ float s_dst[128];
  __shared__ float s_scale;
  __shared__ unsigned s_map[160 * 8];            // per bin: which of the chunk's 256 samples fall on it
  const int gf = blockIdx.y, f = f0 + gf, ki = blockIdx.x, tid = threadIdx.x;
  if (ki >= A.count[f]) return;
  for (int i = tid; i < 160 * 8; i += 256) s_map[i] = 0u;
  __syncthreads();                                             // before any sample ORs its bits in
  const float* rec = A.kp + ((int64_t)f * A.cap + ki) * 8;
  const int koct = __float_as_int(rec[5]);
  int oct = koct & 255;
  const int layer = (koct >> 8) & 255;
  oct = oct < 128 ? oct : (-128 | oct);
  const float scale = oct >= 0 ? 1.f / (float)(1 << oct) : (float)(1 << -oct);
  const float size = rec[2] * scale;
  const float ptx = rec[0] * scale, pty = rec[1] * scale;
  const int o = oct + 1;
  const int cols = A.g.ow[o], rows = A.g.oh[o], st = A.g.os[o];
  const float* img = layer_ptr(A, gf, o, layer);
  float ori = 360.f - rec[3];
  if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
  const float scl = size * 0.5f;
  const int d = 4, n = 8;
  const int px = (int)rintf(ptx), py = (int)rintf(pty);
  double sd, cd;
  det_sincos((double)(ori * (float)(3.14159265358979323846 / 180)), &sd, &cd);
  float cos_t = (float)cd, sin_t = (float)sd;
  const float bins_per_rad = n / 360.f;
  const float exp_scale = -1.f / (d * d * 0.5f);
  const float hist_width = 3.f * scl;
  int radius = (int)rintf(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
  radius = min(radius, (int)sqrt(((double)cols) * cols + ((double)rows) * rows));
  cos_t /= hist_width; sin_t /= hist_width;
  radius = min(radius, 127);                             // (never reached: the operator's constants bound it by 38)
  const int side = 2 * radius + 1;
  // The samples that can pass the window test form, in every row i, ONE interval of j: -2.5 < j cos - i sin < 2.5 and
  // -2.5 < j sin + i cos < 2.5 are two strips (the rotated 5 x 5 bin square covers half of its bounding square), and the
  // image bounds are intervals too.  Each row's interval is computed once, widened by two pixels on both sides -- it only has
  // to be a SUPERSET, every sample is still put through the operator's own test below -- and the chunks run over the
  // concatenated intervals in raster order: about 55 % of the window instead of all of it.
  if (tid < side) {
    const int i = tid - radius;
    float lo = (float)-radius, hi = (float)radius;
    const float is = (float)i * sin_t, ic = (float)i * cos_t;
    if (fabsf(cos_t) > 1e-6f) {
      const float a0 = (is - 2.5f) / cos_t, a1 = (is + 2.5f) / cos_t;
      lo = fmaxf(lo, fminf(a0, a1)); hi = fminf(hi, fmaxf(a0, a1));
    }
    if (fabsf(sin_t) > 1e-6f) {
      const float a0 = (-2.5f - ic) / sin_t, a1 = (2.5f - ic) / sin_t;
      lo = fmaxf(lo, fminf(a0, a1)); hi = fminf(hi, fmaxf(a0, a1));
    }
    int jl = max((int)floorf(lo) - 2, -radius), jh = min((int)ceilf(hi) + 2, radius);
    jl = max(jl, 1 - px); jh = min(jh, cols - 2 - px);
    const int r = py + i;
    const int cnt = (r > 0 && r < rows - 1 && jh >= jl) ? jh - jl + 1 : 0;
    s_rowlo[tid] = jl; s_rowoff[tid] = cnt;
  }
  __syncthreads();
  {                                                      // exclusive prefix over the rows
    const int cnt = tid < side ? s_rowoff[tid] : 0;
    int incl = cnt;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) { const int v = __shfl_up(incl, sh); if ((tid & 63) >= sh) incl += v; }
    if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
    __syncthreads();
    const int excl = incl - cnt + (tid >= 64 ? s_wtot[0] : 0) + (tid >= 128 ? s_wtot[1] : 0) + (tid >= 192 ? s_wtot[2] : 0);
    if (tid < side) s_rowoff[tid] = excl;
    if (tid == side - 1) s_rowoff[side] = excl + cnt;
  }
  __syncthreads();
  const int total = s_rowoff[side];
  float acc = 0.f;
  for (int k0 = 0; k0 < total; k0 += 256) {
    const int k = k0 + tid;
    int base = -1;
    float vv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < total) {
      int rlo = 0, rhi = side;                           // the row whose interval holds sample k: s_rowoff[rlo] <= k < s_rowoff[rlo + 1]
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int mid = (rlo + rhi) >> 1;
        if (rhi - rlo > 1) { if (s_rowoff[mid] <= k) rlo = mid; else rhi = mid; }
      }
      const int i = rlo - radius, j = s_rowlo[rlo] + (k - s_rowoff[rlo]);
      const float c_rot = (float)j * cos_t - (float)i * sin_t;
      const float r_rot = (float)j * sin_t + (float)i * cos_t;
      float rbin = r_rot + (float)(d / 2) - 0.5f;
      float cbin = c_rot + (float)(d / 2) - 0.5f;
      const int r = py + i, c = px + j;
      if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
        const float* p = img + (int64_t)r * st + c;
        const float dx = p[1] - p[-1];
        const float dy = p[-st] - p[st];
        const float Ori = fast_atan2_deg(dy, dx);
        const float Mag = sqrtf(dx * dx + dy * dy);
        const float W = sift_exp32f((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        float obin = (Ori - ori) * bins_per_rad;
        const float mag = Mag * W;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin -= (float)r0; cbin -= (float)c0; obin -= (float)o0;
        if (o0 < 0) o0 += n;
        if (o0 >= n) o0 -= n;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
        const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
        const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
        const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
        vv[0] = v_rco000; vv[1] = v_rco001; vv[2] = v_rco010; vv[3] = v_rco011;
        vv[4] = v_rco100; vv[5] = v_rco101; vv[6] = v_rco110; vv[7] = v_rco111;
        base = ((r0 + 1) << 16) | ((c0 + 1) << 8) | o0;
      }
    }
    // Which samples fall on which bin: a 256-bit map per bin, set by the samples themselves with LDS atomic ORs (order-free,
    // so the result is deterministic).  Round 4: the bin threads no longer walk their maps through two dependent LDS reads
    // per hit (the busiest bin of a chunk -- key-point orientation = dominant gradient direction, so a few bins take most of
    // the samples -- set the latency of the whole chunk, ~36 000 cycles).  Instead the maps are turned into RANKS: a bin's
    // population per 32-sample word and a prefix over the 160 bins give every (sample, corner) the slot of its share in a
    // per-bin list, in sample order = raster order; the bin thread then adds a contiguous list with all loads in flight.
    if (base >= 0) {
      const int rb = base >> 16, cb = (base >> 8) & 255, ob = base & 255;      // r0 + 1, c0 + 1, o0
      const unsigned word = (unsigned)tid >> 5, bit = 1u << (tid & 31);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int r = rb + (q >> 2), c = cb + ((q >> 1) & 1), o = ob + (q & 1);
        if (r >= 1 && r <= 4 && c >= 1 && c <= 4) atomicOr(&s_map[((r - 1) * 40 + (c - 1) * 10 + o) * 8 + word], bit);
      }
    }
    __syncthreads();
    int total = 0;
    if (tid < 160) {                                     // population of the bin's words, exclusive prefix inside the bin
#pragma unroll
      for (int w = 0; w < 8; w++) {
        const unsigned m = s_map[tid * 8 + w];
        s_pref[tid * 8 + w] = (unsigned short)total;
        total += __popc(m);
      }
    }
    // exclusive prefix of `total` over the 160 bins (waves 0..2): wave scan, then the wave totals
    int incl = total;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) { const int v = __shfl_up(incl, sh); if ((tid & 63) >= sh) incl += v; }
    if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
    __syncthreads();
    int mybase = incl - total;
    if (tid >= 64) mybase += s_wtot[0];
    if (tid >= 128) mybase += s_wtot[1];
    if (tid < 160) s_binbase[tid] = mybase;
    __syncthreads();
    if (base >= 0) {
      const int rb = base >> 16, cb = (base >> 8) & 255, ob = base & 255;
      const unsigned word = (unsigned)tid >> 5, below = (1u << (tid & 31)) - 1u;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int r = rb + (q >> 2), c = cb + ((q >> 1) & 1), o = ob + (q & 1);
        if (r >= 1 && r <= 4 && c >= 1 && c <= 4) {
          const int bin = (r - 1) * 40 + (c - 1) * 10 + o;
          const int pos = s_binbase[bin] + (int)s_pref[bin * 8 + word] + __popc(s_map[bin * 8 + word] & below);
          s_list[pos] = vv[q];
        }
      }
    }
    __syncthreads();
    if (tid < 160) {
#pragma unroll
      for (int w = 0; w < 8; w++) s_map[tid * 8 + w] = 0u;                          // clean for the next chunk
      const float* L = s_list + mybase;
      int i = 0;
      for (; i + 8 <= total; i += 8) {                   // all eight words requested before the first addition
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = L[i + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
      }
      for (; i < total; i++) acc += L[i];
    }
    __syncthreads();
  }
  if (tid < 160) s_hist[tid] = acc;
  __syncthreads();
  // circular orientation bins: hist[0] += hist[8]; hist[1] += hist[9]
  if (tid < 128) {
    const int cell = tid >> 3, k = tid & 7;
    float v = s_hist[cell * 10 + k];
    if (k < 2) v += s_hist[cell * 10 + k + 8];
    s_dst[tid] = v;
  }
  __syncthreads();
  if (tid < 64) {                                            // wave 0: the two sequential norms, values broadcast by readlane
    const float a = s_dst[tid], b = s_dst[tid + 64];
    float nrm2 = 0;
#pragma unroll
    for (int k = 0; k < 64; k++) { const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), k)); nrm2 += v * v; }
#pragma unroll
    for (int k = 0; k < 64; k++) { const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), k)); nrm2 += v * v; }
    const float thr = sqrtf(nrm2) * 0.2f;
    const float a2 = fminf(a, thr), b2 = fminf(b, thr);
    nrm2 = 0;
#pragma unroll
    for (int k = 0; k < 64; k++) { const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), k)); nrm2 += v * v; }
#pragma unroll
    for (int k = 0; k < 64; k++) { const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b2), k)); nrm2 += v * v; }
    const float fs = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    s_dst[tid] = a2; s_dst[tid + 64] = b2;
    if (tid == 0) s_scale = fs;
  }
  __syncthreads();
  if (tid < 128) {
    const int v = (int)rintf(s_dst[tid] * s_scale);
    A.desc[((int64_t)f * A.cap + ki) * 128 + tid] = (uint8_t)min(max(v, 0), 255);
  }
}

// getGaussianKernel(n, sigma, CV_32F), n = cvRound(sigma*4*2 + 1) | 1
SiftTaps gauss_taps(double sigma) {
  SiftTaps T{};
  const int n = (int)std::lrint(sigma * 4 * 2 + 1) | 1;
  T.n = std::min(n, SMAXTAPS);
  const double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < T.n; i++) {
    const double x = i - (n - 1) * 0.5;
    T.k[i] = (float)std::exp(scale2X * x * x);
    sum += T.k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < T.n; i++) T.k[i] = (float)(T.k[i] * sum);
  return T;
}

void sift_geometry(int w, int h, EvhSiftGeom& g) {
  g.w = w; g.h = h;
  const int bw = 2 * w, bh = 2 * h;
  g.noct = (int)std::lrint(std::log((double)std::min(bw, bh)) / std::log(2.) - 2) + 1;
  g.noct = std::max(1, std::min(g.noct, EVH_SIFT_MAXOCT));
  int cw = bw, ch = bh;
  int64_t off = 0;
  for (int o = 0; o < g.noct; o++) {
    g.ow[o] = cw; g.oh[o] = ch; g.os[o] = (cw + 15) & ~15;
    g.ooff[o] = off;
    off += (int64_t)SNG * g.os[o] * ch;
    cw /= 2; ch /= 2;
    if (cw < 1 || ch < 1) { g.noct = o + 1; break; }
  }
  g.frame_floats = off;
  g.tmp_floats = (int64_t)g.os[0] * g.oh[0];
}

template <class T>
int salloc(evh_ctx* c, T** p, size_t n) {
  EVH_HIP(c, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  c->bytes_allocated += n * sizeof(T);
  return EVH_SUCCESS;
}

SiftArgs sift_args(evh_ctx* c) {
  SiftArgs A{};
  A.g = c->sg;
  A.pyr = c->d_sift_pyr; A.pyr_frame_floats = c->sift_pyr_frame_floats;
  A.tmp = c->d_sift_tmp; A.tmp_frame_floats = c->sift_tmp_frame_floats;
  A.cand = c->d_sift_cand; A.ncand = c->d_sift_ncand; A.cand_cap = c->sift_cand_cap;
  A.raw = c->d_sift_raw; A.nraw = c->d_sift_nraw; A.srt = c->d_sift_srt; A.kp = c->d_sift_kp;
  A.xy = c->d_sift_xy; A.desc = c->d_sift_desc; A.count = c->d_sift_count; A.flags = c->d_sift_flags;
  A.cap = c->sift_cap;
  return A;
}

}  // namespace

// allocate the SIFT buffers of a context (once): max_sift_features key points per frame slot, the scale space of a
// group of frames sized for the context's largest frame
int evh_sift_allocate(evh_ctx* c, int max_sift_features) {
  if (c->sift_cap) return max_sift_features <= c->sift_cap ? EVH_SUCCESS
                                                            : evh_fail(c, EVH_ERR_CAPACITY, "evh_sift_enable: already enabled with a smaller capacity");
  if (max_sift_features < 64 || max_sift_features > 65536) return evh_fail(c, EVH_ERR_INVALID, "evh_sift_enable: capacity out of range (64..65536)");
  EvhSiftGeom gm;
  sift_geometry(c->max_w, c->max_h, gm);
  // frames whose scale space is resident at once: at most 8 GiB of pyramid
  const size_t per_frame = (size_t)(gm.frame_floats + gm.tmp_floats) * sizeof(float);
  int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)c->max_frames, ((size_t)8 << 30) / per_frame));
  const int cap = (max_sift_features + 63) & ~63;
  const size_t F = (size_t)c->max_frames;
  int rc;
#define S_(call) if ((rc = (call)) != EVH_SUCCESS) { evh_sift_free(c); return rc; }   /* a partial allocation is released */
  S_(salloc(c, &c->d_sift_pyr, (size_t)group * gm.frame_floats + 64));
  S_(salloc(c, &c->d_sift_tmp, (size_t)group * gm.tmp_floats + 64));
  S_(salloc(c, &c->d_sift_cand, F * 4 * cap));
  S_(salloc(c, &c->d_sift_ncand, F));
  S_(salloc(c, &c->d_sift_raw, F * cap * 8));
  S_(salloc(c, &c->d_sift_nraw, F));
  S_(salloc(c, &c->d_sift_srt, F * cap * 8));
  S_(salloc(c, &c->d_sift_kp, F * cap * 8));
  S_(salloc(c, &c->d_sift_xy, F * cap * 2));
  S_(salloc(c, &c->d_sift_desc, F * cap * 128));
  S_(salloc(c, &c->d_sift_count, F));
  S_(salloc(c, &c->d_sift_flags, F));
#undef S_
  EVH_HIP(c, hipMemsetAsync(c->d_sift_count, 0, F * sizeof(int), c->stream));
  EVH_HIP(c, hipMemsetAsync(c->d_sift_flags, 0, F * sizeof(int), c->stream));
  c->sift_cap = cap; c->sift_cand_cap = 4 * cap; c->sift_group = group;
  c->sift_pyr_frame_floats = gm.frame_floats; c->sift_tmp_frame_floats = gm.tmp_floats;
  return EVH_SUCCESS;
}

void evh_sift_free(evh_ctx* c) {
  void* ptrs[] = {c->d_sift_pyr, c->d_sift_tmp, c->d_sift_cand, c->d_sift_ncand, c->d_sift_raw, c->d_sift_nraw, c->d_sift_srt,
                  c->d_sift_kp, c->d_sift_xy, c->d_sift_desc, c->d_sift_count, c->d_sift_flags};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  c->d_sift_pyr = nullptr; c->d_sift_tmp = nullptr; c->d_sift_cand = nullptr; c->d_sift_ncand = nullptr; c->d_sift_raw = nullptr;
  c->d_sift_nraw = nullptr; c->d_sift_srt = nullptr; c->d_sift_kp = nullptr; c->d_sift_xy = nullptr; c->d_sift_desc = nullptr;
  c->d_sift_count = nullptr; c->d_sift_flags = nullptr; c->sift_cap = 0;
}

// SIFT on the frames whose gray level 0 is resident in the context's ORB pyramid (evh_launch_gray_level0 /
// evh_launch_ingest_level0 ran for (w, h)): scale space, extrema, key points, descriptors
int evh_launch_sift(evh_ctx* c, int nframes, int w, int h) {
  if (!c->sift_cap) return evh_fail(c, EVH_ERR_INVALID, "SIFT is not enabled on this context (evh_sift_enable)");
  if (w >= 4096 || h >= 4096) return evh_fail(c, EVH_ERR_UNSUPPORTED, "SIFT: frames must be smaller than 4096 in each dimension");
  EvhSiftGeom g;
  sift_geometry(w, h, g);
  if (g.frame_floats > c->sift_pyr_frame_floats || g.tmp_floats > c->sift_tmp_frame_floats)
    return evh_fail(c, EVH_ERR_CAPACITY, "SIFT: frame larger than the size given to evh_create");
  c->sg = g; c->sift_geom_valid = true;
  SiftArgs A = sift_args(c);
  hipStream_t s = c->stream;
  // sigma schedule of buildGaussianPyramid
  double sig[SNG];
  sig[0] = 1.6;
  const double k = std::pow(2., 1. / SL);
  for (int i = 1; i < SNG; i++) {
    const double sig_prev = std::pow(k, (double)(i - 1)) * 1.6, sig_total = sig_prev * k;
    sig[i] = std::sqrt(sig_total * sig_total - sig_prev * sig_prev);
  }
  const float sigma = 1.6f;
  const float sig_diff = sqrtf(std::max(sigma * sigma - 0.5f * 0.5f * 4, 0.01f));
  SiftTaps T0 = gauss_taps((double)sig_diff), TL[SNG];
  for (int i = 1; i < SNG; i++) TL[i] = gauss_taps(sig[i]);
  const int threshold = (int)std::floor(0.5 * 0.04 / SL * 255);
  const EvhLevel& L0 = c->g.lv[0];
  EVH_HIP(c, hipMemsetAsync(c->d_sift_ncand, 0, sizeof(int) * (size_t)nframes, s));
  EVH_HIP(c, hipMemsetAsync(c->d_sift_nraw, 0, sizeof(int) * (size_t)nframes, s));
  auto blur = [&](const float* src, float* dst, int o, int ng, const SiftTaps& T) {
    const int ow = g.ow[o], oh = g.oh[o], os = g.os[o];
    hipLaunchKernelGGL(k_sift_blur_row, dim3((ow + 255) / 256, oh, ng), dim3(256), 0, s, src, c->sift_pyr_frame_floats, c->d_sift_tmp,
                       c->sift_tmp_frame_floats, ow, oh, os, T);
    hipLaunchKernelGGL(k_sift_blur_col, dim3((ow + SC_W - 1) / SC_W, (oh + SC_H - 1) / SC_H, ng), dim3(256), 0, s, c->d_sift_tmp,
                       c->sift_tmp_frame_floats, dst, c->sift_pyr_frame_floats, ow, oh, os, T);
  };
  for (int f0 = 0; f0 < nframes; f0 += c->sift_group) {
    const int ng = std::min(c->sift_group, nframes - f0);
    if (ng > 21845) return evh_fail(c, EVH_ERR_CAPACITY, "SIFT: too many frames in one group");
    auto LP = [&](int o, int l) { return c->d_sift_pyr + g.ooff[o] + (int64_t)l * g.os[o] * g.oh[o]; };
    // octave 0: frame x2 into the slot of layer 1 (overwritten below), blurred by sig_diff into layer 0
    hipLaunchKernelGGL(k_sift_upsample, dim3((2 * w + 255) / 256, 2 * h, ng), dim3(256), 0, s, c->d_pyr + L0.off, c->g.pyr_frame_bytes,
                       L0.stride, w, h, LP(0, 1), c->sift_pyr_frame_floats, g.os[0], f0);
    blur(LP(0, 1), LP(0, 0), 0, ng, T0);
    for (int o = 0; o < g.noct; o++) {
      if (o > 0) {
        const double ifx = 1. / ((double)g.ow[o] / g.ow[o - 1]), ify = 1. / ((double)g.oh[o] / g.oh[o - 1]);
        hipLaunchKernelGGL(k_sift_down, dim3((g.ow[o] + 255) / 256, g.oh[o], ng), dim3(256), 0, s, LP(o - 1, SL), LP(o, 0),
                           c->sift_pyr_frame_floats, g.ow[o - 1], g.oh[o - 1], g.os[o - 1], g.ow[o], g.oh[o], g.os[o], ifx, ify);
      }
      for (int l = 1; l < SNG; l++) blur(LP(o, l - 1), LP(o, l), o, ng, TL[l]);
      const int ew = g.ow[o] - 2 * SBORDER, eh = g.oh[o] - 2 * SBORDER;
      if (ew > 0 && eh > 0)
        hipLaunchKernelGGL(k_sift_extrema, dim3((ew + 63) / 64, (eh + 3) / 4, SL * ng), dim3(256), 0, s, A, o, f0, threshold);
    }
    EVH_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(k_sift_refine, dim3(std::max(1, std::min(c->sift_cand_cap / SR_WAVES, 2048)), ng), dim3(64 * SR_WAVES), 0, s, A, f0);
    hipLaunchKernelGGL(k_sift_rank, dim3((c->sift_cap + 255) / 256, ng), dim3(256), 0, s, A, f0);
    hipLaunchKernelGGL(k_sift_dedup, dim3(ng), dim3(1024), 0, s, A, f0);
    hipLaunchKernelGGL(k_sift_desc, dim3(c->sift_cap, ng), dim3(256), 0, s, A, f0);
    EVH_HIP(c, hipGetLastError());
  }
  c->sift_frames_resident = nframes;
  return EVH_SUCCESS;
}
