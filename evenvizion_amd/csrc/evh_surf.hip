// evh_surf.hip -- N4, second half: SURF detectAndCompute on the GPU, the MI355X counterpart of
//   cv2.xfeatures2d.SURF_create(extended=1, hessianThreshold=400).detectAndCompute(frame, None)
//                                                                   evenvizion/processing/frame_processing.py:65-67
// (opencv-contrib 3.4.2: 4 octaves x 3 layers, 128-float descriptors, rotation-aware).
//
// k_surf_integral_rows / _cols : integral(img, sum, CV_32S), (h+1) x (w+1)
// k_surf_det                   : box-filter Hessian of the 20 layers: det = Dxx*Dyy - 0.81*Dxy^2 and the trace
// k_surf_maxima                : strict 3x3x3 maxima above the threshold + the quadratic interpolation
// k_surf_rank                  : the operator's output order (KeypointGreater: response descending, ...)
// k_surf_compact               : drops the key points the operator marks for deletion (Haar window larger than the
//                                image / no orientation sample inside it) -- a purely geometric test
// k_surf_describe              : one workgroup per key point: dominant orientation (113 Haar samples, 72 sliding
//                                windows), the rotated (21 s)^2 window sampled bilinearly row by row (each thread walks
//                                one row in the operator's own running-sum order) straight into the INTER_AREA sums of
//                                the 21 x 21 patch, Gaussian-weighted gradients, 4 x 4 x 8 sums, unit length.
// Every float / double sum keeps the operator's order (the library is built with -ffp-contract=off), so results can be
// compared bit for bit with the CPU oracle.
#include "evh_internal.h"
#include "evh_devmath.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int SU_OCT = 4, SU_LAY = 3, SU_NL = (SU_LAY + 2) * SU_OCT;   // 20 layers
constexpr int SU_ORI_N = 113;      // grid points with i^2 + j^2 <= 36
constexpr int SU_PATCH = 20;
constexpr int SU_WINMAX = 768;     // largest window side: 21 * (264 * 1.2 / 9) = 739
constexpr int SU_TAPMAX = 40;      // INTER_AREA taps per output column: ceil(739 / 21) + 2

struct SurfBox { int dx1, dy1, dx2, dy2; float w; };
struct SurfLayerTab { SurfBox Dx[3], Dy[3], Dxy[4]; int size, step, rows, cols, built; int64_t off; };
struct SurfTabs { SurfLayerTab L[SU_NL]; float aptw[SU_ORI_N]; int aptx[SU_ORI_N], apty[SU_ORI_N]; float DW[SU_PATCH * SU_PATCH]; };

struct SurfArgs {
  const SurfTabs* T;
  const uint8_t* gray; int64_t gray_frame_bytes; int gstride; int w, h;
  int* sum; int64_t sum_frame_ints; int sstride;             // [group]
  float* det; float* trace; int64_t det_frame_floats;        // [group]
  float* raw; int* nraw; float* srt; int* nsrt;              // [F][cap][8]
  float* kp; float* xy; float* desc; int* count; int* flags; // final
  int cap;
};

// ---- integral image ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_surf_integral_rows(SurfArgs A, int f0) {
  __shared__ int part[256];
  const int gf = blockIdx.y, y = blockIdx.x;                 // y = 0 .. h: row y of sum; row 0 is zero
  int* S = A.sum + (int64_t)gf * A.sum_frame_ints + (int64_t)y * A.sstride;
  if (y == 0) { for (int x = threadIdx.x; x <= A.w; x += 256) S[x] = 0; return; }
  const uint8_t* g = A.gray + (int64_t)(f0 + gf) * A.gray_frame_bytes + (int64_t)(y - 1) * A.gstride;
  const int per = (A.w + 255) / 256, x0 = threadIdx.x * per, x1 = min(x0 + per, A.w);
  int s = 0;
  for (int x = x0; x < x1; x++) s += g[x];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  if (threadIdx.x == 0) S[0] = 0;
  for (int x = x0; x < x1; x++) { run += g[x]; S[x + 1] = run; }
}
__global__ __launch_bounds__(256) void k_surf_integral_cols(SurfArgs A) {
  const int gf = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (x > A.w) return;
  int* S = A.sum + (int64_t)gf * A.sum_frame_ints + x;
  int run = 0;
  for (int y = 1; y <= A.h; y++) { run += S[(int64_t)y * A.sstride]; S[(int64_t)y * A.sstride] = run; }
}

// calcHaarPattern: every box = int sum * float weight (one float product), accumulated in double, returned as float
__device__ __forceinline__ float haar(const int* o, int st, const SurfBox* f, int n) {
  double d = 0;
  for (int k = 0; k < n; k++) {
    const int v = o[f[k].dy1 * st + f[k].dx1] + o[f[k].dy2 * st + f[k].dx2] - o[f[k].dy2 * st + f[k].dx1] - o[f[k].dy1 * st + f[k].dx2];
    d += (double)((float)v * f[k].w);
  }
  return (float)d;
}

// ---- calcLayerDetAndTrace for all 20 layers -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_surf_det(SurfArgs A) {
  const int z = blockIdx.z, li = z % SU_NL, gf = z / SU_NL;
  const SurfLayerTab& L = A.T->L[li];
  if (!L.built) return;
  const int srows = A.h + 1, scols = A.w + 1;
  const int samples_i = 1 + (srows - 1 - L.size) / L.step, samples_j = 1 + (scols - 1 - L.size) / L.step;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= samples_i || j >= samples_j) return;
  const int margin = (L.size / 2) / L.step;
  const int* sp = A.sum + (int64_t)gf * A.sum_frame_ints + (int64_t)(i * L.step) * A.sstride + j * L.step;
  const float dx = haar(sp, A.sstride, L.Dx, 3), dy = haar(sp, A.sstride, L.Dy, 3), dxy = haar(sp, A.sstride, L.Dxy, 4);
  const int64_t o = (int64_t)gf * A.det_frame_floats + L.off + (int64_t)(i + margin) * L.cols + j + margin;
  A.det[o] = dx * dy - 0.81f * dxy * dxy;
  A.trace[o] = dx + dy;
}

// ---- findMaximaInLayer + interpolateKeypoint --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_surf_maxima(SurfArgs A, int f0, float thr) {
  const int z = blockIdx.z, mi = z % (SU_OCT * SU_LAY), gf = z / (SU_OCT * SU_LAY);
  const int octave = mi / SU_LAY, li = octave * (SU_LAY + 2) + 1 + mi % SU_LAY;
  const SurfLayerTab& L = A.T->L[li];
  const SurfLayerTab& Lp = A.T->L[li - 1];
  const SurfLayerTab& Ln = A.T->L[li + 1];
  const int size = L.size, ss = L.step;
  const int layer_rows = A.h / ss, layer_cols = A.w / ss;
  const int margin = (Ln.size / 2) / ss + 1;
  const int j = margin + blockIdx.x * 64 + (threadIdx.x & 63), i = margin + blockIdx.y * 4 + (threadIdx.x >> 6);
  bool hit = false;
  float kx = 0, ky = 0, ksz = 0, val0 = 0; int lap = 0;
  if (i < layer_rows - margin && j < layer_cols - margin) {
    const int st = L.cols;
    const float* base = A.det + (int64_t)gf * A.det_frame_floats;
    const float* d2 = base + L.off + (int64_t)i * st + j;
    val0 = d2[0];
    if (val0 > thr) {
      const float* d1 = base + Lp.off + (int64_t)i * st + j;
      const float* d3 = base + Ln.off + (int64_t)i * st + j;
      float N9[3][9];
      const float* dp[3] = {d1, d2, d3};
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const float* p = dp[d];
        N9[d][0] = p[-st - 1]; N9[d][1] = p[-st]; N9[d][2] = p[-st + 1]; N9[d][3] = p[-1]; N9[d][4] = p[0]; N9[d][5] = p[1];
        N9[d][6] = p[st - 1]; N9[d][7] = p[st]; N9[d][8] = p[st + 1];
      }
      bool mx = true;
#pragma unroll
      for (int d = 0; d < 3; d++)
#pragma unroll
        for (int q = 0; q < 9; q++)
          if (!(d == 1 && q == 4) && !(val0 > N9[d][q])) mx = false;
      if (mx) {
        const int sum_i = ss * (i - (size / 2) / ss), sum_j = ss * (j - (size / 2) / ss);
        const float center_i = (float)sum_i + (float)(size - 1) * 0.5f, center_j = (float)sum_j + (float)(size - 1) * 0.5f;
        const float tr = A.trace[(int64_t)gf * A.det_frame_floats + L.off + (int64_t)i * st + j];
        lap = (tr > 0) - (tr < 0);
        const int ds = size - Lp.size;
        // interpolateKeypoint: A.solve(b, DECOMP_LU) = Cramer's rule in float
        const float b0 = -(N9[1][5] - N9[1][3]) / 2, b1 = -(N9[1][7] - N9[1][1]) / 2, b2 = -(N9[2][4] - N9[0][4]) / 2;
        const float a00 = N9[1][3] - 2 * N9[1][4] + N9[1][5];
        const float a01 = (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4;
        const float a02 = (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4;
        const float a10 = a01;
        const float a11 = N9[1][1] - 2 * N9[1][4] + N9[1][7];
        const float a12 = (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4;
        const float a20 = a02, a21 = a12;
        const float a22 = N9[0][4] - 2 * N9[1][4] + N9[2][4];
        float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
        float x0 = 0, x1 = 0, x2 = 0;
        if (d != 0) {
          d = 1 / d;
          x0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
          x1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
          x2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
        }
        const bool ok = (x0 != 0 || x1 != 0 || x2 != 0) && fabsf(x0) <= 1 && fabsf(x1) <= 1 && fabsf(x2) <= 1;
        if (ok) {
          kx = center_j + x0 * (float)ss;
          ky = center_i + x1 * (float)ss;
          ksz = (float)(int)rintf((float)size + x2 * (float)ds);
          hit = true;
        }
      }
    }
  }
  const unsigned long long m = __ballot(hit);
  if (m) {
    const int lane = threadIdx.x & 63;
    int basei = 0;
    if (lane == 0) basei = atomicAdd(A.nraw + f0 + gf, __popcll(m));
    basei = __shfl(basei, 0);
    if (hit) {
      const int slot = basei + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < A.cap) {
        float* o = A.raw + ((int64_t)(f0 + gf) * A.cap + slot) * 8;
        o[0] = kx; o[1] = ky; o[2] = ksz; o[3] = -1.f; o[4] = val0; o[5] = __int_as_float(octave); o[6] = __int_as_float(lap); o[7] = 0.f;
      }
    }
  }
}

// ---- std::sort(keypoints, KeypointGreater): rank by counting ------------------------------------------------------------------------
struct SKp { float x, y, size, resp; int oct; };
__device__ __forceinline__ bool kp_greater(const SKp& a, const SKp& b) {
  if (a.resp > b.resp) return true;
  if (a.resp < b.resp) return false;
  if (a.size > b.size) return true;
  if (a.size < b.size) return false;
  if (a.oct > b.oct) return true;
  if (a.oct < b.oct) return false;
  if (a.y < b.y) return false;
  if (a.y > b.y) return true;
  return a.x < b.x;
}
__global__ __launch_bounds__(256) void k_surf_rank(SurfArgs A, int f0) {
  // rank = number of key points that sort before this one (KeypointGreater).  The response decides almost every
  // comparison: a tile of responses goes through LDS, the full record is fetched only where they are equal.
  __shared__ float T[2048];
  const int f = f0 + blockIdx.y;
  const int n = A.nraw[f];
  if (n > A.cap) return;
  if ((int)blockIdx.x * 256 >= n) return;
  const float* R = A.raw + (int64_t)f * A.cap * 8;
  const int i = blockIdx.x * 256 + threadIdx.x;
  SKp me{0, 0, 0, 0, 0};
  if (i < n) { const float* p = R + (int64_t)i * 8; me = SKp{p[0], p[1], p[2], p[4], __float_as_int(p[5])}; }
  int rank = 0;
  auto tie = [&](int j) {
    const float* p = R + (int64_t)j * 8;
    const SKp o{p[0], p[1], p[2], p[4], __float_as_int(p[5])};
    return kp_greater(o, me) || (!kp_greater(me, o) && j < i);
  };
  for (int t0 = 0; t0 < n; t0 += 2048) {
    const int tn = min(2048, n - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < tn; k += 256) T[k] = R[(int64_t)(t0 + k) * 8 + 4];
    __syncthreads();
    if (i < n) {
      int j = 0;
      for (; j + 8 <= tn; j += 8) {
        float orr[8];
#pragma unroll
        for (int u = 0; u < 8; u++) orr[u] = T[j + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (orr[u] > me.resp) rank++;
          else if (!(orr[u] < me.resp) && tie(t0 + j + u)) rank++;
        }
      }
      for (; j < tn; j++) {
        const float orr = T[j];
        if (orr > me.resp) rank++;
        else if (!(orr < me.resp) && tie(t0 + j)) rank++;
      }
    }
  }
  if (i < n) {
    const float* p = R + (int64_t)i * 8;
    float* o = A.srt + ((int64_t)f * A.cap + rank) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = p[k];
  }
}

// ---- key points the operator marks for deletion (size = -1): geometry only -----------------------------------------------------------
__device__ __forceinline__ bool surf_keeps(const SurfArgs& A, float cx, float cy, float size) {
  const int srows = A.h + 1, scols = A.w + 1;
  const float s = size * 1.2f / 9.0f;
  const int g = 2 * (int)rintf(2 * s);
  if (srows < g || scols < g) return false;
  for (int kk = 0; kk < SU_ORI_N; kk++) {
    const int x = (int)rintf(cx + (float)A.T->aptx[kk] * s - (float)(g - 1) / 2);
    const int y = (int)rintf(cy + (float)A.T->apty[kk] * s - (float)(g - 1) / 2);
    if (!(y < 0 || y >= srows - g || x < 0 || x >= scols - g)) return true;      // nangle > 0
  }
  return false;
}
__global__ __launch_bounds__(1024) void k_surf_compact(SurfArgs A, int f0) {
  __shared__ int wtot[16];
  __shared__ int s_base;
  const int f = f0 + blockIdx.x;
  const int n = A.nraw[f];
  if (n > A.cap) { if (threadIdx.x == 0) { A.count[f] = 0; A.flags[f] = 1; } return; }
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  const float* S = A.srt + (int64_t)f * A.cap * 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    bool keep = false;
    if (i < n) keep = surf_keeps(A, S[(int64_t)i * 8], S[(int64_t)i * 8 + 1], S[(int64_t)i * 8 + 2]);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wtot[wv] = __popcll(m);
    __syncthreads();
    int off = s_base, tot = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) off += wtot[k]; tot += wtot[k]; }
    if (keep) {
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      const float* b = S + (int64_t)i * 8;
      float* o = A.kp + ((int64_t)f * A.cap + slot) * 8;
#pragma unroll
      for (int k = 0; k < 8; k++) o[k] = b[k];
      A.xy[((int64_t)f * A.cap + slot) * 2] = b[0]; A.xy[((int64_t)f * A.cap + slot) * 2 + 1] = b[1];
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) { A.count[f] = s_base; A.flags[f] = 0; }
}

// resizeHaarPattern for the two-box gradient wavelets (oldSize 4)
__device__ __forceinline__ void grad_boxes(int g, SurfBox* dx_t, SurfBox* dy_t) {
  const int dx_s[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}};
  const int dy_s[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
  const float ratio = (float)g / 4;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    SurfBox a, b;
    a.dx1 = (int)rintf(ratio * dx_s[k][0]); a.dy1 = (int)rintf(ratio * dx_s[k][1]); a.dx2 = (int)rintf(ratio * dx_s[k][2]); a.dy2 = (int)rintf(ratio * dx_s[k][3]);
    a.w = (float)dx_s[k][4] / ((float)(a.dx2 - a.dx1) * (float)(a.dy2 - a.dy1));
    b.dx1 = (int)rintf(ratio * dy_s[k][0]); b.dy1 = (int)rintf(ratio * dy_s[k][1]); b.dx2 = (int)rintf(ratio * dy_s[k][2]); b.dy2 = (int)rintf(ratio * dy_s[k][3]);
    b.w = (float)dy_s[k][4] / ((float)(b.dx2 - b.dx1) * (float)(b.dy2 - b.dy1));
    dx_t[k] = a; dy_t[k] = b;
  }
}

// ---- SURFInvoker: orientation + descriptor, one workgroup per key point --------------------------------------------------------------
struct DescLds {
  float X[SU_ORI_N + 3], Y[SU_ORI_N + 3], ang[SU_ORI_N + 3];
  float wmod[72], wsx[72], wsy[72];
  float start_x[SU_WINMAX], start_y[SU_WINMAX];
  short tsi[21 * SU_TAPMAX]; float tal[21 * SU_TAPMAX]; int tcnt[21];
  unsigned char patch[21 * 21 + 3];
  float DX[SU_PATCH * SU_PATCH], DY[SU_PATCH * SU_PATCH];
  float vec[128];
  int nangle, cnt0;
  float dir, sin_dir, cos_dir, scale;
};

// Two launches per group: key points whose window has at most SU_WINSMALL rows (nearly all of them: sizes up to ~45) with a
// small dynamic LDS -- five workgroups per compute unit instead of one -- and the few larger ones with the full buffer; a
// workgroup whose key point belongs to the other launch leaves at once.
constexpr int SU_WINSMALL = 128;
__global__ __launch_bounds__(256) void k_surf_describe(SurfArgs A, int f0, int win_lo, int win_hi) {
  extern __shared__ float s_rowbuf[];              // [win_size][21] horizontal INTER_AREA sums (float, or int bits on the integer path)
  __shared__ DescLds S;
  const int gf = blockIdx.y, f = f0 + gf, ki = blockIdx.x, tid = threadIdx.x;
  if (ki >= A.count[f]) return;
  float* rec = A.kp + ((int64_t)f * A.cap + ki) * 8;
  const float cx = rec[0], cy = rec[1], size = rec[2];
  const int srows = A.h + 1, scols = A.w + 1;
  const float s = size * 1.2f / 9.0f;
  {
    const int win_class = (int)((float)(SU_PATCH + 1) * s);       // = `win` below; out-of-range windows (never emitted) go with the large class
    const bool small = win_class >= 1 && win_class <= SU_WINSMALL;
    if (small ? win_lo != 0 : win_hi != SU_WINMAX) return;
  }
  const int g = 2 * (int)rintf(2 * s);
  const int* sum = A.sum + (int64_t)gf * A.sum_frame_ints;
  // ---- orientation samples, kept in kk order
  {
    bool valid = false; float vxw = 0, vyw = 0;
    if (tid < SU_ORI_N) {
      const int x = (int)rintf(cx + (float)A.T->aptx[tid] * s - (float)(g - 1) / 2);
      const int y = (int)rintf(cy + (float)A.T->apty[tid] * s - (float)(g - 1) / 2);
      if (!(y < 0 || y >= srows - g || x < 0 || x >= scols - g)) {
        SurfBox dx_t[2], dy_t[2];
        grad_boxes(g, dx_t, dy_t);
        const int* ptr = sum + (int64_t)y * A.sstride + x;
        const float vx = haar(ptr, A.sstride, dx_t, 2), vy = haar(ptr, A.sstride, dy_t, 2);
        vxw = vx * A.T->aptw[tid]; vyw = vy * A.T->aptw[tid];
        valid = true;
      }
    }
    const unsigned long long m = __ballot(valid);
    const int lane = tid & 63, wv = tid >> 6;
    if (wv == 0 && lane == 0) S.cnt0 = __popcll(m);
    __syncthreads();
    if (valid) {
      const int slot = (wv ? S.cnt0 : 0) + __popcll(m & ((1ull << lane) - 1ull));
      S.X[slot] = vxw; S.Y[slot] = vyw; S.ang[slot] = fast_atan2_deg(vyw, vxw);
    }
    if (wv == 1 && lane == 0) S.nangle = S.cnt0 + __popcll(m);
    __syncthreads();
  }
  const int nangle = S.nangle;
  if (tid < 72) {                                   // the 72 sliding windows, each summed in sample order
    const int i = tid * 5;
    float sumx = 0, sumy = 0;
    for (int j = 0; j < nangle; j++) {
      const int d = abs((int)rintf(S.ang[j]) - i);
      if (d < 30 || d > 330) { sumx += S.X[j]; sumy += S.Y[j]; }
    }
    S.wsx[tid] = sumx; S.wsy[tid] = sumy; S.wmod[tid] = sumx * sumx + sumy * sumy;
  }
  __syncthreads();
  if (tid == 0) {
    float bestx = 0, besty = 0, mod = 0;
    for (int i = 0; i < 72; i++)
      if (S.wmod[i] > mod) { mod = S.wmod[i]; bestx = S.wsx[i]; besty = S.wsy[i]; }
    float dir = fast_atan2_deg(-besty, bestx);
    rec[3] = dir;                                   // kp.angle
    dir *= (float)(3.14159265358979323846 / 180);
    double sd, cd;
    det_sincos((double)dir, &sd, &cd);
    S.sin_dir = -(float)sd; S.cos_dir = (float)cd;
  }
  __syncthreads();
  const float sin_dir = S.sin_dir, cos_dir = S.cos_dir;
  const int win = (int)((float)(SU_PATCH + 1) * s);
  if (win < 1 || win > SU_WINMAX) {                 // cannot happen for sizes the detector emits (<= 264)
    for (int k = tid; k < 128; k += 256) A.desc[((int64_t)f * A.cap + ki) * 128 + k] = 0.f;
    return;
  }
  if (tid == 0) {                                   // start_x += sin_dir, start_y += cos_dir: float running sums over the rows
    const float win_offset = -(float)(win - 1) / 2;
    float sx = cx + win_offset * cos_dir + win_offset * sin_dir;
    float sy = cy - win_offset * sin_dir + win_offset * cos_dir;
    for (int i = 0; i < win; i++) { S.start_x[i] = sx; S.start_y[i] = sy; sx += sin_dir; sy += cos_dir; }
  }
  // INTER_AREA tables (win -> 21), identical for both axes
  const double inv_scale = (double)(SU_PATCH + 1) / win, scale = 1. / inv_scale;
  const int iscale = (int)rint(scale);
  const bool identity = win == SU_PATCH + 1;
  const bool fast = !identity && fabs(scale - iscale) < 2.220446049250313e-16;
  if (!identity && !fast && tid < 21) {
    const int dx = tid;
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = fmin(scale, (double)win - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = min(sx2, win - 1);
    sx1 = min(sx1, sx2);
    int n = 0;
    if (sx1 - fsx1 > 1e-3) { S.tsi[dx * SU_TAPMAX + n] = (short)(sx1 - 1); S.tal[dx * SU_TAPMAX + n] = (float)((sx1 - fsx1) / cell); n++; }
    for (int sx = sx1; sx < sx2; sx++) { S.tsi[dx * SU_TAPMAX + n] = (short)sx; S.tal[dx * SU_TAPMAX + n] = (float)(1.0 / cell); n++; }
    if (fsx2 - sx2 > 1e-3) { S.tsi[dx * SU_TAPMAX + n] = (short)sx2; S.tal[dx * SU_TAPMAX + n] = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell); n++; }
    S.tcnt[dx] = n;
  }
  __syncthreads();
  // ---- the rotated window: thread = one row, pixels in the operator's running-sum order, folded into the row's 21 sums
  const uint8_t* img = A.gray + (int64_t)f * A.gray_frame_bytes;
  const int ncols1 = A.w - 1, nrows1 = A.h - 1, gst = A.gstride;
  for (int i = tid; i < win; i += 256) {
    double pixel_x = (double)S.start_x[i], pixel_y = (double)S.start_y[i];
    int j = 0;
    auto sample = [&]() -> int {
      const int ix = (int)floor(pixel_x), iy = (int)floor(pixel_y);
      if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
        const float a = (float)(pixel_x - (double)ix), b = (float)(pixel_y - (double)iy);
        const uint8_t* p = img + (int64_t)iy * gst + ix;
        const float v = (float)p[0] * (1.f - a) * (1.f - b) + (float)p[1] * a * (1.f - b) + (float)p[gst] * (1.f - a) * b + (float)p[gst + 1] * a * b;
        return (int)(unsigned char)(int)rintf(v);
      }
      const int x = min(max((int)rint(pixel_x), 0), ncols1), y = min(max((int)rint(pixel_y), 0), nrows1);
      return img[(int64_t)y * gst + x];
    };
    int val = sample();
    if (identity || fast) {
      const int isx = identity ? 1 : iscale;
      for (int dx = 0; dx < 21; dx++) {
        int acc = 0;
        for (int k = 0; k < isx; k++) {
          const int sx = dx * isx + k;
          while (j < sx) { pixel_x += (double)cos_dir; pixel_y -= (double)sin_dir; j++; val = sample(); }
          acc += val;
        }
        s_rowbuf[i * 21 + dx] = __int_as_float(acc);
      }
    } else {
      for (int dx = 0; dx < 21; dx++) {
        float buf = 0.f;
        const int n = S.tcnt[dx];
        for (int k = 0; k < n; k++) {
          const int sx = S.tsi[dx * SU_TAPMAX + k];
          while (j < sx) { pixel_x += (double)cos_dir; pixel_y -= (double)sin_dir; j++; val = sample(); }
          buf = buf + (float)val * S.tal[dx * SU_TAPMAX + k];
        }
        s_rowbuf[i * 21 + dx] = buf;
      }
    }
  }
  __syncthreads();
  // ---- vertical INTER_AREA sums -> the 21 x 21 uint8 patch
  for (int e = tid; e < 21 * 21; e += 256) {
    const int dy = e / 21, dx = e - dy * 21;
    int out;
    if (identity || fast) {
      const int isy = identity ? 1 : iscale;
      int acc = 0;
      for (int k = 0; k < isy; k++) acc += __float_as_int(s_rowbuf[(dy * isy + k) * 21 + dx]);
      if (identity) out = acc;
      else if (isy == 2) out = (acc + 2) >> 2;
      else out = (int)rintf((float)acc * (1.f / (float)(isy * isy)));
    } else {
      float sumv = 0.f;
      const int n = S.tcnt[dy];
      for (int k = 0; k < n; k++) {
        const float term = S.tal[dy * SU_TAPMAX + k] * s_rowbuf[S.tsi[dy * SU_TAPMAX + k] * 21 + dx];
        sumv = k == 0 ? term : sumv + term;
      }
      out = (int)rintf(sumv);
    }
    S.patch[e] = (unsigned char)min(max(out, 0), 255);
  }
  __syncthreads();
  // ---- Gaussian-weighted gradients of the patch
  for (int e = tid; e < SU_PATCH * SU_PATCH; e += 256) {
    const int i = e / SU_PATCH, j = e - i * SU_PATCH;
    const float dw = A.T->DW[e];
    const int p00 = S.patch[i * 21 + j], p01 = S.patch[i * 21 + j + 1], p10 = S.patch[(i + 1) * 21 + j], p11 = S.patch[(i + 1) * 21 + j + 1];
    S.DX[e] = (float)(p01 - p00 + p11 - p10) * dw;
    S.DY[e] = (float)(p10 - p00 + p11 - p01) * dw;
  }
  __syncthreads();
  // ---- 4 x 4 cells x 8 sums, each in raster order over its 5 x 5 samples
  if (tid < 128) {
    const int cell = tid >> 3, comp = tid & 7, ci = cell >> 2, cj = cell & 3;
    float acc = 0.f;
    for (int y = ci * 5; y < ci * 5 + 5; y++)
      for (int x = cj * 5; x < cj * 5 + 5; x++) {
        const float tx = S.DX[y * SU_PATCH + x], ty = S.DY[y * SU_PATCH + x];
        if (comp < 4) {
          const bool up = ty >= 0;
          if (comp == 0 && up) acc += tx;
          if (comp == 1 && up) acc += fabsf(tx);
          if (comp == 2 && !up) acc += tx;
          if (comp == 3 && !up) acc += fabsf(tx);
        } else {
          const bool rt = tx >= 0;
          if (comp == 4 && rt) acc += ty;
          if (comp == 5 && rt) acc += fabsf(ty);
          if (comp == 6 && !rt) acc += ty;
          if (comp == 7 && !rt) acc += fabsf(ty);
        }
      }
    S.vec[tid] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    double square_mag = 0;
    for (int k = 0; k < 128; k++) square_mag += (double)(S.vec[k] * S.vec[k]);
    S.scale = (float)(1. / (sqrt(square_mag) + (double)FLT_EPSILON));
  }
  __syncthreads();
  if (tid < 128) A.desc[((int64_t)f * A.cap + ki) * 128 + tid] = S.vec[tid] * S.scale;
}

void gauss_kernel_f(int n, double sigma, std::vector<float>& k) {
  k.resize(n);
  const double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double x = i - (n - 1) * 0.5;
    k[i] = (float)std::exp(scale2X * x * x);
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

void resize_boxes(const int src[][5], SurfBox* dst, int n, int oldSize, int newSize) {
  const float ratio = (float)newSize / oldSize;
  for (int k = 0; k < n; k++) {
    dst[k].dx1 = (int)lrintf(ratio * src[k][0]); dst[k].dy1 = (int)lrintf(ratio * src[k][1]);
    dst[k].dx2 = (int)lrintf(ratio * src[k][2]); dst[k].dy2 = (int)lrintf(ratio * src[k][3]);
    dst[k].w = src[k][4] / ((float)(dst[k].dx2 - dst[k].dx1) * (dst[k].dy2 - dst[k].dy1));
  }
}

// layer tables of fastHessianDetector for a w x h frame; returns floats per frame of the det (= trace) plane set
int64_t surf_tables(int w, int h, SurfTabs& T) {
  const int dx_s[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
  const int dy_s[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
  const int dxy_s[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};
  int64_t off = 0;
  int index = 0, step = 1;
  for (int octave = 0; octave < SU_OCT; octave++) {
    for (int layer = 0; layer < SU_LAY + 2; layer++, index++) {
      SurfLayerTab& L = T.L[index];
      L.rows = h / step; L.cols = w / step; L.size = (9 + 6 * layer) << octave; L.step = step;
      L.built = !(L.size > h || L.size > w);
      resize_boxes(dx_s, L.Dx, 3, 9, L.size);
      resize_boxes(dy_s, L.Dy, 3, 9, L.size);
      resize_boxes(dxy_s, L.Dxy, 4, 9, L.size);
      L.off = off;
      off += (int64_t)std::max(L.rows, 1) * std::max(L.cols, 1);
    }
    step *= 2;
  }
  std::vector<float> G_ori, G_desc;
  gauss_kernel_f(13, 2.5f, G_ori);
  gauss_kernel_f(SU_PATCH, 3.3f, G_desc);
  int n = 0;
  for (int i = -6; i <= 6; i++)
    for (int j = -6; j <= 6; j++)
      if (i * i + j * j <= 36) { T.aptx[n] = i; T.apty[n] = j; T.aptw[n] = G_ori[i + 6] * G_ori[j + 6]; n++; }
  for (int i = 0; i < SU_PATCH; i++)
    for (int j = 0; j < SU_PATCH; j++) T.DW[i * SU_PATCH + j] = G_desc[i] * G_desc[j];
  return off;
}

template <class T>
int salloc(evh_ctx* c, T** p, size_t n) {
  EVH_HIP(c, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  c->bytes_allocated += n * sizeof(T);
  return EVH_SUCCESS;
}

}  // namespace

int evh_surf_allocate(evh_ctx* c, int max_surf_features) {
  if (c->surf_cap) return max_surf_features <= c->surf_cap ? EVH_SUCCESS
                                                            : evh_fail(c, EVH_ERR_CAPACITY, "evh_surf_enable: already enabled with a smaller capacity");
  if (max_surf_features < 64 || max_surf_features > 65536) return evh_fail(c, EVH_ERR_INVALID, "evh_surf_enable: capacity out of range (64..65536)");
  SurfTabs T{};
  const int64_t det_floats = surf_tables(c->max_w, c->max_h, T);
  const int sstride = (c->max_w + 1 + 15) & ~15;
  const int64_t sum_ints = (int64_t)sstride * (c->max_h + 1);
  const size_t per_frame = (size_t)(2 * det_floats + sum_ints) * 4;
  const int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)c->max_frames, ((size_t)4 << 30) / per_frame));
  const int cap = (max_surf_features + 63) & ~63;
  const size_t F = (size_t)c->max_frames;
  int rc;
#define S_(call) if ((rc = (call)) != EVH_SUCCESS) { evh_surf_free(c); return rc; }   /* a partial allocation is released */
  S_(salloc(c, &c->d_surf_tabs, sizeof(SurfTabs)));
  S_(salloc(c, &c->d_surf_sum, (size_t)group * sum_ints + 64));
  S_(salloc(c, &c->d_surf_det, (size_t)group * det_floats + 64));
  S_(salloc(c, &c->d_surf_trace, (size_t)group * det_floats + 64));
  S_(salloc(c, &c->d_surf_raw, F * cap * 8));
  S_(salloc(c, &c->d_surf_nraw, F));
  S_(salloc(c, &c->d_surf_srt, F * cap * 8));
  S_(salloc(c, &c->d_surf_kp, F * cap * 8));
  S_(salloc(c, &c->d_surf_xy, F * cap * 2));
  S_(salloc(c, &c->d_surf_desc, F * cap * 128));
  S_(salloc(c, &c->d_surf_count, F));
  S_(salloc(c, &c->d_surf_flags, F));
#undef S_
  EVH_HIP(c, hipMemsetAsync(c->d_surf_count, 0, F * sizeof(int), c->stream));
  EVH_HIP(c, hipMemsetAsync(c->d_surf_flags, 0, F * sizeof(int), c->stream));
  c->surf_cap = cap; c->surf_group = group; c->surf_sum_frame_ints = sum_ints; c->surf_det_frame_floats = det_floats;
  c->surf_tab_w = c->surf_tab_h = 0;
  return EVH_SUCCESS;
}

void evh_surf_free(evh_ctx* c) {
  void* ptrs[] = {c->d_surf_tabs, c->d_surf_sum, c->d_surf_det, c->d_surf_trace, c->d_surf_raw, c->d_surf_nraw, c->d_surf_srt,
                  c->d_surf_kp, c->d_surf_xy, c->d_surf_desc, c->d_surf_count, c->d_surf_flags};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  c->d_surf_tabs = nullptr; c->d_surf_sum = nullptr; c->d_surf_det = nullptr; c->d_surf_trace = nullptr; c->d_surf_raw = nullptr;
  c->d_surf_nraw = nullptr; c->d_surf_srt = nullptr; c->d_surf_kp = nullptr; c->d_surf_xy = nullptr; c->d_surf_desc = nullptr;
  c->d_surf_count = nullptr; c->d_surf_flags = nullptr; c->surf_cap = 0;
}

// SURF on the frames whose gray level 0 is resident in the context's ORB pyramid
int evh_launch_surf(evh_ctx* c, int nframes, int w, int h, float hessian_threshold) {
  if (!c->surf_cap) return evh_fail(c, EVH_ERR_INVALID, "SURF is not enabled on this context (evh_surf_enable)");
  if (w > c->max_w || h > c->max_h) return evh_fail(c, EVH_ERR_CAPACITY, "SURF: frame larger than the size given to evh_create");
  hipStream_t s = c->stream;
  if (c->surf_tab_w != w || c->surf_tab_h != h) {
    SurfTabs T{};
    const int64_t det_floats = surf_tables(w, h, T);
    if (det_floats > c->surf_det_frame_floats) return evh_fail(c, EVH_ERR_CAPACITY, "SURF: geometry exceeds the buffers of evh_surf_enable");
    EVH_HIP(c, hipStreamSynchronize(s));               // earlier launches still read the old tables
    EVH_HIP(c, hipMemcpy(c->d_surf_tabs, &T, sizeof(T), hipMemcpyHostToDevice));
    c->surf_tab_w = w; c->surf_tab_h = h;
  }
  const EvhLevel& L0 = c->g.lv[0];
  SurfArgs A{};
  A.T = reinterpret_cast<const SurfTabs*>(c->d_surf_tabs);
  A.gray = c->d_pyr + L0.off; A.gray_frame_bytes = c->g.pyr_frame_bytes; A.gstride = L0.stride; A.w = w; A.h = h;
  A.sum = c->d_surf_sum; A.sum_frame_ints = c->surf_sum_frame_ints; A.sstride = (w + 1 + 15) & ~15;
  A.det = c->d_surf_det; A.trace = c->d_surf_trace; A.det_frame_floats = c->surf_det_frame_floats;
  A.raw = c->d_surf_raw; A.nraw = c->d_surf_nraw; A.srt = c->d_surf_srt; A.kp = c->d_surf_kp; A.xy = c->d_surf_xy;
  A.desc = c->d_surf_desc; A.count = c->d_surf_count; A.flags = c->d_surf_flags; A.cap = c->surf_cap;
  EVH_HIP(c, hipMemsetAsync(c->d_surf_nraw, 0, sizeof(int) * (size_t)nframes, s));
  const size_t lds = sizeof(float) * 21 * SU_WINMAX, lds_small = sizeof(float) * 21 * SU_WINSMALL;
  EVH_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_surf_describe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int f0 = 0; f0 < nframes; f0 += c->surf_group) {
    const int ng = std::min(c->surf_group, nframes - f0);
    if (ng > 3000) return evh_fail(c, EVH_ERR_CAPACITY, "SURF: too many frames in one group");
    hipLaunchKernelGGL(k_surf_integral_rows, dim3(h + 1, ng), dim3(256), 0, s, A, f0);
    hipLaunchKernelGGL(k_surf_integral_cols, dim3((w + 256) / 256, ng), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_surf_det, dim3((w + 63) / 64, (h + 3) / 4, SU_NL * ng), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_surf_maxima, dim3((w + 63) / 64, (h + 3) / 4, SU_OCT * SU_LAY * ng), dim3(256), 0, s, A, f0, hessian_threshold);
    hipLaunchKernelGGL(k_surf_rank, dim3((c->surf_cap + 255) / 256, ng), dim3(256), 0, s, A, f0);
    hipLaunchKernelGGL(k_surf_compact, dim3(ng), dim3(1024), 0, s, A, f0);
    hipLaunchKernelGGL(k_surf_describe, dim3(c->surf_cap, ng), dim3(256), lds_small, s, A, f0, 0, SU_WINSMALL);
    hipLaunchKernelGGL(k_surf_describe, dim3(c->surf_cap, ng), dim3(256), lds, s, A, f0, SU_WINSMALL, SU_WINMAX);
    EVH_HIP(c, hipGetLastError());
  }
  c->surf_frames_resident = nframes;
  return EVH_SUCCESS;
}
