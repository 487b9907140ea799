// evh_image.hip -- K0: area-average downscale, the MI355X counterpart of imutils.resize(frame, width=) ->
// cv2.resize(INTER_AREA) at evenvizion/processing/video_processing.py:62,73.  Shrinking (the reference's
// "resize_width to speed up" use) = area sums; enlarging = the operator's bilinear emulation (k_resize_linear_area);
// equal sizes are a plain copy.  Weights are float32 tables built on the host
// exactly as the operator builds them; each output value is accumulated in float32 in table order
// (inner sum over source columns, outer sum over source rows), then rounded half-to-even and saturated.
#include "evh_internal.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct AreaTabDev {
  const int* xs; const int* xcnt; const int* xsi; const float* xal;   // per dst column: first entry, count; entries
  const int* ys; const int* ycnt; const int* ysi; const float* yal;
};

__global__ void k_resize_area(const uint8_t* __restrict__ src, int cn, int64_t src_stride, int64_t src_img_stride,
                              uint8_t* __restrict__ dst, int dw, int dh, int64_t dst_stride, int64_t dst_img_stride,
                              AreaTabDev T) {
  const int img = blockIdx.z;
  const int dy = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // dx*cn + c
  if (e >= dw * cn) return;
  const int dx = e / cn, c = e - dx * cn;
  const uint8_t* S = src + (int64_t)img * src_img_stride;
  const int x0 = T.xs[dx], xn = T.xcnt[dx], y0 = T.ys[dy], yn = T.ycnt[dy];
  float sum = 0.f;
  for (int j = 0; j < yn; j++) {
    const uint8_t* row = S + (int64_t)T.ysi[y0 + j] * src_stride + c;
    float buf = 0.f;
    for (int k = 0; k < xn; k++) buf = buf + (float)row[(int64_t)T.xsi[x0 + k] * cn] * T.xal[x0 + k];
    float term = T.yal[y0 + j] * buf;
    sum = j == 0 ? term : sum + term;
  }
  int v = (int)rintf(sum);
  dst[(int64_t)img * dst_img_stride + (int64_t)dy * dst_stride + e] = (uint8_t)min(max(v, 0), 255);
}

__global__ void k_resize_area_int(const uint8_t* __restrict__ src, int cn, int64_t src_stride, int64_t src_img_stride,
                                  uint8_t* __restrict__ dst, int dw, int dh, int64_t dst_stride, int64_t dst_img_stride,
                                  int isx, int isy) {
  const int img = blockIdx.z, dy = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= dw * cn) return;
  const int dx = e / cn, c = e - dx * cn;
  const uint8_t* S = src + (int64_t)img * src_img_stride;
  int sum = 0;
  for (int j = 0; j < isy; j++)
    for (int i = 0; i < isx; i++) sum += S[(int64_t)(dy * isy + j) * src_stride + (int64_t)(dx * isx + i) * cn + c];
  int v;
  if (isx == 2 && isy == 2) v = (sum + 2) >> 2;
  else {
    const float scale = 1.f / (float)(isx * isy);
    v = (int)rintf((float)sum * scale);
  }
  dst[(int64_t)img * dst_img_stride + (int64_t)dy * dst_stride + e] = (uint8_t)min(max(v, 0), 255);
}

// N2: fused ingest.  The full-size frame is read ONCE: per output pixel the INTER_AREA sums of its channels
// (same float32 tables and accumulation order as k_resize_area, rounded to uint8 per channel exactly as the resized
// image would hold them), then cvtColor's gray weights on those bytes, written straight into pyramid level 0.  The
// resized BGR image of video_processing.py:62,73 never exists in memory.
__device__ __forceinline__ uint8_t gray_of(int b, int g, int r) { return (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14); }

__global__ __launch_bounds__(256) void k_ingest_area(const uint8_t* __restrict__ src, int cn, int64_t src_stride,
                                                     int64_t src_img_stride, uint8_t* __restrict__ pyr, int64_t pyr_frame_bytes,
                                                     int dw, int dh, int dst_stride, AreaTabDev T) {
  const int img = blockIdx.z, dy = blockIdx.y;
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  if (dx >= dw) return;
  const uint8_t* S = src + (int64_t)img * src_img_stride;
  const int x0 = T.xs[dx], xn = T.xcnt[dx], y0 = T.ys[dy], yn = T.ycnt[dy];
  float sum[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < yn; j++) {
    const uint8_t* row = S + (int64_t)T.ysi[y0 + j] * src_stride;
    float buf[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < xn; k++) {
      const uint8_t* px = row + (int64_t)T.xsi[x0 + k] * cn;
      const float a = T.xal[x0 + k];
#pragma unroll
      for (int c = 0; c < 3; c++) if (c < cn) buf[c] = buf[c] + (float)px[c] * a;
    }
    const float beta = T.yal[y0 + j];
#pragma unroll
    for (int c = 0; c < 3; c++) { const float term = beta * buf[c]; sum[c] = j == 0 ? term : sum[c] + term; }
  }
  int v[3];
#pragma unroll
  for (int c = 0; c < 3; c++) v[c] = min(max((int)rintf(sum[c]), 0), 255);
  pyr[(int64_t)img * pyr_frame_bytes + (int64_t)dy * dst_stride + dx] = cn == 3 ? gray_of(v[0], v[1], v[2]) : (uint8_t)v[0];
}

__global__ __launch_bounds__(256) void k_ingest_area_int(const uint8_t* __restrict__ src, int cn, int64_t src_stride,
                                                         int64_t src_img_stride, uint8_t* __restrict__ pyr,
                                                         int64_t pyr_frame_bytes, int dw, int dh, int dst_stride, int isx,
                                                         int isy) {
  const int img = blockIdx.z, dy = blockIdx.y;
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  if (dx >= dw) return;
  const uint8_t* S = src + (int64_t)img * src_img_stride;
  int sum[3] = {0, 0, 0};
  for (int j = 0; j < isy; j++) {
    const uint8_t* row = S + (int64_t)(dy * isy + j) * src_stride + (int64_t)dx * isx * cn;
    for (int i = 0; i < isx; i++)
#pragma unroll
      for (int c = 0; c < 3; c++) if (c < cn) sum[c] += row[i * cn + c];
  }
  int v[3];
  const float scale = 1.f / (float)(isx * isy);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int r = (isx == 2 && isy == 2) ? (sum[c] + 2) >> 2 : (int)rintf((float)sum[c] * scale);
    v[c] = min(max(r, 0), 255);
  }
  pyr[(int64_t)img * pyr_frame_bytes + (int64_t)dy * dst_stride + dx] = cn == 3 ? gray_of(v[0], v[1], v[2]) : (uint8_t)v[0];
}

// INTER_AREA when ENLARGING (imutils.resize(frame, width=) with width > frame width, video_processing.py:62,73): the
// operator emulates it with its 8-bit bilinear machinery -- 11-bit coefficients from area-mode tables
//   s = floor(d*scale), f = (float)((d+1) - (s+1)*inv_scale), f = f <= 0 ? 0 : f - floor(f),
// horizontal pass in int (x2048), vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2 >> 2.  One thread per
// output pixel; the coefficients are recomputed per thread in the operator's own f64/f32 arithmetic.
struct LinAreaCoef { int s; int a0, a1; };
__device__ __forceinline__ LinAreaCoef lin_area_coef(int d, int ssize, double scale, double inv) {
  int s = (int)floor((double)d * scale);
  float f = (float)((double)(d + 1) - (double)(s + 1) * inv);
  f = f <= 0.f ? 0.f : f - floorf(f);
  if (s < 0) { f = 0.f; s = 0; }
  bool edge = false;
  if (s + 1 >= ssize) { edge = true; if (s >= ssize - 1) { f = 0.f; s = ssize - 1; } }
  LinAreaCoef c;
  c.s = s;
  c.a0 = min(max((int)rintf((1.f - f) * 2048.f), -32768), 32767);
  c.a1 = min(max((int)rintf(f * 2048.f), -32768), 32767);
  if (edge) { c.a0 = 2048; c.a1 = 0; }           // D[dx] = S[sx] * ONE beyond xmax
  return c;
}
__device__ __forceinline__ int lin_area_px(const uint8_t* S, int64_t stride, int cn, int c, int sw, int sh,
                                           const LinAreaCoef& X, const LinAreaCoef& Y, int b0, int b1) {
  const int y0 = min(Y.s, sh - 1), y1 = min(Y.s + 1, sh - 1);
  const int x1 = min(X.s + 1, sw - 1);
  const uint8_t* r0 = S + (int64_t)y0 * stride;
  const uint8_t* r1 = S + (int64_t)y1 * stride;
  const int h0 = r0[X.s * cn + c] * X.a0 + r0[x1 * cn + c] * X.a1;
  const int h1 = r1[X.s * cn + c] * X.a0 + r1[x1 * cn + c] * X.a1;
  return ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xFF;
}
// vertical coefficients: no edge substitution (the operator clips the ROWS instead)
__device__ __forceinline__ void lin_area_beta(int dy, double scale, double inv, int& sy, int& b0, int& b1) {
  sy = (int)floor((double)dy * scale);
  float f = (float)((double)(dy + 1) - (double)(sy + 1) * inv);
  f = f <= 0.f ? 0.f : f - floorf(f);
  b0 = min(max((int)rintf((1.f - f) * 2048.f), -32768), 32767);
  b1 = min(max((int)rintf(f * 2048.f), -32768), 32767);
}
template <bool INGEST>
__global__ __launch_bounds__(256) void k_resize_linear_area(const uint8_t* __restrict__ src, int cn, int sw, int sh,
                                                            int64_t src_stride, int64_t src_img_stride,
                                                            uint8_t* __restrict__ dst, int dw, int dh, int64_t dst_stride,
                                                            int64_t dst_img_stride, double scale_x, double inv_x,
                                                            double scale_y, double inv_y) {
  const int img = blockIdx.z, dy = blockIdx.y;
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  if (dx >= dw) return;
  const uint8_t* S = src + (int64_t)img * src_img_stride;
  const LinAreaCoef X = lin_area_coef(dx, sw, scale_x, inv_x);
  LinAreaCoef Y; int b0, b1;
  lin_area_beta(dy, scale_y, inv_y, Y.s, b0, b1);
  Y.s = min(max(Y.s, 0), sh - 1);
  int v[3] = {0, 0, 0};
#pragma unroll
  for (int c = 0; c < 3; c++) if (c < cn) v[c] = lin_area_px(S, src_stride, cn, c, sw, sh, X, Y, b0, b1);
  uint8_t* D = dst + (int64_t)img * dst_img_stride + (int64_t)dy * dst_stride;
  if (INGEST) D[dx] = cn == 3 ? gray_of(v[0], v[1], v[2]) : (uint8_t)v[0];
  else
#pragma unroll
    for (int c = 0; c < 3; c++) if (c < cn) D[dx * cn + c] = (uint8_t)v[c];
}

// N3: fixed-plane coordinate field of processing_visualization.py:407-408 -- every pixel (x, y) of
// the resized frame mapped through the frame's superposed H -- and its maximum coordinate (the value
// heatmap_video_processing returns and evenvizion_component.py writes to metrics_file.txt).
__global__ __launch_bounds__(256) void k_fixed_plane(const double* __restrict__ Hs, int w, int h,
                                                     double* __restrict__ field, unsigned long long* __restrict__ out_max) {
  const int f = blockIdx.y;
  const double* H = Hs + 9 * f;
  const double h0 = H[0], h1 = H[1], h2 = H[2], h3 = H[3], h4 = H[4], h5 = H[5], h6 = H[6], h7 = H[7], h8 = H[8];
  double m = -INFINITY;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
    const int y = i / w, x = i - y * w;
    const double dx = (double)x, dy = (double)y;
    const double d = (h6 * dx + h7 * dy) + h8;
    const double u = ((h0 * dx + h1 * dy) + h2) / d;
    const double v = ((h3 * dx + h4 * dy) + h5) / d;
    if (field) { field[((int64_t)f * w * h + i) * 2] = u; field[((int64_t)f * w * h + i) * 2 + 1] = v; }
    m = fmax(m, fmax(u, v));
  }
  for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0 && m > -INFINITY) {
    // order-preserving key of a double so that an integer atomicMax gives the floating-point maximum
    unsigned long long b = (unsigned long long)__double_as_longlong(m);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    atomicMax(&out_max[f], b);
  }
}

// N1: utils.superposition_dict (utils.py:184-211) as one sequential scan: out[0] = H[0] (matrix_H_first),
// out[i] = np.dot(H[i], out[i-1]) / [2][2].  np.dot(3x3, 3x3) = the forward FMA chain pinned by the reference-captured
// fixtures (tests/golden/glue_goldens.json "sup_false"), the same order k_ransac_final_stream uses.
__global__ __launch_bounds__(64) void k_superposition_scan(const double* __restrict__ H, int n, double* __restrict__ out) {
  __shared__ double S[9];
  const int lane = threadIdx.x;
  if (lane < 9) { S[lane] = H[lane]; out[lane] = H[lane]; }
  __syncthreads();
  for (int i = 1; i < n; i++) {
    double P = 0;
    if (lane < 9) {
      const double* Hi = H + 9 * (int64_t)i;
      const int r = lane / 3, c = lane - 3 * r;
      P = fma(Hi[3 * r + 2], S[6 + c], fma(Hi[3 * r + 1], S[3 + c], Hi[3 * r] * S[c]));
    }
    const double P8 = __shfl(P, 8);
    __syncthreads();
    if (lane < 9) { const double v = P / P8; S[lane] = v; out[9 * (int64_t)i + lane] = v; }
    __syncthreads();
  }
}

// N1: fixed_coordinate_system.from_original_to_fix / from_fix_to_original (fixed_coordinate_system.py:19-69, 72-122)
// batched: point i -> np.around(np.dot(M[idx[i]], (kx*x, ky*y, 1))[:2] / [2], decimals).  M is H itself or (for the
// inverse direction) the caller's inverted matrix.  np.dot(3x3, vec) = fma(h0, x, h1*y) + h2 (fixture "hv"); np.around
// = rint(v * 10^d) / 10^d (half to even); decimals < 0: no rounding.
__global__ __launch_bounds__(256) void k_transform_points(const double* __restrict__ M, const int* __restrict__ idx,
                                                          const double* __restrict__ pts, int n, double kx, double ky,
                                                          int decimals, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* H = M + 9 * (int64_t)idx[i];
  const double x = kx * pts[2 * i], y = ky * pts[2 * i + 1];
  const double tx = fma(H[0], x, H[1] * y) + H[2];
  const double ty = fma(H[3], x, H[4] * y) + H[5];
  const double tw = fma(H[6], x, H[7] * y) + H[8];
  double u = tx / tw, v = ty / tw;
  if (decimals >= 0) {
    double sc = 1.0;
    for (int d = 0; d < decimals; d++) sc *= 10.0;
    u = __builtin_rint(u * sc) / sc;
    v = __builtin_rint(v * sc) / sc;
  }
  out[2 * i] = u; out[2 * i + 1] = v;
}

struct HostTab { std::vector<int> start, cnt, si; std::vector<float> al; };

void build_area_tab(int ssize, int dsize, double scale, HostTab& t) {
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    double cell = std::min(scale, ssize - fsx1);
    int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    t.start.push_back((int)t.si.size());
    if (sx1 - fsx1 > 1e-3) { t.si.push_back(sx1 - 1); t.al.push_back((float)((sx1 - fsx1) / cell)); }
    for (int sx = sx1; sx < sx2; sx++) { t.si.push_back(sx); t.al.push_back((float)(1.0 / cell)); }
    if (fsx2 - sx2 > 1e-3) { t.si.push_back(sx2); t.al.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)); }
    t.cnt.push_back((int)t.si.size() - t.start.back());
  }
}

}  // namespace

int evh_launch_resize_area(evh_ctx* c, const uint8_t* d_src, int nimg, int sw, int sh, int cn, int64_t src_stride,
                           int64_t src_img_stride, uint8_t* d_dst, int dw, int dh, int64_t dst_stride,
                           int64_t dst_img_stride) {
  if (dw == sw && dh == sh) {
    for (int i = 0; i < nimg; i++)
      EVH_HIP(c, hipMemcpy2DAsync(d_dst + i * dst_img_stride, dst_stride, d_src + i * src_img_stride, src_stride,
                                  (size_t)sw * cn, sh, hipMemcpyDeviceToDevice, c->stream));
    return EVH_SUCCESS;
  }
  const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
  const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
  if (scale_x < 1 || scale_y < 1) {          // enlarging: the operator's bilinear emulation of INTER_AREA
    hipLaunchKernelGGL(k_resize_linear_area<false>, dim3((dw + 255) / 256, dh, nimg), dim3(256), 0, c->stream, d_src, cn, sw,
                       sh, src_stride, src_img_stride, d_dst, dw, dh, dst_stride, dst_img_stride, scale_x, inv_x, scale_y,
                       inv_y);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  dim3 grid((dw * cn + 255) / 256, dh, nimg);
  const int isx = (int)std::lrint(scale_x), isy = (int)std::lrint(scale_y);
  if (std::fabs(scale_x - isx) < 2.220446049250313e-16 && std::fabs(scale_y - isy) < 2.220446049250313e-16) {
    hipLaunchKernelGGL(k_resize_area_int, grid, dim3(256), 0, c->stream, d_src, cn, src_stride, src_img_stride, d_dst, dw,
                       dh, dst_stride, dst_img_stride, isx, isy);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  HostTab xt, yt;
  build_area_tab(sw, dw, scale_x, xt);
  build_area_tab(sh, dh, scale_y, yt);
  // one device allocation for all eight tables, released after the launch completes
  const size_t nx = xt.si.size(), ny = yt.si.size();
  std::vector<int> blob;
  blob.insert(blob.end(), xt.start.begin(), xt.start.end());
  blob.insert(blob.end(), xt.cnt.begin(), xt.cnt.end());
  blob.insert(blob.end(), xt.si.begin(), xt.si.end());
  for (float f : xt.al) { int v; std::memcpy(&v, &f, 4); blob.push_back(v); }
  blob.insert(blob.end(), yt.start.begin(), yt.start.end());
  blob.insert(blob.end(), yt.cnt.begin(), yt.cnt.end());
  blob.insert(blob.end(), yt.si.begin(), yt.si.end());
  for (float f : yt.al) { int v; std::memcpy(&v, &f, 4); blob.push_back(v); }
  int* d_blob = nullptr;
  EVH_HIP(c, hipMalloc(&d_blob, blob.size() * sizeof(int)));
  hipError_t e = hipMemcpyAsync(d_blob, blob.data(), blob.size() * sizeof(int), hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) { (void)hipFree(d_blob); return evh_fail(c, EVH_ERR_HIP, "resize table upload failed"); }
  AreaTabDev T;
  int* p = d_blob;
  T.xs = p; p += dw; T.xcnt = p; p += dw; T.xsi = p; p += nx; T.xal = reinterpret_cast<float*>(p); p += nx;
  T.ys = p; p += dh; T.ycnt = p; p += dh; T.ysi = p; p += ny; T.yal = reinterpret_cast<float*>(p);
  hipLaunchKernelGGL(k_resize_area, grid, dim3(256), 0, c->stream, d_src, cn, src_stride, src_img_stride, d_dst, dw, dh,
                     dst_stride, dst_img_stride, T);
  e = hipGetLastError();
  (void)hipStreamSynchronize(c->stream);  // the pageable upload and the table lifetime both end here
  (void)hipFree(d_blob);
  if (e != hipSuccess) return evh_fail(c, EVH_ERR_HIP, std::string("k_resize_area: ") + hipGetErrorString(e));
  return EVH_SUCCESS;
}

// INTER_AREA tables of one (source, destination) geometry, kept on the device by the context (a stream resizes
// every chunk with the same geometry)
static int area_tables(evh_ctx* c, int sw, int sh, int dw, int dh, double scale_x, double scale_y, AreaTabDev& T) {
  const int64_t key = ((int64_t)sw << 48) ^ ((int64_t)sh << 32) ^ ((int64_t)dw << 16) ^ (int64_t)dh;
  if (c->area_key != key || !c->d_area_tab) {
    HostTab xt, yt;
    build_area_tab(sw, dw, scale_x, xt);
    build_area_tab(sh, dh, scale_y, yt);
    std::vector<int> blob;
    blob.insert(blob.end(), xt.start.begin(), xt.start.end());
    blob.insert(blob.end(), xt.cnt.begin(), xt.cnt.end());
    blob.insert(blob.end(), xt.si.begin(), xt.si.end());
    for (float f : xt.al) { int v; std::memcpy(&v, &f, 4); blob.push_back(v); }
    blob.insert(blob.end(), yt.start.begin(), yt.start.end());
    blob.insert(blob.end(), yt.cnt.begin(), yt.cnt.end());
    blob.insert(blob.end(), yt.si.begin(), yt.si.end());
    for (float f : yt.al) { int v; std::memcpy(&v, &f, 4); blob.push_back(v); }
    EVH_HIP(c, hipStreamSynchronize(c->stream));            // the previous tables may still be in use
    if (c->d_area_tab) { (void)hipFree(c->d_area_tab); c->d_area_tab = nullptr; }
    EVH_HIP(c, hipMalloc(&c->d_area_tab, blob.size() * sizeof(int)));
    EVH_HIP(c, hipMemcpy(c->d_area_tab, blob.data(), blob.size() * sizeof(int), hipMemcpyHostToDevice));
    c->area_key = key; c->area_nx = (int)xt.si.size(); c->area_ny = (int)yt.si.size();
  }
  int* p = c->d_area_tab;
  const size_t nx = (size_t)c->area_nx, ny = (size_t)c->area_ny;
  T.xs = p; p += dw; T.xcnt = p; p += dw; T.xsi = p; p += nx; T.xal = reinterpret_cast<float*>(p); p += nx;
  T.ys = p; p += dh; T.ycnt = p; p += dh; T.ysi = p; p += ny; T.yal = reinterpret_cast<float*>(p);
  return EVH_SUCCESS;
}

// level 0 of every frame straight from the source frames (k_ingest_area* when shrinking, k_resize_linear_area when enlarging)
int evh_launch_ingest_level0(evh_ctx* c, const uint8_t* d_src, int nimg, int sw, int sh, int cn, int64_t src_stride,
                             int64_t src_img_stride, int dw, int dh) {
  const EvhLevel& L = c->g.lv[0];
  const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;      // the operator's own arithmetic
  const double sx = 1. / inv_x, sy = 1. / inv_y;
  (void)scale_x; (void)scale_y;
  dim3 grid((dw + 255) / 256, dh, nimg);
  c->level1_fused = false;
  if (sx < 1 || sy < 1) {                    // enlarging: bilinear emulation, gray weights on the rounded channels
    hipLaunchKernelGGL(k_resize_linear_area<true>, grid, dim3(256), 0, c->stream, d_src, cn, sw, sh, src_stride,
                       src_img_stride, c->d_pyr + L.off, dw, dh, (int64_t)L.stride, c->g.pyr_frame_bytes, sx, inv_x, sy, inv_y);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  const int isx = (int)std::lrint(sx), isy = (int)std::lrint(sy);
  if (std::fabs(sx - isx) < 2.220446049250313e-16 && std::fabs(sy - isy) < 2.220446049250313e-16) {
    hipLaunchKernelGGL(k_ingest_area_int, grid, dim3(256), 0, c->stream, d_src, cn, src_stride, src_img_stride, c->d_pyr + L.off,
                       c->g.pyr_frame_bytes, dw, dh, L.stride, isx, isy);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  AreaTabDev T;
  int rc = area_tables(c, sw, sh, dw, dh, sx, sy, T);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ingest_area, grid, dim3(256), 0, c->stream, d_src, cn, src_stride, src_img_stride, c->d_pyr + L.off,
                     c->g.pyr_frame_bytes, dw, dh, L.stride, T);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_superposition_scan(evh_ctx* c, const double* d_H, int n, double* d_out) {
  hipLaunchKernelGGL(k_superposition_scan, dim3(1), dim3(64), 0, c->stream, d_H, n, d_out);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
int evh_launch_transform_points(evh_ctx* c, const double* d_M, const int* d_idx, const double* d_pts, int n, double kx,
                                double ky, int decimals, double* d_out) {
  hipLaunchKernelGGL(k_transform_points, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_M, d_idx, d_pts, n, kx, ky,
                     decimals, d_out);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_fixed_plane(evh_ctx* c, const double* d_H, int n, int w, int h, double* d_field, unsigned long long* d_max) {
  EVH_HIP(c, hipMemsetAsync(d_max, 0, sizeof(unsigned long long) * (size_t)n, c->stream));
  const int blocks = std::min((w * h + 255) / 256, 1024);
  hipLaunchKernelGGL(k_fixed_plane, dim3(blocks, n), dim3(256), 0, c->stream, d_H, w, h, d_field, d_max);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
