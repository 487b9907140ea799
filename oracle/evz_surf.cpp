/*
 * oracle/evz_surf.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), SURF half.
 *
 * Restates what the reference executes at
 *   evenvizion/processing/frame_processing.py:65-67   cv2.xfeatures2d.SURF_create(extended=1, hessianThreshold=400)
 *                                                     .detectAndCompute(frame, None)
 * i.e. opencv-contrib 3.4.2 xfeatures2d/src/surf.cpp with nOctaves 4, nOctaveLayers 3, extended (128 floats),
 * upright = false: integral image, box-filter Hessian (det = Dxx*Dyy - 0.81*Dxy^2) on 4 x 5 layers, strict 3x3x3 maxima
 * above the threshold, quadratic interpolation (Matx33f::solve = Cramer's rule in float), key points sorted by
 * KeypointGreater (response, size, octave, y, x), dominant orientation from Haar responses in a sliding 60-degree window,
 * a rotated (21 s) x (21 s) window resized to 21 x 21 with INTER_AREA, 4 x 4 x 8 sums of Gaussian-weighted gradients,
 * unit length.
 *
 * PARITY STATUS: restated from recall; pinned jointly, since round 4, by the reference's own video and recorded result (tests/test_capture_golden.py: all 120 matrices of dict_with_homography_matrix.json reproduced to the last digit) (opencv-contrib-python==3.4.2.17, requirements.txt:3, is absent
 * from /root/reference and from this image; the reference holds no vector at this boundary).  sin / cos of the dominant
 * direction go through the oracle's deterministic evo_sincos (double, rounded to float) instead of libm's sinf / cosf so
 * that the HIP build can reproduce every bit.
 *
 * Compile with -ffp-contract=off: float / double expressions below are one IEEE operation at a time.
 */
#include "evz_oracle.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

const int SURF_ORI_SEARCH_INC = 5, SURF_HAAR_SIZE0 = 9, SURF_HAAR_SIZE_INC = 6;
const float SURF_ORI_SIGMA = 2.5f, SURF_DESC_SIGMA = 3.3f;
const int N_OCTAVES = 4, N_LAYERS = 3, ORI_RADIUS = 6, ORI_WIN = 60, PATCH_SZ = 20;
const float HESSIAN_THRESHOLD = 400.f;

inline int round_f(float v) { return (int)lrintf(v); }
inline int round_d(double v) { return (int)lrint(v); }
inline int floor_d(double v) { int i = (int)v; return i - (i > v); }

struct SurfHF { int p0 = 0, p1 = 0, p2 = 0, p3 = 0; float w = 0; };

inline float calc_haar(const int* origin, const SurfHF* f, int n) {
  double d = 0;
  for (int k = 0; k < n; k++) d += (origin[f[k].p0] + origin[f[k].p3] - origin[f[k].p1] - origin[f[k].p2]) * f[k].w;
  return (float)d;
}

void resize_haar(const int src[][5], SurfHF* dst, int n, int oldSize, int newSize, int widthStep) {
  const float ratio = (float)newSize / oldSize;
  for (int k = 0; k < n; k++) {
    const int dx1 = round_f(ratio * src[k][0]), dy1 = round_f(ratio * src[k][1]);
    const int dx2 = round_f(ratio * src[k][2]), dy2 = round_f(ratio * src[k][3]);
    dst[k].p0 = dy1 * widthStep + dx1; dst[k].p1 = dy2 * widthStep + dx1;
    dst[k].p2 = dy1 * widthStep + dx2; dst[k].p3 = dy2 * widthStep + dx2;
    dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
  }
}

struct KP { float x, y, size, angle, response; int octave, class_id; };

struct Layer { int rows = 0, cols = 0, size = 0, step = 0; bool built = false; std::vector<float> det, trace; };

void gauss_kernel_f(int n, double sigma, std::vector<float>& k) {   /* getGaussianKernel(n, sigma, CV_32F), n > 7 */
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

/* interpolateKeypoint: quadratic fit in the 3x3x3 neighbourhood, Cramer's rule in float */
bool interpolate(const float N9[3][9], int dx, int dy, int ds, KP& kpt) {
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
  const bool ok = (x0 != 0 || x1 != 0 || x2 != 0) && std::fabs(x0) <= 1 && std::fabs(x1) <= 1 && std::fabs(x2) <= 1;
  if (ok) {
    kpt.x += x0 * dx;
    kpt.y += x1 * dy;
    kpt.size = (float)round_f(kpt.size + x2 * ds);
  }
  return ok;
}

bool kp_greater(const KP& a, const KP& b) {   /* KeypointGreater */
  if (a.response > b.response) return true;
  if (a.response < b.response) return false;
  if (a.size > b.size) return true;
  if (a.size < b.size) return false;
  if (a.octave > b.octave) return true;
  if (a.octave < b.octave) return false;
  if (a.y < b.y) return false;
  if (a.y > b.y) return true;
  return a.x < b.x;
}

void fast_hessian(const std::vector<int>& sum, int srows, int scols, std::vector<KP>& kps) {
  const int nTotal = (N_LAYERS + 2) * N_OCTAVES;
  std::vector<Layer> L(nTotal);
  int index = 0, step = 1;
  for (int octave = 0; octave < N_OCTAVES; octave++) {
    for (int layer = 0; layer < N_LAYERS + 2; layer++, index++) {
      L[index].rows = (srows - 1) / step; L[index].cols = (scols - 1) / step;
      L[index].size = (SURF_HAAR_SIZE0 + SURF_HAAR_SIZE_INC * layer) << octave;
      L[index].step = step;
    }
    step *= 2;
  }
  const int dx_s[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
  const int dy_s[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
  const int dxy_s[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};
  for (Layer& l : L) {                                   /* calcLayerDetAndTrace */
    l.det.assign((size_t)std::max(l.rows, 0) * std::max(l.cols, 0), 0.f);
    l.trace.assign(l.det.size(), 0.f);
    const int size = l.size, ss = l.step;
    if (size > srows - 1 || size > scols - 1) continue;
    SurfHF Dx[3], Dy[3], Dxy[4];
    resize_haar(dx_s, Dx, 3, 9, size, scols);
    resize_haar(dy_s, Dy, 3, 9, size, scols);
    resize_haar(dxy_s, Dxy, 4, 9, size, scols);
    const int samples_i = 1 + (srows - 1 - size) / ss, samples_j = 1 + (scols - 1 - size) / ss;
    const int margin = (size / 2) / ss;
    for (int i = 0; i < samples_i; i++) {
      const int* sp = &sum[(size_t)(i * ss) * scols];
      for (int j = 0; j < samples_j; j++) {
        const float dx = calc_haar(sp, Dx, 3), dy = calc_haar(sp, Dy, 3), dxy = calc_haar(sp, Dxy, 4);
        sp += ss;
        l.det[(size_t)(i + margin) * l.cols + j + margin] = dx * dy - 0.81f * dxy * dxy;
        l.trace[(size_t)(i + margin) * l.cols + j + margin] = dx + dy;
      }
    }
    l.built = true;
  }
  for (int octave = 0; octave < N_OCTAVES; octave++)      /* findMaximaInLayer over the middle layers */
    for (int ml = 1; ml <= N_LAYERS; ml++) {
      const int layer = octave * (N_LAYERS + 2) + ml;
      const Layer& l = L[layer];
      const int size = l.size, ss = l.step;
      const int layer_rows = (srows - 1) / ss, layer_cols = (scols - 1) / ss;
      const int margin = (L[layer + 1].size / 2) / ss + 1;
      const int st = l.cols;
      for (int i = margin; i < layer_rows - margin; i++)
        for (int j = margin; j < layer_cols - margin; j++) {
          const float val0 = l.det[(size_t)i * st + j];
          if (!(val0 > HESSIAN_THRESHOLD)) continue;
          const int sum_i = ss * (i - (size / 2) / ss), sum_j = ss * (j - (size / 2) / ss);
          float N9[3][9];
          for (int d = 0; d < 3; d++) {
            const float* p = &L[layer - 1 + d].det[(size_t)i * st + j];
            const float v[9] = {p[-st - 1], p[-st], p[-st + 1], p[-1], p[0], p[1], p[st - 1], p[st], p[st + 1]};
            memcpy(N9[d], v, sizeof(v));
          }
          bool mx = true;
          for (int d = 0; d < 3 && mx; d++)
            for (int q = 0; q < 9; q++) {
              if (d == 1 && q == 4) continue;
              if (!(val0 > N9[d][q])) { mx = false; break; }
            }
          if (!mx) continue;
          const float center_i = sum_i + (size - 1) * 0.5f, center_j = sum_j + (size - 1) * 0.5f;
          const float tr = l.trace[(size_t)i * st + j];
          KP kpt{center_j, center_i, (float)size, -1.f, val0, octave, (tr > 0) - (tr < 0)};
          const int ds = size - L[layer - 1].size;
          if (interpolate(N9, ss, ss, ds, kpt)) kps.push_back(kpt);
        }
    }
  std::stable_sort(kps.begin(), kps.end(), kp_greater);
}

/* cv2.resize(win, (21, 21), INTER_AREA) of a square 8-bit window -- evo_resize_area's arithmetic */
void area21(const std::vector<uint8_t>& win, int ws, uint8_t* patch) { evo_resize_area(win.data(), ws, ws, 1, patch, PATCH_SZ + 1, PATCH_SZ + 1); }

}  // namespace

/* integral(img, sum, CV_32S): (h+1) x (w+1) */
extern "C" void evo_integral(const uint8_t* gray, int w, int h, int32_t* sum) {
  const int sc = w + 1;
  for (int x = 0; x <= w; x++) sum[x] = 0;
  for (int y = 0; y < h; y++) {
    int s = 0;
    sum[(size_t)(y + 1) * sc] = 0;
    for (int x = 0; x < w; x++) {
      s += gray[(size_t)y * w + x];
      sum[(size_t)(y + 1) * sc + x + 1] = sum[(size_t)y * sc + x + 1] + s;
    }
  }
}

/* SURF_create(extended=1, hessianThreshold=400).detectAndCompute(gray, None): key points in the operator's own order
 * (KeypointGreater: response descending), descriptors float32[N,128].  Returns the count (first cap written). */
extern "C" int evo_surf_detect(const uint8_t* gray, int w, int h, float* xy, float* desc, float* size, float* angle,
                               float* response, int* octave, int* laplacian, int cap) {
  const int srows = h + 1, scols = w + 1;
  std::vector<int> sum((size_t)srows * scols);
  evo_integral(gray, w, h, sum.data());
  std::vector<KP> kps;
  fast_hessian(sum, srows, scols, kps);
  const int N = (int)kps.size();
  /* SURFInvoker set-up */
  std::vector<float> G_ori, G_desc;
  gauss_kernel_f(2 * ORI_RADIUS + 1, SURF_ORI_SIGMA, G_ori);
  gauss_kernel_f(PATCH_SZ, SURF_DESC_SIGMA, G_desc);
  std::vector<int> aptx, apty; std::vector<float> aptw;
  for (int i = -ORI_RADIUS; i <= ORI_RADIUS; i++)
    for (int j = -ORI_RADIUS; j <= ORI_RADIUS; j++)
      if (i * i + j * j <= ORI_RADIUS * ORI_RADIUS) {
        aptx.push_back(i); apty.push_back(j);
        aptw.push_back(G_ori[i + ORI_RADIUS] * G_ori[j + ORI_RADIUS]);
      }
  const int nOriSamples = (int)aptx.size();
  float DW[PATCH_SZ * PATCH_SZ];
  for (int i = 0; i < PATCH_SZ; i++)
    for (int j = 0; j < PATCH_SZ; j++) DW[i * PATCH_SZ + j] = G_desc[i] * G_desc[j];
  const int dx_s[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}};
  const int dy_s[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
  std::vector<float> descs((size_t)N * 128, 0.f);
  for (int k = 0; k < N; k++) {
    KP& kp = kps[k];
    const float sz = kp.size;
    const float cx = kp.x, cy = kp.y;
    const float s = sz * 1.2f / 9.0f;
    const int grad_wav_size = 2 * round_f(2 * s);
    if (srows < grad_wav_size || scols < grad_wav_size) { kp.size = -1; continue; }
    SurfHF dx_t[2], dy_t[2];
    resize_haar(dx_s, dx_t, 2, 4, grad_wav_size, scols);
    resize_haar(dy_s, dy_t, 2, 4, grad_wav_size, scols);
    std::vector<float> X, Y, ang;
    for (int kk = 0; kk < nOriSamples; kk++) {
      const int x = round_f(cx + aptx[kk] * s - (float)(grad_wav_size - 1) / 2);
      const int y = round_f(cy + apty[kk] * s - (float)(grad_wav_size - 1) / 2);
      if (y < 0 || y >= srows - grad_wav_size || x < 0 || x >= scols - grad_wav_size) continue;
      const int* ptr = &sum[(size_t)y * scols + x];
      const float vx = calc_haar(ptr, dx_t, 2), vy = calc_haar(ptr, dy_t, 2);
      X.push_back(vx * aptw[kk]); Y.push_back(vy * aptw[kk]);
    }
    const int nangle = (int)X.size();
    if (nangle == 0) { kp.size = -1; continue; }
    ang.resize(nangle);
    for (int j = 0; j < nangle; j++) ang[j] = evo_fast_atan2(Y[j], X[j]);       /* phase(X, Y, angle, true) */
    float bestx = 0, besty = 0, descriptor_mod = 0;
    for (int i = 0; i < 360; i += SURF_ORI_SEARCH_INC) {
      float sumx = 0, sumy = 0;
      for (int j = 0; j < nangle; j++) {
        const int d = std::abs(round_f(ang[j]) - i);
        if (d < ORI_WIN / 2 || d > 360 - ORI_WIN / 2) { sumx += X[j]; sumy += Y[j]; }
      }
      const float temp_mod = sumx * sumx + sumy * sumy;
      if (temp_mod > descriptor_mod) { descriptor_mod = temp_mod; bestx = sumx; besty = sumy; }
    }
    float descriptor_dir = evo_fast_atan2(-besty, bestx);
    kp.angle = descriptor_dir;
    /* the rotated window of (21 s)^2 pixels, bilinear, rounded to uint8 */
    const int win_size = (int)((PATCH_SZ + 1) * s);
    std::vector<uint8_t> win((size_t)win_size * win_size);
    descriptor_dir *= (float)(M_PI / 180);
    double sd, cd;
    evo_sincos((double)descriptor_dir, &sd, &cd);
    const float sin_dir = -(float)sd, cos_dir = (float)cd;
    const float win_offset = -(float)(win_size - 1) / 2;
    float start_x = cx + win_offset * cos_dir + win_offset * sin_dir;
    float start_y = cy - win_offset * sin_dir + win_offset * cos_dir;
    const int ncols1 = w - 1, nrows1 = h - 1;
    for (int i = 0; i < win_size; i++, start_x += sin_dir, start_y += cos_dir) {
      double pixel_x = start_x, pixel_y = start_y;
      for (int j = 0; j < win_size; j++, pixel_x += cos_dir, pixel_y -= sin_dir) {
        const int ix = floor_d(pixel_x), iy = floor_d(pixel_y);
        if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
          const float a = (float)(pixel_x - ix), b = (float)(pixel_y - iy);
          const uint8_t* p = gray + (size_t)iy * w + ix;
          win[(size_t)i * win_size + j] =
              (uint8_t)round_f(p[0] * (1.f - a) * (1.f - b) + p[1] * a * (1.f - b) + p[w] * (1.f - a) * b + p[w + 1] * a * b);
        } else {
          const int x = std::min(std::max(round_d(pixel_x), 0), ncols1);
          const int y = std::min(std::max(round_d(pixel_y), 0), nrows1);
          win[(size_t)i * win_size + j] = gray[(size_t)y * w + x];
        }
      }
    }
    uint8_t PATCH[PATCH_SZ + 1][PATCH_SZ + 1];
    area21(win, win_size, &PATCH[0][0]);
    float DX[PATCH_SZ][PATCH_SZ], DY[PATCH_SZ][PATCH_SZ];
    for (int i = 0; i < PATCH_SZ; i++)
      for (int j = 0; j < PATCH_SZ; j++) {
        const float dw = DW[i * PATCH_SZ + j];
        DX[i][j] = (PATCH[i][j + 1] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i + 1][j]) * dw;
        DY[i][j] = (PATCH[i + 1][j] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i][j + 1]) * dw;
      }
    float* vec = &descs[(size_t)k * 128];
    double square_mag = 0;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        for (int y = i * 5; y < i * 5 + 5; y++)
          for (int x = j * 5; x < j * 5 + 5; x++) {
            const float tx = DX[y][x], ty = DY[y][x];
            if (ty >= 0) { vec[0] += tx; vec[1] += (float)fabs(tx); } else { vec[2] += tx; vec[3] += (float)fabs(tx); }
            if (tx >= 0) { vec[4] += ty; vec[5] += (float)fabs(ty); } else { vec[6] += ty; vec[7] += (float)fabs(ty); }
          }
        for (int kk = 0; kk < 8; kk++) square_mag += vec[kk] * vec[kk];
        vec += 8;
      }
    vec = &descs[(size_t)k * 128];
    const float scale = (float)(1. / (std::sqrt(square_mag) + FLT_EPSILON));
    for (int kk = 0; kk < 128; kk++) vec[kk] *= scale;
  }
  int n = 0;
  for (int k = 0; k < N; k++) {
    if (!(kps[k].size > 0)) continue;
    if (n < cap) {
      if (xy) { xy[2 * n] = kps[k].x; xy[2 * n + 1] = kps[k].y; }
      if (desc) memcpy(desc + (size_t)n * 128, &descs[(size_t)k * 128], 128 * sizeof(float));
      if (size) size[n] = kps[k].size;
      if (angle) angle[n] = kps[k].angle;
      if (response) response[n] = kps[k].response;
      if (octave) octave[n] = kps[k].octave;
      if (laplacian) laplacian[n] = kps[k].class_id;
    }
    n++;
  }
  return n;
}
