/*
 * oracle/evz_sift.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), SIFT half.
 *
 * Restates what the reference executes at
 *   evenvizion/processing/frame_processing.py:62-64   cv2.xfeatures2d.SIFT_create().detectAndCompute(frame, None)
 *   evenvizion/processing/matching.py:102-108         BruteForce knnMatch on the float32[N,128] descriptors
 * i.e. opencv-contrib 3.4.2 xfeatures2d/src/sift.cpp with its defaults (nfeatures 0, nOctaveLayers 3,
 * contrastThreshold 0.04, edgeThreshold 10, sigma 1.6; float scale space, SIFT_FIXPT_SCALE 1) and the imgproc /
 * core routines it calls (resize INTER_LINEAR x2 and INTER_NEAREST /2 on float, GaussianBlur on float = separable
 * filter with the plain row form and the symmetric column form, hal::exp32f / fastAtan2 / magnitude32f,
 * Matx33f::solve by Cramer's rule, KeyPointsFilter::removeDuplicatedSorted).
 *
 * PARITY STATUS: restated from recall; pinned jointly with the other operators, since round 4, by the reference's own video and
 * recorded result (tests/test_capture_golden.py: all 120 matrices of dict_with_homography_matrix.json reproduced to the last
 * digit).  opencv-contrib-python==3.4.2.17 (requirements.txt:3) is a third-party wheel absent from /root/reference and from
 * this image; the reference holds no vector at this boundary alone, so the wheel's build decisions (SSE2 / AVX2 / FMA3
 * dispatch) were settled by that comparison wherever it could see them:
 *   - the float Gaussian filter fuses its multiply-adds in the vector bodies only (gaussian_blur below) -- decisive: 0, 71 or
 *     120 of the 120 pairs come out exact depending on this alone;
 *   - fusing fastAtan2's / exp32f's polynomials or magnitude's x*x + y*y changes a handful of orientations in their last bit
 *     and no descriptor byte on the reference's video, so the golden cannot tell: this file keeps the plain, non-fused forms,
 *     and for exp32f the float-polynomial form of its vector body;
 *   - powf / cosf / sinf are replaced by deterministic double evaluations rounded to float (det_exp2, evo_sincos) so that the
 *     HIP build can reproduce every bit; the golden agrees with them on every pair.
 *
 * Compile with -ffp-contract=off: float / double expressions below are one IEEE operation at a time.
 */
#include "evz_oracle.h"
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

const int SIFT_DESCR_WIDTH = 4, SIFT_DESCR_HIST_BINS = 8, SIFT_IMG_BORDER = 5, SIFT_MAX_INTERP_STEPS = 5,
          SIFT_ORI_HIST_BINS = 36, N_OCTAVE_LAYERS = 3;
const float SIFT_INIT_SIGMA = 0.5f, SIFT_ORI_SIG_FCTR = 1.5f, SIFT_ORI_RADIUS = 3 * SIFT_ORI_SIG_FCTR,
            SIFT_ORI_PEAK_RATIO = 0.8f, SIFT_DESCR_SCL_FCTR = 3.f, SIFT_DESCR_MAG_THR = 0.2f, SIFT_INT_DESCR_FCTR = 512.f;
const double CONTRAST_THRESHOLD = 0.04, EDGE_THRESHOLD = 10, SIGMA = 1.6;

inline int round_f(float v) { return (int)lrintf(v); }
inline int round_d(double v) { return (int)lrint(v); }
inline int floor_f(float v) { int i = (int)v; return i - (i > v); }
inline int floor_d(double v) { int i = (int)v; return i - (i > v); }
inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
  return p;
}

struct Img {
  int w = 0, h = 0;
  std::vector<float> d;
  void create(int w_, int h_) { w = w_; h = h_; d.assign((size_t)w * h, 0.f); }
  float at(int r, int c) const { return d[(size_t)r * w + c]; }
  float& at(int r, int c) { return d[(size_t)r * w + c]; }
};

/* resize(src, dst, Size(2w, 2h), INTER_LINEAR) on float: fx = (float)((dx+0.5)*0.5 - 0.5), left tap floor(fx) */
void lin_tab2(int ssize, int dsize, std::vector<int>& ofs, std::vector<float>& a0, std::vector<float>& a1, int& vmax,
              bool clamp_frac) {
  const double scale = 1. / ((double)dsize / ssize);
  ofs.resize(dsize); a0.resize(dsize); a1.resize(dsize);
  vmax = dsize;
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = floor_f(f);
    f -= s;
    if (clamp_frac) {          /* x axis: samples off either end collapse onto the edge sample */
      if (s < 0) { f = 0; s = 0; }
      if (s + 1 >= ssize) { vmax = std::min(vmax, d); if (s >= ssize - 1) { f = 0; s = ssize - 1; } }
    }
    ofs[d] = s; a0[d] = 1.f - f; a1[d] = f;
  }
}
void upsample2(const Img& src, Img& dst) {
  dst.create(src.w * 2, src.h * 2);
  std::vector<int> xo, yo; std::vector<float> xa0, xa1, ya0, ya1; int xmax, ymax;
  lin_tab2(src.w, dst.w, xo, xa0, xa1, xmax, true);
  lin_tab2(src.h, dst.h, yo, ya0, ya1, ymax, false);
  auto clip = [](int v, int n) { return v < 0 ? 0 : v >= n ? n - 1 : v; };
  std::vector<float> r0(dst.w), r1(dst.w);
  auto hrow = [&](int sy, std::vector<float>& D) {
    const float* S = &src.d[(size_t)sy * src.w];
    for (int dx = 0; dx < dst.w; dx++)
      D[dx] = dx < xmax ? S[xo[dx]] * xa0[dx] + S[xo[dx] + 1] * xa1[dx] : S[xo[dx]] * 1.f;
  };
  for (int dy = 0; dy < dst.h; dy++) {
    hrow(clip(yo[dy], src.h), r0);
    hrow(clip(yo[dy] + 1, src.h), r1);
    for (int dx = 0; dx < dst.w; dx++) dst.at(dy, dx) = r0[dx] * ya0[dy] + r1[dx] * ya1[dy];
  }
}

/* resize(src, dst, Size(w/2, h/2), INTER_NEAREST): sx = min(floor(dx * (1/(dw/sw))), sw-1) */
void downsample_nearest(const Img& src, Img& dst) {
  dst.create(src.w / 2, src.h / 2);
  const double ifx = 1. / ((double)dst.w / src.w), ify = 1. / ((double)dst.h / src.h);
  for (int y = 0; y < dst.h; y++) {
    const int sy = std::min(floor_d(y * ify), src.h - 1);
    for (int x = 0; x < dst.w; x++) dst.at(y, x) = src.at(sy, std::min(floor_d(x * ifx), src.w - 1));
  }
}

/* getGaussianKernel(n, sigma, CV_32F) with n = cvRound(sigma*4*2 + 1) | 1 (float images) */
void gauss_kernel(double sigma, std::vector<float>& k) {
  const int n = round_d(sigma * 4 * 2 + 1) | 1;
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

/* GaussianBlur(src, dst, Size(), sigma, sigma), float, BORDER_REFLECT_101: row filter s = k0*S0; s (+)= kj*Sj (left to right),
 * column filter s = k_mid*S0; s (+)= k_j*(S[+j] + S[-j]) (the symmetric form).  Which multiply-adds are FUSED is decided by the
 * reference's recorded run (tests/test_capture_golden.py, tools/golden_compare.py): the wheel's AVX2 + FMA3 dispatch of the float
 * filters fuses the 8-lane row body, columns [0, w & ~7), and the 16-column body of the column filter, columns [0, w & ~15);
 * the remaining columns go through the 4-lane / scalar remainders, whose products are rounded before the add.  With exactly
 * this split all 120 recorded pairs are reproduced to the last printed digit.  For the record: no fusion anywhere, 109 of 120
 * within 1e-3 and none exact; fusion everywhere, 120 within 1e-3 and 71 exact; double accumulation, 98 within 1e-3. */
int g_blur_mode = 2;   /* 2: the split above (pinned); 0: no fusion; 1: fusion in every column -- evo_set_sift_blur_mode, for the record */
void gaussian_blur(const Img& src, Img& dst, double sigma) {
  std::vector<float> k;
  gauss_kernel(sigma, k);
  const int n = (int)k.size(), r = n / 2, w = src.w, h = src.h;
  const int wrow = g_blur_mode == 2 ? (w & ~7) : g_blur_mode == 1 ? w : 0, wcol = g_blur_mode == 2 ? (w & ~15) : g_blur_mode == 1 ? w : 0;
  Img tmp; tmp.create(w, h);
  std::vector<int> xi(w + 2 * r);
  for (int i = 0; i < w + 2 * r; i++) xi[i] = reflect101(i - r, w);
  for (int y = 0; y < h; y++) {
    const float* S = &src.d[(size_t)y * w];
    for (int x = 0; x < w; x++) {
      float s = k[0] * S[xi[x]];
      if (x < wrow) { for (int j = 1; j < n; j++) s = fmaf(k[j], S[xi[x + j]], s); }         /* the 8-lane FMA body */
      else { for (int j = 1; j < n; j++) s = s + k[j] * S[xi[x + j]]; }                        /* the scalar remainder */
      tmp.at(y, x) = s;
    }
  }
  Img out; out.create(w, h);
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      float s = k[r] * tmp.at(y, x);
      if (x < wcol) {                                                                     /* the 16-column FMA body */
        for (int j = 1; j <= r; j++) s = fmaf(k[r + j], tmp.at(reflect101(y + j, h), x) + tmp.at(reflect101(y - j, h), x), s);
      } else {                                                                                 /* 4-lane and scalar remainders */
        for (int j = 1; j <= r; j++) s = s + k[r + j] * (tmp.at(reflect101(y + j, h), x) + tmp.at(reflect101(y - j, h), x));
      }
      out.at(y, x) = s;
    }
  }
  dst = out;
}

/* cv::hal::exp32f (the float-polynomial body) */
const float kExpTab[64] = {
#include "sift_exptab.inc"
};
float exp32f(float x) {
  const double A0 = .9670371139572337719125840413672004409288e-2;
  const float A4 = (float)(1.000000000000002438532970795181890933776 / A0),
              A3 = (float)(.6931471805521448196800669615864773144641 / A0),
              A2 = (float)(.2402265109513301490103372422686535526573 / A0),
              A1 = (float)(.5550339366753125211915322047004666939128e-1 / A0);
  const double exp_prescale = 1.4426950408889634073599246810019 * 64, exp_postscale = 1. / 64, exp_max_val = 3000. * 64;
  const float minval = (float)(-exp_max_val / exp_prescale), maxval = (float)(exp_max_val / exp_prescale);
  float xc = std::min(std::max(x, minval), maxval);
  double xd = (double)xc * exp_prescale;
  int xi = (int)lrint(xd);
  float xf = (float)(xd - (double)xi) * (float)exp_postscale;
  int e = (xi >> 6) + 127;
  e = e < 0 ? 0 : e > 255 ? 255 : e;
  uint32_t bits = (uint32_t)e << 23;
  float p2; memcpy(&p2, &bits, 4);
  float yf = kExpTab[xi & 63] * p2;
  float zf = xf + A1;
  zf = zf * xf + A2; zf = zf * xf + A3; zf = zf * xf + A4;
  return zf * yf;
}

/* 2^x for the key-point size (sift.cpp uses powf(2.f, x)): double Taylor series of e^(f ln 2), one IEEE operation at
 * a time, rounded to float */
float det_exp2(float x) {
  const double xd = (double)x;
  const double fn = std::nearbyint(xd);
  const double r = (xd - fn) * 0.6931471805599453094;
  double p = 1.0 / 6227020800.0;                    /* 1/13! */
  const double inv[13] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                          1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
  for (int i = 0; i < 13; i++) p = p * r + inv[i];
  return (float)std::ldexp(p, (int)fn);
}

struct KP { float x, y, size, angle, response; int octave; };

struct Pyr {
  int nOctaves = 0;
  std::vector<Img> g;     /* nOctaves * 6 */
  std::vector<Img> dog;   /* nOctaves * 5 */
};

void build_pyramids(const uint8_t* gray, int w, int h, Pyr& P) {
  Img f; f.create(w, h);
  for (size_t i = 0; i < (size_t)w * h; i++) f.d[i] = (float)gray[i];
  Img base;
  upsample2(f, base);
  const float sigma = (float)SIGMA;
  const float sig_diff = sqrtf(std::max(sigma * sigma - SIFT_INIT_SIGMA * SIFT_INIT_SIGMA * 4, 0.01f));
  gaussian_blur(base, base, sig_diff);
  P.nOctaves = round_d(std::log((double)std::min(base.w, base.h)) / std::log(2.) - 2) + 1;
  const int L = N_OCTAVE_LAYERS;
  std::vector<double> sig(L + 3);
  sig[0] = SIGMA;
  const double k = std::pow(2., 1. / L);
  for (int i = 1; i < L + 3; i++) {
    const double sig_prev = std::pow(k, (double)(i - 1)) * SIGMA, sig_total = sig_prev * k;
    sig[i] = std::sqrt(sig_total * sig_total - sig_prev * sig_prev);
  }
  P.g.resize((size_t)P.nOctaves * (L + 3));
  for (int o = 0; o < P.nOctaves; o++)
    for (int i = 0; i < L + 3; i++) {
      Img& dst = P.g[o * (L + 3) + i];
      if (o == 0 && i == 0) dst = base;
      else if (i == 0) downsample_nearest(P.g[(o - 1) * (L + 3) + L], dst);
      else gaussian_blur(P.g[o * (L + 3) + i - 1], dst, sig[i]);
    }
  P.dog.resize((size_t)P.nOctaves * (L + 2));
  for (int o = 0; o < P.nOctaves; o++)
    for (int i = 0; i < L + 2; i++) {
      const Img& a = P.g[o * (L + 3) + i]; const Img& b = P.g[o * (L + 3) + i + 1];
      Img& d = P.dog[o * (L + 2) + i];
      d.create(a.w, a.h);
      for (size_t j = 0; j < d.d.size(); j++) d.d[j] = b.d[j] - a.d[j];
    }
}

bool adjust_local_extrema(const Pyr& P, KP& kpt, int octv, int& layer, int& r, int& c) {
  const int L = N_OCTAVE_LAYERS;
  const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale,
              cross_deriv_scale = img_scale * 0.25f;
  float xi = 0, xr = 0, xc = 0, contr = 0;
  int i = 0;
  for (; i < SIFT_MAX_INTERP_STEPS; i++) {
    const int idx = octv * (L + 2) + layer;
    const Img& img = P.dog[idx]; const Img& prev = P.dog[idx - 1]; const Img& next = P.dog[idx + 1];
    const float dD[3] = {(img.at(r, c + 1) - img.at(r, c - 1)) * deriv_scale, (img.at(r + 1, c) - img.at(r - 1, c)) * deriv_scale,
                         (next.at(r, c) - prev.at(r, c)) * deriv_scale};
    const float v2 = img.at(r, c) * 2;
    const float dxx = (img.at(r, c + 1) + img.at(r, c - 1) - v2) * second_deriv_scale;
    const float dyy = (img.at(r + 1, c) + img.at(r - 1, c) - v2) * second_deriv_scale;
    const float dss = (next.at(r, c) + prev.at(r, c) - v2) * second_deriv_scale;
    const float dxy = (img.at(r + 1, c + 1) - img.at(r + 1, c - 1) - img.at(r - 1, c + 1) + img.at(r - 1, c - 1)) * cross_deriv_scale;
    const float dxs = (next.at(r, c + 1) - next.at(r, c - 1) - prev.at(r, c + 1) + prev.at(r, c - 1)) * cross_deriv_scale;
    const float dys = (next.at(r + 1, c) - next.at(r - 1, c) - prev.at(r + 1, c) + prev.at(r - 1, c)) * cross_deriv_scale;
    /* Matx33f H(dxx,dxy,dxs, dxy,dyy,dys, dxs,dys,dss); X = H.solve(dD, DECOMP_LU) = Cramer's rule in float */
    const float a00 = dxx, a01 = dxy, a02 = dxs, a10 = dxy, a11 = dyy, a12 = dys, a20 = dxs, a21 = dys, a22 = dss;
    const float b0 = dD[0], b1 = dD[1], b2 = dD[2];
    float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
    float X0 = 0, X1 = 0, X2 = 0;
    if (d != 0) {
      d = 1 / d;
      X0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
      X1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
      X2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
    }
    xi = -X2; xr = -X1; xc = -X0;
    if (std::fabs(xi) < 0.5f && std::fabs(xr) < 0.5f && std::fabs(xc) < 0.5f) break;
    if (std::fabs(xi) > (float)(INT_MAX / 3) || std::fabs(xr) > (float)(INT_MAX / 3) || std::fabs(xc) > (float)(INT_MAX / 3))
      return false;
    c += round_f(xc); r += round_f(xr); layer += round_f(xi);
    if (layer < 1 || layer > L || c < SIFT_IMG_BORDER || c >= img.w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER ||
        r >= img.h - SIFT_IMG_BORDER)
      return false;
  }
  if (i >= SIFT_MAX_INTERP_STEPS) return false;
  {
    const int idx = octv * (L + 2) + layer;
    const Img& img = P.dog[idx]; const Img& prev = P.dog[idx - 1]; const Img& next = P.dog[idx + 1];
    const float dD[3] = {(img.at(r, c + 1) - img.at(r, c - 1)) * deriv_scale, (img.at(r + 1, c) - img.at(r - 1, c)) * deriv_scale,
                         (next.at(r, c) - prev.at(r, c)) * deriv_scale};
    float t = 0;
    t += dD[0] * xc; t += dD[1] * xr; t += dD[2] * xi;
    contr = img.at(r, c) * img_scale + t * 0.5f;
    if (std::fabs(contr) * L < (float)CONTRAST_THRESHOLD) return false;
    const float v2 = img.at(r, c) * 2.f;
    const float dxx = (img.at(r, c + 1) + img.at(r, c - 1) - v2) * second_deriv_scale;
    const float dyy = (img.at(r + 1, c) + img.at(r - 1, c) - v2) * second_deriv_scale;
    const float dxy = (img.at(r + 1, c + 1) - img.at(r + 1, c - 1) - img.at(r - 1, c + 1) + img.at(r - 1, c - 1)) * cross_deriv_scale;
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    const float et = (float)EDGE_THRESHOLD;
    if (det <= 0 || tr * tr * et >= (et + 1) * (et + 1) * det) return false;
  }
  kpt.x = (c + xc) * (1 << octv);
  kpt.y = (r + xr) * (1 << octv);
  kpt.octave = octv + (layer << 8) + (round_d((xi + 0.5) * 255) << 16);
  kpt.size = (float)SIGMA * det_exp2((layer + xi) / L) * (1 << octv) * 2;
  kpt.response = std::fabs(contr);
  return true;
}

float calc_orientation_hist(const Img& img, int px, int py, int radius, float sigma, float* hist, int n) {
  const float expf_scale = -1.f / (2.f * sigma * sigma);
  std::vector<float> temp(n + 4, 0.f);
  float* temphist = temp.data() + 2;
  for (int i = -radius; i <= radius; i++) {
    const int y = py + i;
    if (y <= 0 || y >= img.h - 1) continue;
    for (int j = -radius; j <= radius; j++) {
      const int x = px + j;
      if (x <= 0 || x >= img.w - 1) continue;
      const float dx = img.at(y, x + 1) - img.at(y, x - 1);
      const float dy = img.at(y - 1, x) - img.at(y + 1, x);
      const float W = exp32f((i * i + j * j) * expf_scale);
      const float Ori = evo_fast_atan2(dy, dx);
      const float Mag = std::sqrt(dx * dx + dy * dy);
      int bin = round_f((n / 360.f) * Ori);
      if (bin >= n) bin -= n;
      if (bin < 0) bin += n;
      temphist[bin] += W * Mag;
    }
  }
  temphist[-1] = temphist[n - 1]; temphist[-2] = temphist[n - 2];
  temphist[n] = temphist[0]; temphist[n + 1] = temphist[1];
  for (int i = 0; i < n; i++)
    hist[i] = (temphist[i - 2] + temphist[i + 2]) * (1.f / 16.f) + (temphist[i - 1] + temphist[i + 1]) * (4.f / 16.f) +
              temphist[i] * (6.f / 16.f);
  float maxval = hist[0];
  for (int i = 1; i < n; i++) maxval = std::max(maxval, hist[i]);
  return maxval;
}

void find_extrema(const Pyr& P, std::vector<KP>& kps) {
  const int L = N_OCTAVE_LAYERS, n = SIFT_ORI_HIST_BINS;
  const int threshold = floor_d(0.5 * CONTRAST_THRESHOLD / L * 255);
  float hist[SIFT_ORI_HIST_BINS];
  for (int o = 0; o < P.nOctaves; o++)
    for (int i = 1; i <= L; i++) {
      const int idx = o * (L + 2) + i;
      const Img& img = P.dog[idx]; const Img& prev = P.dog[idx - 1]; const Img& next = P.dog[idx + 1];
      const int rows = img.h, cols = img.w;
      for (int r = SIFT_IMG_BORDER; r < rows - SIFT_IMG_BORDER; r++)
        for (int c = SIFT_IMG_BORDER; c < cols - SIFT_IMG_BORDER; c++) {
          const float val = img.at(r, c);
          if (!(std::fabs(val) > threshold)) continue;
          bool ismax = val > 0, ismin = val < 0;
          for (int dr = -1; dr <= 1 && (ismax || ismin); dr++)
            for (int dc = -1; dc <= 1; dc++) {
              const float a = img.at(r + dr, c + dc), b = prev.at(r + dr, c + dc), d = next.at(r + dr, c + dc);
              if (!(val >= a && val >= b && val >= d)) ismax = false;
              if (!(val <= a && val <= b && val <= d)) ismin = false;
            }
          if (!(ismax || ismin)) continue;
          KP kpt{};
          int r1 = r, c1 = c, layer = i;
          if (!adjust_local_extrema(P, kpt, o, layer, r1, c1)) continue;
          const float scl_octv = kpt.size * 0.5f / (1 << o);
          const float omax = calc_orientation_hist(P.g[o * (L + 3) + layer], c1, r1, round_f(SIFT_ORI_RADIUS * scl_octv),
                                                   SIFT_ORI_SIG_FCTR * scl_octv, hist, n);
          const float mag_thr = (float)(omax * SIFT_ORI_PEAK_RATIO);
          for (int j = 0; j < n; j++) {
            const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
            if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
              float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
              bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
              kpt.angle = 360.f - (float)((360.f / n) * bin);
              if (std::fabs(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
              kps.push_back(kpt);
            }
          }
        }
    }
}

/* KeyPointsFilter::removeDuplicatedSorted: sort by (x, y, size desc, angle, response desc, octave desc), then drop
 * every key point equal to its predecessor in (x, y, size, angle) */
bool kp_less(const KP& a, const KP& b) {
  if (a.x != b.x) return a.x < b.x;
  if (a.y != b.y) return a.y < b.y;
  if (a.size != b.size) return a.size > b.size;
  if (a.angle != b.angle) return a.angle < b.angle;
  if (a.response != b.response) return a.response > b.response;
  if (a.octave != b.octave) return a.octave > b.octave;
  return false;
}
void remove_duplicated_sorted(std::vector<KP>& k) {
  const int n = (int)k.size();
  if (n < 2) return;
  std::stable_sort(k.begin(), k.end(), kp_less);
  int i = 0;
  for (int j = 1; j < n; j++)
    if (k[i].x != k[j].x || k[i].y != k[j].y || k[i].size != k[j].size || k[i].angle != k[j].angle) k[++i] = k[j];
  k.resize(i + 1);
}

void calc_descriptor(const Img& img, float ptx, float pty, float ori, float scl, uint8_t* out) {
  const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
  const int px = round_f(ptx), py = round_f(pty);
  double sd, cd;
  evo_sincos((double)(ori * (float)(M_PI / 180)), &sd, &cd);
  float cos_t = (float)cd, sin_t = (float)sd;
  const float bins_per_rad = n / 360.f;
  const float exp_scale = -1.f / (d * d * 0.5f);
  const float hist_width = SIFT_DESCR_SCL_FCTR * scl;
  int radius = round_f(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
  radius = std::min(radius, (int)std::sqrt(((double)img.w) * img.w + ((double)img.h) * img.h));
  cos_t /= hist_width; sin_t /= hist_width;
  const int rows = img.h, cols = img.w;
  float hist[(SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_HIST_BINS + 2)];
  for (float& v : hist) v = 0.f;
  for (int i = -radius; i <= radius; i++)
    for (int j = -radius; j <= radius; j++) {
      const float c_rot = j * cos_t - i * sin_t;
      const float r_rot = j * sin_t + i * cos_t;
      float rbin = r_rot + d / 2 - 0.5f;
      float cbin = c_rot + d / 2 - 0.5f;
      const int r = py + i, c = px + j;
      if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
        const float dx = img.at(r, c + 1) - img.at(r, c - 1);
        const float dy = img.at(r - 1, c) - img.at(r + 1, c);
        const float Ori = evo_fast_atan2(dy, dx);
        const float Mag = std::sqrt(dx * dx + dy * dy);
        const float W = exp32f((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        float obin = (Ori - ori) * bins_per_rad;
        const float mag = Mag * W;
        const int r0 = floor_f(rbin), c0 = floor_f(cbin);
        int o0 = floor_f(obin);
        rbin -= r0; cbin -= c0; obin -= o0;
        if (o0 < 0) o0 += n;
        if (o0 >= n) o0 -= n;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
        const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
        const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
        const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
        const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
        hist[idx] += v_rco000; hist[idx + 1] += v_rco001;
        hist[idx + (n + 2)] += v_rco010; hist[idx + (n + 3)] += v_rco011;
        hist[idx + (d + 2) * (n + 2)] += v_rco100; hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
        hist[idx + (d + 3) * (n + 2)] += v_rco110; hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
      }
    }
  float dst[128];
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) {
      const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
      hist[idx] += hist[idx + n];
      hist[idx + 1] += hist[idx + n + 1];
      for (int k = 0; k < n; k++) dst[(i * d + j) * n + k] = hist[idx + k];
    }
  float nrm2 = 0;
  const int len = d * d * n;
  for (int k = 0; k < len; k++) nrm2 += dst[k] * dst[k];
  const float thr = std::sqrt(nrm2) * SIFT_DESCR_MAG_THR;
  nrm2 = 0;
  for (int i = 0; i < len; i++) { const float val = std::min(dst[i], thr); dst[i] = val; nrm2 += val * val; }
  nrm2 = SIFT_INT_DESCR_FCTR / std::max(std::sqrt(nrm2), FLT_EPSILON);
  for (int k = 0; k < len; k++) {
    const int v = round_f(dst[k] * nrm2);
    out[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
  }
}

}  // namespace

/* number of octaves and their sizes for a w x h frame (octave 0 = the doubled image) */
extern "C" int evo_sift_layout(int w, int h, int* ow, int* oh, int cap) {
  const int bw = 2 * w, bh = 2 * h;
  const int n = round_d(std::log((double)std::min(bw, bh)) / std::log(2.) - 2) + 1;
  int cw = bw, ch = bh;
  for (int o = 0; o < n && o < cap; o++) { ow[o] = cw; oh[o] = ch; cw /= 2; ch /= 2; }
  return n;
}

/* the Gaussian scale space, every octave's 6 layers tightly packed one after another (octave-major); returns floats */
extern "C" int64_t evo_sift_gauss_pyramid(const uint8_t* gray, int w, int h, float* out, int64_t cap) {
  Pyr P;
  build_pyramids(gray, w, h, P);
  int64_t n = 0;
  for (const Img& g : P.g) {
    if (out && n + (int64_t)g.d.size() <= cap) memcpy(out + n, g.d.data(), g.d.size() * sizeof(float));
    n += (int64_t)g.d.size();
  }
  return n;
}

/* cv2.xfeatures2d.SIFT_create().detectAndCompute(gray, None): key points in the operator's own order
 * (removeDuplicatedSorted), pt / size already scaled back to the input frame.  desc holds the descriptor VALUES
 * (0..255; the operator returns them as float32).  Returns the count (only the first cap are written). */
extern "C" int evo_sift_detect(const uint8_t* gray, int w, int h, float* xy, uint8_t* desc, int* octave, float* size,
                               float* angle, float* response, int cap) {
  Pyr P;
  build_pyramids(gray, w, h, P);
  std::vector<KP> kps;
  find_extrema(P, kps);
  remove_duplicated_sorted(kps);
  const int firstOctave = -1;
  for (KP& k : kps) {
    const float scale = 1.f / (float)(1 << -firstOctave);
    k.octave = (k.octave & ~255) | ((k.octave + firstOctave) & 255);
    k.x *= scale; k.y *= scale; k.size *= scale;
  }
  const int L = N_OCTAVE_LAYERS;
  for (int i = 0; i < (int)kps.size() && i < cap; i++) {
    const KP& k = kps[i];
    int oct = k.octave & 255, layer = (k.octave >> 8) & 255;
    oct = oct < 128 ? oct : (-128 | oct);
    const float scale = oct >= 0 ? 1.f / (1 << oct) : (float)(1 << -oct);
    const float sz = k.size * scale;
    float ang = 360.f - k.angle;
    if (std::fabs(ang - 360.f) < FLT_EPSILON) ang = 0.f;
    if (desc)
      calc_descriptor(P.g[(oct - firstOctave) * (L + 3) + layer], k.x * scale, k.y * scale, ang, sz * 0.5f, desc + (size_t)i * 128);
    if (xy) { xy[2 * i] = k.x; xy[2 * i + 1] = k.y; }
    if (octave) octave[i] = k.octave;
    if (size) size[i] = k.size;
    if (angle) angle[i] = k.angle;
    if (response) response[i] = k.response;
  }
  return (int)kps.size();
}

extern "C" float evo_sift_exp32f(float x) { return exp32f(x); }
extern "C" void evo_set_sift_blur_mode(int mode) { g_blur_mode = (mode >= 0 && mode <= 2) ? mode : 2; }
extern "C" int evo_get_sift_blur_mode(void) { return g_blur_mode; }
extern "C" float evo_sift_exp2(float x) { return det_exp2(x); }

/* BruteForce (NORM_L2) 2-NN on float descriptors, matching.py:102-108 on float32[N,dim]: hal::normL2Sqr_ (two 4-lane
 * partial sums over steps of 8, then (d0+d1) lanes added left to right, scalar tail), dist = sqrt, K = 2 insertion on
 * strictly smaller float distance.  idx = -1 where fewer than 2 train rows exist. */
extern "C" void evo_knn2_l2f32(const float* q, int nq, const float* t, int nt, int dim, int32_t* idx, float* dist) {
  for (int i = 0; i < nq; i++) {
    float best[2] = {FLT_MAX, FLT_MAX};
    int bi[2] = {-1, -1};
    const float* a = q + (size_t)i * dim;
    for (int j = 0; j < nt; j++) {
      const float* b = t + (size_t)j * dim;
      float d0[4] = {0, 0, 0, 0}, d1[4] = {0, 0, 0, 0};
      int k = 0;
      for (; k <= dim - 8; k += 8)
        for (int l = 0; l < 4; l++) {
          const float t0 = a[k + l] - b[k + l], t1 = a[k + 4 + l] - b[k + 4 + l];
          d0[l] = d0[l] + t0 * t0;
          d1[l] = d1[l] + t1 * t1;
        }
      float buf[4];
      for (int l = 0; l < 4; l++) buf[l] = d0[l] + d1[l];
      float d = buf[0] + buf[1] + buf[2] + buf[3];
      for (; k < dim; k++) { const float tt = a[k] - b[k]; d += tt * tt; }
      const float ds = std::sqrt(d);
      if (ds < best[1]) {
        int kk;
        for (kk = 0; kk >= 0 && best[kk] > ds; kk--) { bi[kk + 1] = bi[kk]; best[kk + 1] = best[kk]; }
        bi[kk + 1] = j; best[kk + 1] = ds;
      }
    }
    idx[2 * i] = bi[0]; idx[2 * i + 1] = bi[1];
    dist[2 * i] = best[0]; dist[2 * i + 1] = best[1];
  }
}

/* ---- multi-type pairs: FrameProcessing(frame, features_type_list).concatenate_all_features_types (frame_processing.py:91-104)
 * followed by compute_homography / matrix_superposition as get_homography_dict drives them (video_processing.py:67-105) ---- */
extern "C" int evo_ratio_unique_f32(const int32_t* idx, const float* dist, int nq, double ratio, int32_t* out_q, int32_t* out_t) {
  std::vector<int> sq, st;
  std::vector<int> claims;
  for (int i = 0; i < nq; i++) {
    if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) continue;                       /* len(matches) != 2 */
    if ((double)dist[2 * i] < (double)dist[2 * i + 1] * ratio) {              /* matching.py:190 on Python floats */
      sq.push_back(i); st.push_back(idx[2 * i]);
      if ((int)claims.size() <= idx[2 * i]) claims.resize(idx[2 * i] + 1, 0);
      claims[idx[2 * i]]++;
    }
  }
  int m = 0;
  for (size_t k = 0; k < sq.size(); k++)
    if (claims[st[k]] == 1) { out_q[m] = sq[k]; out_t[m] = st[k]; m++; }
  return m;
}

/* KeyPoints.match_static_kps (matching.py:131-163) on float descriptors */
extern "C" int evo_match_static_f32_ex(const float* xy_a, const float* desc_a, int na, const float* xy_b, const float* desc_b, int nb,
                                       int dim, int force_max, float* oa, float* ob, int* out_n) {
  *out_n = 0;
  if (na == 0 || nb == 0) return EVO_NO_DESCRIPTORS;
  std::vector<int32_t> idx(2 * (size_t)na), mq(na), mt(na);
  std::vector<float> dist(2 * (size_t)na);
  evo_knn2_l2f32(desc_a, na, desc_b, nb, dim, idx.data(), dist.data());
  const int m = evo_ratio_unique_f32(idx.data(), dist.data(), na, 0.5, mq.data(), mt.data());
  if (m < 4) return EVO_FEW_MATCHES;
  std::vector<float> pa(2 * (size_t)m), pb(2 * (size_t)m), ua(2 * (size_t)m), ub(2 * (size_t)m);
  for (int i = 0; i < m; i++) {
    pa[2 * i] = xy_a[2 * mq[i]]; pa[2 * i + 1] = xy_a[2 * mq[i] + 1];
    pb[2 * i] = xy_b[2 * mt[i]]; pb[2 * i + 1] = xy_b[2 * mt[i] + 1];
  }
  const int u = evo_remove_double(pa.data(), pb.data(), m, ua.data(), ub.data());
  double H[9];
  std::vector<uint8_t> mask(u);
  if (!evo_find_homography_ex(ua.data(), ub.data(), u, 3.0, 2000, 0.995, force_max, H, mask.data(), nullptr)) return EVO_NO_PROVISIONAL_H;
  *out_n = evo_static_filter(H, ua.data(), ub.data(), u, oa, ob);
  return EVO_OK;
}
extern "C" int evo_match_static_f32(const float* xy_a, const float* desc_a, int na, const float* xy_b, const float* desc_b, int nb,
                                    int dim, float* oa, float* ob, int* out_n) {
  return evo_match_static_f32_ex(xy_a, desc_a, na, xy_b, desc_b, nb, dim, 0, oa, ob, out_n);
}

namespace {
struct TFeat { std::vector<float> xy; std::vector<uint8_t> d8; std::vector<float> df; int n = 0; };
void detect_type(const uint8_t* gray, int w, int h, int nfeatures, int type, TFeat& f) {
  if (type == 1) {                                           /* SIFT */
    int cap = 65536;
    f.xy.resize(2 * (size_t)cap); f.d8.resize((size_t)128 * cap);
    f.n = evo_sift_detect(gray, w, h, f.xy.data(), f.d8.data(), nullptr, nullptr, nullptr, nullptr, cap);
    if (f.n > cap) f.n = cap;
    f.df.resize((size_t)128 * f.n);
    for (size_t i = 0; i < f.df.size(); i++) f.df[i] = (float)f.d8[i];
  } else if (type == 2) {                                    /* SURF */
    int cap = 65536;
    f.xy.resize(2 * (size_t)cap); f.df.resize((size_t)128 * cap);
    f.n = evo_surf_detect(gray, w, h, f.xy.data(), f.df.data(), nullptr, nullptr, nullptr, nullptr, nullptr, cap);
    if (f.n > cap) f.n = cap;
  } else {                                                   /* ORB */
    int cap = nfeatures * 2 + 4096;
    f.xy.resize(2 * (size_t)cap); f.d8.resize((size_t)32 * cap);
    std::vector<int> oc(cap), lx(cap), ly(cap); std::vector<float> rs(cap), an(cap);
    f.n = evo_orb_detect(gray, w, h, nfeatures, f.xy.data(), f.d8.data(), oc.data(), lx.data(), ly.data(), rs.data(), an.data(), cap);
  }
}
}  // namespace

/* one stream, a list of feature types (0 = ORB, 1 = SIFT, 2 = SURF) processed in list order; H [F-1][9], status [F-1]; returns the
 * index of a failing FIRST pair or -1 (as evo_stream_gray) */
/* _ex: force_max = every RANSAC runs its 2000 iterations (BASELINE configs[2]); Hsup_forced (or NULL) [F-1][9] replaces the
 * running superposition: pair k (k >= 1, 0-based) is solved in the plane Hsup_forced[k-1] instead of the plane the stream itself
 * accumulated -- used to compare pair by pair with a recorded reference run without the drift of earlier pairs; npts (or NULL)
 * receives the number of point pairs handed to the final RANSAC per pair. */
extern "C" int evo_stream_gray_types_ex(const uint8_t* frames, int nframes, int w, int h, int nfeatures, const int* types,
                                        int ntypes, int force_max, const double* Hsup_forced, double* H, int* status, int* npts) {
  const size_t fs = (size_t)w * h;
  std::vector<TFeat> prev(ntypes), cur(ntypes);
  for (int t = 0; t < ntypes; t++) detect_type(frames, w, h, nfeatures, types[t], prev[t]);
  double Hsup[9], Hprev[9];
  bool first = true, have_prev = false;
  for (int k = 1; k < nframes; k++) {
    for (int t = 0; t < ntypes; t++) detect_type(frames + (size_t)k * fs, w, h, nfeatures, types[t], cur[t]);
    double* Hk = H + 9 * (size_t)(k - 1);
    int st = EVO_OK;
    std::vector<float> alla, allb;
    for (int t = 0; t < ntypes && st == EVO_OK; t++) {      /* a NoMatchesException of one type propagates */
      const TFeat& a = cur[t]; const TFeat& b = prev[t];
      std::vector<float> oa(2 * (size_t)std::max(a.n, 1)), ob(2 * (size_t)std::max(a.n, 1));
      int n = 0;
      if (types[t] != 0) st = evo_match_static_f32_ex(a.xy.data(), a.df.data(), a.n, b.xy.data(), b.df.data(), b.n, 128, force_max, oa.data(), ob.data(), &n);
      else st = evo_match_static_ex(a.xy.data(), a.d8.data(), a.n, b.xy.data(), b.d8.data(), b.n, force_max, oa.data(), ob.data(), &n);
      if (st == EVO_OK) { alla.insert(alla.end(), oa.begin(), oa.begin() + 2 * n); allb.insert(allb.end(), ob.begin(), ob.begin() + 2 * n); }
    }
    if (npts) npts[k - 1] = 0;
    if (st == EVO_OK) {
      const int n = (int)alla.size() / 2;
      std::vector<float> ua(alla.size() + 2), ub(alla.size() + 2);
      const int u = evo_remove_double(alla.data(), allb.data(), n, ua.data(), ub.data());
      if (npts) npts[k - 1] = u;
      if (Hsup_forced && k >= 2) memcpy(Hsup, Hsup_forced + 9 * (size_t)(k - 2), sizeof(Hsup));
      st = evo_compute_homography_ex(ua.data(), ub.data(), u, first ? nullptr : Hsup, force_max, Hk);
    }
    status[k - 1] = st;
    if (st != EVO_OK) {
      if (!have_prev) return k - 1;
      memcpy(Hk, Hprev, sizeof(Hprev));
    }
    double S[9];
    evo_matrix_superposition(Hk, Hsup, first ? 1 : 0, S);
    memcpy(Hsup, S, sizeof(S));
    memcpy(Hprev, Hk, sizeof(Hprev));
    first = false; have_prev = true;
    std::swap(prev, cur);
  }
  return -1;
}

extern "C" int evo_stream_gray_types(const uint8_t* frames, int nframes, int w, int h, int nfeatures, const int* types,
                                     int ntypes, double* H, int* status) {
  return evo_stream_gray_types_ex(frames, nframes, w, h, nfeatures, types, ntypes, 0, nullptr, H, status, nullptr);
}
