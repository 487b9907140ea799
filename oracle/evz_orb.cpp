/*
 * oracle/evz_orb.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), image + ORB half.
 *
 * Restates what the reference executes at
 *   evenvizion/processing/video_processing.py:62,73   imutils.resize(frame, width=) -> cv2.resize(INTER_AREA)
 *   evenvizion/processing/frame_processing.py:59-61   cv2.ORB_create().detectAndCompute(frame, None)
 * The arithmetic lives in opencv-contrib-python==3.4.2.17 (requirements.txt:3), which is not part of
 * /root/reference; what follows restates the published OpenCV 3.4 algorithm (imgproc resize/color/smooth,
 * features2d fast/orb).  No reference TEST pins these operators one by one; they are pinned jointly, since round 4, by the reference's own video and recorded result (tests/test_capture_golden.py: all 120 matrices of dict_with_homography_matrix.json reproduced to the last digit) -- see evz_oracle.h.
 *
 * Compile with -ffp-contract=off: float expressions below are evaluated one IEEE operation at a time.
 */
#include "evz_oracle.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline int round_f(float v) { return (int)lrintf(v); }   // cvRound(float): round-half-even
inline int round_d(double v) { return (int)lrint(v); }   // cvRound(double)
inline int floor_d(double v) { int i = (int)v; return i - (i > v); }
inline int ceil_d(double v) { int i = (int)v; return i + (i < v); }
inline uint8_t sat_u8_f(float v) { int i = round_f(v); return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i); }
inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
  return p;
}

const int kPattern[256 * 4] = {
#include "orb_pattern.inc"
};

}  // namespace

/* ------------------------------------------------------------------------------------------------ */
/* cvtColor(BGR2GRAY), 8-bit: Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14                             */
extern "C" void evo_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray) {
  for (int y = 0; y < h; y++) {
    const uint8_t* s = bgr + (size_t)y * stride;
    uint8_t* d = gray + (size_t)y * w;
    for (int x = 0; x < w; x++)
      d[x] = (uint8_t)((s[3 * x] * 1868 + s[3 * x + 1] * 9617 + s[3 * x + 2] * 4899 + 8192) >> 14);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* cv2.resize(..., INTER_AREA) (imutils.resize, video_processing.py:62,73): area sums when shrinking, the    */
/* bilinear emulation when enlarging                                                                   */
namespace {
struct DecAlpha { int si, di; float alpha; };
void area_tab(int ssize, int dsize, double scale, std::vector<DecAlpha>& tab) {
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    double cell = std::min(scale, ssize - fsx1);
    int sx1 = ceil_d(fsx1), sx2 = floor_d(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cell)});
    for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, dx, (float)(1.0 / cell)});
    if (fsx2 - sx2 > 1e-3)
      tab.push_back({sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
  }
}
}  // namespace

extern "C" int evo_resize_area(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh) {
  if (dw == sw && dh == sh) { memcpy(dst, src, (size_t)sw * sh * cn); return 0; }
  double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
  double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
  if (scale_x < 1 || scale_y < 1) {
    // "true area interpolation is only implemented for scale_x >= 1 && scale_y >= 1; in other cases it is emulated
    // using some variant of bilinear interpolation" (imgproc resize): INTER_AREA then runs the 8-bit INTER_LINEAR
    // machinery (11-bit coefficients, HResizeLinear / VResizeLinear) with area-mode coefficient tables:
    //   sx = floor(dx*scale), fx = (float)((dx+1) - (sx+1)*inv_scale), fx = fx <= 0 ? 0 : fx - floor(fx).
    // Reached by imutils.resize(frame, width=) with width > frame width (video_processing.py:62,73).
    std::vector<int> xo(dw), yo(dh);
    std::vector<short> xa(2 * (size_t)dw), ya(2 * (size_t)dh);
    auto tab = [](int ssize, int dsize, double scale, double inv, int* ofs, short* al, int& vmax) {
      vmax = dsize;
      for (int d = 0; d < dsize; d++) {
        int s = floor_d(d * scale);
        float f = (float)((d + 1) - (s + 1) * inv);
        f = f <= 0 ? 0.f : f - (float)floor_d((double)f);
        if (s < 0) { f = 0; s = 0; }
        if (s + 1 >= ssize) { vmax = std::min(vmax, d); if (s >= ssize - 1) { f = 0; s = ssize - 1; } }
        ofs[d] = s;
        const float c0 = 1.f - f, c1 = f;
        auto sat = [](float v) { int i = round_f(v); return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i); };
        al[2 * d] = sat(c0 * 2048); al[2 * d + 1] = sat(c1 * 2048);
      }
    };
    int xmax, ymax;
    tab(sw, dw, scale_x, inv_x, xo.data(), xa.data(), xmax);
    tab(sh, dh, scale_y, inv_y, yo.data(), ya.data(), ymax);
    (void)ymax;
    std::vector<int> r0((size_t)dw * cn), r1((size_t)dw * cn);
    auto hrow = [&](int sy, std::vector<int>& D) {
      const uint8_t* S = src + (size_t)sy * sw * cn;
      for (int dx = 0; dx < dw; dx++)
        for (int c = 0; c < cn; c++) {
          const int sx = xo[dx];
          D[dx * cn + c] = dx < xmax ? S[sx * cn + c] * xa[2 * dx] + S[(sx + 1) * cn + c] * xa[2 * dx + 1]
                                     : S[sx * cn + c] * 2048;
        }
    };
    auto clip = [](int v, int a, int b) { return v >= a ? (v < b ? v : b - 1) : a; };
    for (int dy = 0; dy < dh; dy++) {
      hrow(clip(yo[dy], 0, sh), r0);
      hrow(clip(yo[dy] + 1, 0, sh), r1);
      const int b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
      uint8_t* D = dst + (size_t)dy * dw * cn;
      for (int i = 0; i < dw * cn; i++) D[i] = (uint8_t)((((b0 * (r0[i] >> 4)) >> 16) + ((b1 * (r1[i] >> 4)) >> 16) + 2) >> 2);
    }
    return 0;
  }
  int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
  bool fast = std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON;
  if (fast) {
    int area = isx * isy;
    float scale = 1.f / area;
    for (int dy = 0; dy < dh; dy++)
      for (int dx = 0; dx < dw; dx++)
        for (int c = 0; c < cn; c++) {
          int sum = 0;
          for (int j = 0; j < isy; j++)
            for (int i = 0; i < isx; i++) sum += src[((size_t)(dy * isy + j) * sw + dx * isx + i) * cn + c];
          uint8_t v;
          if (isx == 2 && isy == 2) v = (uint8_t)((sum + 2) >> 2);  // 2x2 fast vector path rounds half up
          else v = sat_u8_f(sum * scale);
          dst[((size_t)dy * dw + dx) * cn + c] = v;
        }
    return 0;
  }
  std::vector<DecAlpha> xt, yt;
  area_tab(sw, dw, scale_x, xt);
  area_tab(sh, dh, scale_y, yt);
  std::vector<float> buf((size_t)dw * cn), sum((size_t)dw * cn, 0.f);
  int prev_dy = yt.empty() ? 0 : yt[0].di;
  for (size_t j = 0; j < yt.size(); j++) {
    float beta = yt[j].alpha;
    int dy = yt[j].di, sy = yt[j].si;
    const uint8_t* S = src + (size_t)sy * sw * cn;
    std::fill(buf.begin(), buf.end(), 0.f);
    for (const DecAlpha& t : xt)
      for (int c = 0; c < cn; c++) buf[t.di * cn + c] = buf[t.di * cn + c] + S[t.si * cn + c] * t.alpha;
    if (dy != prev_dy) {
      uint8_t* D = dst + (size_t)prev_dy * dw * cn;
      for (int i = 0; i < dw * cn; i++) { D[i] = sat_u8_f(sum[i]); sum[i] = beta * buf[i]; }
      prev_dy = dy;
    } else {
      for (int i = 0; i < dw * cn; i++) sum[i] += beta * buf[i];
    }
  }
  uint8_t* D = dst + (size_t)prev_dy * dw * cn;
  for (int i = 0; i < dw * cn; i++) D[i] = sat_u8_f(sum[i]);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* resize(INTER_LINEAR_EXACT), 8-bit single channel: 8.8 fixed-point coefficients both axes, horizontal
 * pass kept in 8.8, vertical pass in 16.16, final (v + 32768) >> 16.  ORB builds level l from level l-1. */
namespace {
struct LinTab { std::vector<int> ofs; std::vector<uint16_t> c0, c1; int lo, hi; };
void lin_tab(int ssize, int dsize, LinTab& t) {
  double inv_scale = (double)dsize / ssize;
  double scale = 1.0 / inv_scale;
  t.ofs.assign(dsize, 0); t.c0.assign(dsize, 0); t.c1.assign(dsize, 0);
  t.lo = 0; t.hi = dsize;
  for (int v = 0; v < dsize; v++) {
    double fval = scale * ((double)v + 0.5) - 0.5;
    int ival = floor_d(fval);
    if (ival >= 0 && ssize > 1) {
      if (ival < ssize - 1) {
        t.ofs[v] = ival;
        int c1 = round_d((fval - (double)ival) * 256.0);
        t.c1[v] = (uint16_t)c1; t.c0[v] = (uint16_t)(256 - c1);
      } else { t.ofs[v] = ssize - 1; t.hi = std::min(t.hi, v); }
    } else t.lo = std::max(t.lo, v + 1);
  }
}
}  // namespace

extern "C" void evo_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  LinTab xt, yt;
  lin_tab(sw, dw, xt);
  lin_tab(sh, dh, yt);
  auto hline = [&](int sy, std::vector<uint16_t>& out) {
    const uint8_t* S = src + (size_t)sy * sw;
    int i = 0;
    for (; i < xt.lo; i++) out[i] = (uint16_t)(S[0] << 8);
    for (; i < xt.hi; i++) out[i] = (uint16_t)(xt.c0[i] * S[xt.ofs[i]] + xt.c1[i] * S[xt.ofs[i] + 1]);
    for (; i < dw; i++) out[i] = (uint16_t)(S[sw - 1] << 8);
  };
  std::vector<uint16_t> r0(dw), r1(dw);
  for (int dy = 0; dy < dh; dy++) {
    uint8_t* D = dst + (size_t)dy * dw;
    if (dy < yt.lo || dy >= yt.hi) {
      hline(dy < yt.lo ? 0 : sh - 1, r0);
      for (int i = 0; i < dw; i++) D[i] = (uint8_t)((r0[i] + 128) >> 8);
    } else {
      hline(yt.ofs[dy], r0);
      hline(yt.ofs[dy] + 1, r1);
      uint32_t m0 = yt.c0[dy], m1 = yt.c1[dy];
      for (int i = 0; i < dw; i++) D[i] = (uint8_t)((r0[i] * m0 + r1[i] * m1 + 32768u) >> 16);
    }
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* ORB_create() defaults: nlevels 8, scaleFactor 1.2f, edgeThreshold 31, patchSize 31, fastThreshold 20 */
extern "C" void evo_orb_layout(int w, int h, int nfeatures, int* lw, int* lh, float* lscale, int* lquota) {
  const int nlevels = 8;
  double scaleFactor = (double)1.2f;
  for (int l = 0; l < nlevels; l++) {
    float s = (float)std::pow(scaleFactor, (double)l);
    lscale[l] = s;
    lw[l] = round_f(w / s);
    lh[l] = round_f(h / s);
  }
  float factor = (float)(1.0 / scaleFactor);
  float ndes = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    lquota[l] = round_f(ndes);
    sum += lquota[l];
    ndes *= factor;
  }
  lquota[nlevels - 1] = std::max(nfeatures - sum, 0);
}

extern "C" int64_t evo_orb_pyramid(const uint8_t* gray, int w, int h, uint8_t* out) {
  int lw[8], lh[8], q[8]; float ls[8];
  evo_orb_layout(w, h, 500, lw, lh, ls, q);
  int64_t off = 0;
  const uint8_t* prev = gray; int pw = w, ph = h;
  for (int l = 0; l < 8; l++) {
    uint8_t* cur = out + off;
    if (l == 0) memcpy(cur, gray, (size_t)w * h);
    else evo_resize_linear_exact(prev, pw, ph, cur, lw[l], lh[l]);
    prev = cur; pw = lw[l]; ph = lh[l];
    off += (int64_t)lw[l] * lh[l];
  }
  return off;
}

/* ------------------------------------------------------------------------------------------------ */
/* FAST-9/16 (threshold strict), corner score = largest threshold for which the pixel stays a corner,
 * 3x3 non-max suppression with strict '>' against all 8 neighbours.                                  */
namespace {
const int kRing[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

inline bool fast_is_corner(const uint8_t* p, int stride, int thr) {
  int v = p[0];
  int bright = 0, dark = 0;  // 16-bit ring masks
  for (int k = 0; k < 16; k++) {
    int r = p[kRing[k][1] * stride + kRing[k][0]];
    if (r > v + thr) bright |= 1 << k;
    if (r < v - thr) dark |= 1 << k;
  }
  for (int pass = 0; pass < 2; pass++) {
    unsigned m = pass ? dark : bright;
    m |= m << 16;
    for (int s = 0; s < 16; s++)
      if (((m >> s) & 0x1FF) == 0x1FF) return true;
  }
  return false;
}

inline int fast_score(const uint8_t* p, int stride, int thr) {
  int d[25];
  int v = p[0];
  for (int k = 0; k < 25; k++) d[k] = v - p[kRing[k & 15][1] * stride + kRing[k & 15][0]];
  int a0 = thr;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min(d[k + 1], d[k + 2]);
    for (int j = 3; j <= 8; j++) a = std::min(a, d[k + j]);
    a0 = std::max(a0, std::min(a, d[k]));
    a0 = std::max(a0, std::min(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max(d[k + 1], d[k + 2]);
    for (int j = 3; j <= 8; j++) b = std::max(b, d[k + j]);
    b0 = std::min(b0, std::max(b, d[k]));
    b0 = std::min(b0, std::max(b, d[k + 9]));
  }
  return -b0 - 1;
}

struct Corner { int x, y, score; };

void fast_nms(const uint8_t* img, int w, int h, int thr, std::vector<Corner>& out) {
  out.clear();
  if (w < 7 || h < 7) return;
  std::vector<uint8_t> sc((size_t)w * h, 0);  // scores of corners, 0 elsewhere (score <= 255)
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      const uint8_t* p = img + (size_t)y * w + x;
      if (fast_is_corner(p, w, thr)) sc[(size_t)y * w + x] = (uint8_t)fast_score(p, w, thr);
    }
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      int s = sc[(size_t)y * w + x];
      if (!s) continue;
      const uint8_t* c = &sc[(size_t)y * w + x];
      if (s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] && s > c[w - 1] && s > c[w] &&
          s > c[w + 1])
        out.push_back({x, y, s});
    }
}

/* ---- KeyPointsFilter::retainBest (features2d/src/keypoint.cpp), including the ORDER it leaves the survivors in ----
 * OpenCV:  std::nth_element(begin, begin + nth, end, response-greater); amb = kp[n-1].response;
 *          new_end = std::partition(begin + n, end, response >= amb); resize.
 * Both calls permute the vector, and everything downstream (match order -> RANSAC sample order) depends on that permutation,
 * so the oracle runs the same two algorithms of libstdc++ on the same sequence.  nth_element is libstdc++'s introselect;
 * two generations of its pivot rule exist (GCC < 4.8.2 / 4.9: median of first, mid, last-1 moved to first; later: median of
 * first+1, mid, last-1), and two versions of OpenCV's call (nth = n in 3.x, n - 1 after the 2018 fix).  g_order_mode:
 *   0  set semantics, row-major order kept (rounds 1-3 of this repository)
 *   1  nth = n,     current libstdc++        2  nth = n - 1, current libstdc++
 *   3  nth = n,     old pivot rule           4  nth = n - 1, old pivot rule
 * Mode 1 is the one the reference's own run agrees with (tests/test_capture_golden.py, profiles/r04_golden_pinning.txt: 120 of
 * 120 matrices of dict_with_homography_matrix.json equal to the last digit, against 0 in mode 0, 66 in mode 2, 1 and 1 in modes 3
 * and 4) and is the default;
 * evo_set_orb_order() / the environment variable EVO_ORB_ORDER select another. */
int g_order_mode = -1;
long g_depth_hits = 0;   // nth_element calls that reached introselect's depth limit (heap-select fall-back)
int order_mode() {
  if (g_order_mode < 0) { const char* e = getenv("EVO_ORB_ORDER"); g_order_mode = e ? atoi(e) : 1; }
  return g_order_mode;
}
template <class It, class C> void old_move_median_first(It a, It b, It c, C comp) {
  if (comp(*a, *b)) { if (comp(*b, *c)) std::iter_swap(a, b); else if (comp(*a, *c)) std::iter_swap(a, c); }
  else if (comp(*a, *c)) return;
  else if (comp(*b, *c)) std::iter_swap(a, c);
  else std::iter_swap(a, b);
}
template <class It, class C> void new_move_median_to_first(It r, It a, It b, It c, C comp) {
  if (comp(*a, *b)) { if (comp(*b, *c)) std::iter_swap(r, b); else if (comp(*a, *c)) std::iter_swap(r, c); else std::iter_swap(r, a); }
  else if (comp(*a, *c)) std::iter_swap(r, a);
  else if (comp(*b, *c)) std::iter_swap(r, c);
  else std::iter_swap(r, b);
}
template <class It, class C> It unguarded_partition(It first, It last, It pivot, C comp) {
  for (;;) {
    while (comp(*first, *pivot)) ++first;
    --last;
    while (comp(*pivot, *last)) --last;
    if (!(first < last)) return first;
    std::iter_swap(first, last);
    ++first;
  }
}
template <class It, class C> void insertion_sort(It first, It last, C comp) {
  if (first == last) return;
  for (It i = first + 1; i != last; ++i) {
    auto val = *i;
    if (comp(val, *first)) { std::move_backward(first, i, i + 1); *first = val; }
    else { It l = i, nx = i; --nx; while (comp(val, *nx)) { *l = *nx; l = nx; --nx; } *l = val; }
  }
}
/* libstdc++ __introselect; old_rule selects the pre-4.9 pivot.  The depth-limit fall-back (heap select) is reached only on
 * adversarial input; it is reported instead of emulated. */
template <class It, class C> bool introselect(It first, It nth, It last, C comp, bool old_rule) {
  if (first == last || nth == last) return true;
  long n = last - first; int lg = 0; while (n > 1) { n >>= 1; ++lg; }
  int depth = 2 * lg;
  while (last - first > 3) {
    if (depth == 0) return false;
    --depth;
    It mid = first + (last - first) / 2;
    if (old_rule) old_move_median_first(first, mid, last - 1, comp);
    else new_move_median_to_first(first, first + 1, mid, last - 1, comp);
    It cut = unguarded_partition(first + 1, last, first, comp);
    if (cut <= nth) first = cut; else last = cut;
  }
  insertion_sort(first, last, comp);
  return true;
}
template <class T, class F>
void retain_best(std::vector<T>& v, int n, F resp) {
  if (n < 0 || (int)v.size() <= n) return;
  if (n == 0) { v.clear(); return; }
  const int mode = order_mode();
  if (mode == 0) {
    std::vector<float> r(v.size());
    for (size_t i = 0; i < v.size(); i++) r[i] = resp(v[i]);
    std::vector<float> s = r;
    std::nth_element(s.begin(), s.begin() + (n - 1), s.end(), std::greater<float>());
    float cut = s[n - 1];
    std::vector<T> keep;
    for (size_t i = 0; i < v.size(); i++)
      if (r[i] >= cut) keep.push_back(v[i]);  // stable: row-major order survives
    v.swap(keep);
    return;
  }
  auto greater = [&](const T& a, const T& b) { return resp(a) > resp(b); };
  const int nth = (mode == 1 || mode == 3) ? n : n - 1;
  std::vector<T> backup = v;
  if (!introselect(v.begin(), v.begin() + nth, v.end(), greater, mode >= 3)) {
    ++g_depth_hits;
    if (mode >= 3) fprintf(stderr, "evz_orb: introselect depth limit reached (old-rule heap fall-back not emulated)\n");
    v = backup;
    std::nth_element(v.begin(), v.begin() + nth, v.end(), greater);
  }
  const float amb = resp(v[n - 1]);
  auto ne = std::partition(v.begin() + n, v.end(), [&](const T& k) { return resp(k) >= amb; });
  v.resize(ne - v.begin());
}

void level_candidates(const uint8_t* img, int w, int h, int quota, std::vector<Corner>& c) {
  fast_nms(img, w, h, 20, c);
  const int border = 31;
  if (h <= border * 2 || w <= border * 2) { c.clear(); return; }
  std::vector<Corner> in;
  for (const Corner& k : c)
    if (k.x >= border && k.x < w - border && k.y >= border && k.y < h - border) in.push_back(k);
  c.swap(in);
  retain_best(c, 2 * quota, [](const Corner& k) { return (float)k.score; });
}

float harris_response(const uint8_t* img, int w, int x0, int y0) {
  const int blockSize = 7, r = blockSize / 2;
  float scale = 1.f / ((1 << 2) * blockSize * 255.f);
  float scale_sq_sq = scale * scale * scale * scale;
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < blockSize; i++)
    for (int j = 0; j < blockSize; j++) {
      const uint8_t* p = img + (size_t)(y0 - r + i) * w + (x0 - r + j);
      int Ix = (p[1] - p[-1]) * 2 + (p[-w + 1] - p[-w - 1]) + (p[w + 1] - p[w - 1]);
      int Iy = (p[w] - p[-w]) * 2 + (p[w - 1] - p[-w - 1]) + (p[w + 1] - p[-w + 1]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
  float fa = (float)a, fb = (float)b, fc = (float)c;
  float t1 = fa * fb;
  float t2 = fc * fc;
  float s = fa + fb;
  float t3 = (0.04f * s) * s;
  return ((t1 - t2) - t3) * scale_sq_sq;
}

struct Umax { int u[17]; };
Umax make_umax() {
  Umax m; const int half = 15;
  int vmax = floor_d(half * std::sqrt(2.f) / 2 + 1);
  int vmin = ceil_d(half * std::sqrt(2.f) / 2);
  for (int v = 0; v <= vmax; v++) m.u[v] = round_d(std::sqrt((double)half * half - v * v));
  for (int v = half, v0 = 0; v >= vmin; --v) {
    while (m.u[v0] == m.u[v0 + 1]) ++v0;
    m.u[v] = v0; ++v0;
  }
  return m;
}

float ic_angle(const uint8_t* img, int w, int x, int y, const Umax& um) {
  const int half = 15;
  const uint8_t* c = img + (size_t)y * w + x;
  int m01 = 0, m10 = 0;
  for (int u = -half; u <= half; ++u) m10 += u * c[u];
  for (int v = 1; v <= half; ++v) {
    int vs = 0, d = um.u[v];
    for (int u = -d; u <= d; ++u) {
      int vp = c[u + v * w], vm = c[u - v * w];
      vs += (vp - vm);
      m10 += u * (vp + vm);
    }
    m01 += v * vs;
  }
  return evo_fast_atan2((float)m01, (float)m10);
}

}  // namespace

extern "C" float evo_fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* Deterministic sin/cos for x in [0, 2*pi]: two-step Cody-Waite reduction by pi/2 and the classic
 * degree-13/14 minimax kernels, evaluated with plain multiplies and adds in a fixed order so that the
 * GPU build can reproduce every bit.  OpenCV evaluates (float)cos(angle), (float)sin(angle) through libm;
 * tests check that the float-rounded results agree.                                                    */
extern "C" void evo_sincos(double x, double* so, double* co) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  const double pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
  double fn = std::nearbyint(x * two_over_pi);
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

extern "C" void evo_set_orb_order(int mode) { g_order_mode = (mode >= 0 && mode <= 4) ? mode : 1; }
extern "C" int evo_get_orb_order(void) { return order_mode(); }
extern "C" long evo_orb_depth_limit_hits(void) { return g_depth_hits; }

extern "C" int evo_fast_nms(const uint8_t* img, int w, int h, int thr, int* xs, int* ys, int* scores, int cap) {
  std::vector<Corner> c;
  fast_nms(img, w, h, thr, c);
  for (int i = 0; i < (int)c.size() && i < cap; i++) { xs[i] = c[i].x; ys[i] = c[i].y; scores[i] = c[i].score; }
  return (int)c.size();
}

/* the map behind fast_nms: FAST-9/16 corner score at `thr` (0 where the pixel is not a corner); for the independent check of
 * the corner predicate against skimage.feature.corner_fast (tests/golden/skimage_fast9.npz) */
extern "C" void evo_fast_score_map(const uint8_t* img, int w, int h, int thr, uint8_t* out) {
  memset(out, 0, (size_t)w * h);
  if (w < 7 || h < 7) return;
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      const uint8_t* p = img + (size_t)y * w + x;
      if (fast_is_corner(p, w, thr)) out[(size_t)y * w + x] = (uint8_t)fast_score(p, w, thr);
    }
}

extern "C" int evo_orb_level_candidates(const uint8_t* img, int w, int h, int quota, int* xs, int* ys, int* scores,
                                        int cap) {
  std::vector<Corner> c;
  level_candidates(img, w, h, quota, c);
  for (int i = 0; i < (int)c.size() && i < cap; i++) { xs[i] = c[i].x; ys[i] = c[i].y; scores[i] = c[i].score; }
  return (int)c.size();
}

/* GaussianBlur(7x7, sigma 2) on 8-bit data: separable, coefficients quantised to 8 fractional bits per pass
 * (cvRound(k*256)), 32-bit sums, one rounding at the end: (v + 32768) >> 16; BORDER_REFLECT_101.        */
namespace {
void gauss_kernel7(int* k) {
  float cf[7]; double sum = 0;
  double scale2X = -0.5 / (2.0 * 2.0);
  for (int i = 0; i < 7; i++) { double x = i - 3; cf[i] = (float)std::exp(scale2X * x * x); sum += cf[i]; }
  sum = 1. / sum;
  for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); k[i] = round_f(cf[i] * 256.f); }
}
}  // namespace

extern "C" void evo_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
  int k[7]; gauss_kernel7(k);
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = -3; i <= 3; i++) s += k[i + 3] * src[(size_t)y * w + reflect101(x + i, w)];
      tmp[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int j = -3; j <= 3; j++) s += k[j + 3] * tmp[(size_t)reflect101(y + j, h) * w + x];
      dst[(size_t)y * w + x] = (uint8_t)((s + 32768) >> 16);
    }
}

extern "C" int evo_orb_detect(const uint8_t* gray, int w, int h, int nfeatures, float* xy, uint8_t* desc,
                              int* octave, int* lx, int* ly, float* response, float* angle, int cap) {
  int lw[8], lh[8], quota[8]; float ls[8];
  evo_orb_layout(w, h, nfeatures, lw, lh, ls, quota);
  std::vector<uint8_t> pyr;
  {
    int64_t tot = 0;
    for (int l = 0; l < 8; l++) tot += (int64_t)lw[l] * lh[l];
    pyr.resize((size_t)tot);
    const uint8_t* prev = gray; int pw = w, ph = h; int64_t off = 0;
    for (int l = 0; l < 8; l++) {
      uint8_t* cur = pyr.data() + off;
      if (l == 0) memcpy(cur, gray, (size_t)w * h);
      else evo_resize_linear_exact(prev, pw, ph, cur, lw[l], lh[l]);
      prev = cur; pw = lw[l]; ph = lh[l];
      off += (int64_t)lw[l] * lh[l];
    }
  }
  static const Umax um = make_umax();
  struct Kp { int x, y, score; float resp, ang; };
  int n = 0;
  int64_t off = 0;
  for (int l = 0; l < 8; l++) {
    const uint8_t* img = pyr.data() + off;
    int W = lw[l], H = lh[l];
    off += (int64_t)W * H;
    std::vector<Corner> c;
    level_candidates(img, W, H, quota[l], c);
    std::vector<Kp> k;
    for (const Corner& q : c) k.push_back({q.x, q.y, q.score, harris_response(img, W, q.x, q.y), 0.f});
    retain_best(k, quota[l], [](const Kp& q) { return q.resp; });
    if (k.empty()) continue;
    for (Kp& q : k) q.ang = ic_angle(img, W, q.x, q.y, um);
    std::vector<uint8_t> blur((size_t)W * H);
    evo_gaussian_blur7(img, W, H, blur.data());
    float s = ls[l];
    for (const Kp& q : k) {
      if (n >= cap) return n;
      float px = (float)q.x * s, py = (float)q.y * s;  // keypoint.pt *= layerScale
      xy[2 * n] = px; xy[2 * n + 1] = py;
      octave[n] = l; lx[n] = q.x; ly[n] = q.y; response[n] = q.resp; angle[n] = q.ang;
      // descriptor (computeOrbDescriptors): centre recovered from pt, taps steered by the keypoint angle
      float inv = 1.f / s;
      int cx = round_f(px * inv), cy = round_f(py * inv);
      float ang = q.ang * (float)(M_PI / 180.f);
      double sd, cd;
      evo_sincos((double)ang, &sd, &cd);
      float a = (float)cd, b = (float)sd;
      const uint8_t* center = blur.data() + (size_t)cy * W + cx;
      uint8_t* d = desc + (size_t)n * 32;
      for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int bit = 0; bit < 8; bit++) {
          const int* p = kPattern + (i * 16 + bit * 2) * 2;  // two points: (p[0],p[1]) and (p[2],p[3])
          float x0 = p[0] * a - p[1] * b, y0 = p[0] * b + p[1] * a;
          float x1 = p[2] * a - p[3] * b, y1 = p[2] * b + p[3] * a;
          int t0 = center[round_f(y0) * W + round_f(x0)];
          int t1 = center[round_f(y1) * W + round_f(x1)];
          val |= (t0 < t1) << bit;
        }
        d[i] = (uint8_t)val;
      }
      n++;
    }
  }
  return n;
}
