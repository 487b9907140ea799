/*
 * oracle/evz_homography.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), match + homography half.
 *
 * Restates
 *   evenvizion/processing/matching.py:102-108   cv2.DescriptorMatcher_create("BruteForce").knnMatch(q, t, 2)
 *   evenvizion/processing/matching.py:166-239   lowes_ratio_test / filter_corresponding_points
 *   evenvizion/processing/utils.py:41-68        remove_double_matching
 *   evenvizion/processing/matching.py:156-157,  cv2.findHomography(a, b, cv2.RANSAC, 3.0)
 *   evenvizion/processing/utils.py:356-358
 *   evenvizion/processing/utils.py:258-325      find_point_displacement / get_largest_group_points
 *   evenvizion/processing/utils.py:328-363      compute_homography
 *   evenvizion/processing/utils.py:118-145      matrix_superposition
 *   evenvizion/processing/video_processing.py:58-107  the stream loop
 * findHomography's arithmetic (calib3d ptsetreg/fundam/levmarq, core Jacobi) is restated from the published
 * OpenCV 3.4 algorithm; pinned jointly with the other operators, since round 4, by the reference's own video and recorded result (tests/test_capture_golden.py: all 120 matrices of dict_with_homography_matrix.json reproduced to the last digit) -- see evz_oracle.h.  The Python-glue functions are pinned by the
 * fixtures in tests/golden/ captured from the reference.
 *
 * Compile with -ffp-contract=off.
 */
#include "evz_oracle.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {
inline int round_d(double v) { return (int)lrint(v); }
}

/* ------------------------------------------------------------------------------------------------ */
/* BFMatcher(NORM_L2).knnMatch(k=2) on uint8 rows: D = sum (q-t)^2 (exact), ascending train scan,
 * insertion only on strictly smaller distance => ties keep the lowest train index.                  */
template <class DistF>
static void knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, uint32_t* d2, DistF dist) {
  for (int i = 0; i < nq; i++) {
    uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu; int i0 = -1, i1 = -1;
    for (int j = 0; j < nt; j++) {
      uint32_t d = dist(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
      else if (d < b1) { b1 = d; i1 = j; }
    }
    idx[2 * i] = i0; idx[2 * i + 1] = i1; d2[2 * i] = b0; d2[2 * i + 1] = b1;
  }
}
extern "C" void evo_knn2_l2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, uint32_t* d2) {
  knn2(q, nq, t, nt, idx, d2, [](const uint8_t* a, const uint8_t* b) {
    uint32_t s = 0;
    for (int k = 0; k < 32; k++) { int d = (int)a[k] - (int)b[k]; s += (uint32_t)(d * d); }
    return s;
  });
}
extern "C" void evo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, uint32_t* d2) {
  knn2(q, nq, t, nt, idx, d2, [](const uint8_t* a, const uint8_t* b) {
    uint32_t s = 0;
    for (int k = 0; k < 32; k++) s += (uint32_t)__builtin_popcount((unsigned)(a[k] ^ b[k]));
    return s;
  });
}

/* lowes_ratio_test (matching.py:186-198) + filter_corresponding_points (matching.py:226-239).
 * DMatch.distance = sqrt((float)D); test "d0 < d1 * ratio" evaluated in double like Python does.
 * Output order = insertion order of train_idx_dict = ascending query index of the surviving claims.  */
extern "C" int evo_ratio_unique(const int32_t* idx, const uint32_t* d2, int nq, double ratio, int32_t* out_q,
                                int32_t* out_t) {
  std::vector<int> sq, st;
  std::unordered_map<int, int> claims;
  for (int i = 0; i < nq; i++) {
    if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) continue;  // len(matches) != 2
    double dist0 = (double)std::sqrt((float)d2[2 * i]);
    double dist1 = (double)std::sqrt((float)d2[2 * i + 1]);
    if (dist0 < dist1 * ratio) { sq.push_back(i); st.push_back(idx[2 * i]); claims[idx[2 * i]]++; }
  }
  int m = 0;
  for (size_t k = 0; k < sq.size(); k++)
    if (claims[st[k]] == 1) { out_q[m] = sq[k]; out_t[m] = st[k]; m++; }
  return m;
}

/* remove_double_matching (utils.py:60-68): dict keyed on (ax, ay): first-seen order, last value wins */
extern "C" int evo_remove_double(const float* a, const float* b, int n, float* oa, float* ob) {
  std::map<std::pair<float, float>, int> pos;
  int m = 0;
  for (int i = 0; i < n; i++) {
    std::pair<float, float> key(a[2 * i], a[2 * i + 1]);
    auto it = pos.find(key);
    int p;
    if (it == pos.end()) { p = m++; pos[key] = p; oa[2 * p] = a[2 * i]; oa[2 * p + 1] = a[2 * i + 1]; }
    else p = it->second;
    ob[2 * p] = b[2 * i]; ob[2 * p + 1] = b[2 * i + 1];
  }
  return m;
}

/* ------------------------------------------------------------------------------------------------ */
/* Symmetric eigen-solver: cyclic-by-largest-pivot Jacobi, eigenvalues sorted descending, eigenvectors
 * as rows of V.  A is n x n row-major (upper triangle is used and destroyed).                       */
namespace {
inline double hyp(double a, double b) {
  a = std::fabs(a); b = std::fabs(b);
  if (a > b) { b /= a; return a * std::sqrt(1 + b * b); }
  if (b > 0) { a /= b; return b * std::sqrt(1 + a * a); }
  return 0;
}
}  // namespace

extern "C" void evo_jacobi(double* A, int n, double* W, double* V) {
  const double eps = DBL_EPSILON;
  int i, j, k, m;
  for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i * n + j] = 0; V[i * n + i] = 1; }
  int iters, maxIters = n * n * 30;
  std::vector<int> indR(n), indC(n);
  double mv = 0;
  for (k = 0; k < n; k++) {
    W[k] = A[(n + 1) * k];
    if (k < n - 1) {
      for (m = k + 1, mv = std::fabs(A[n * k + m]), i = k + 2; i < n; i++) {
        double val = std::fabs(A[n * k + i]);
        if (mv < val) mv = val, m = i;
      }
      indR[k] = m;
    }
    if (k > 0) {
      for (m = 0, mv = std::fabs(A[k]), i = 1; i < k; i++) {
        double val = std::fabs(A[n * i + k]);
        if (mv < val) mv = val, m = i;
      }
      indC[k] = m;
    }
  }
  if (n > 1) for (iters = 0; iters < maxIters; iters++) {
    for (k = 0, mv = std::fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
      double val = std::fabs(A[n * i + indR[i]]);
      if (mv < val) mv = val, k = i;
    }
    int l = indR[k];
    for (i = 1; i < n; i++) {
      double val = std::fabs(A[n * indC[i] + i]);
      if (mv < val) mv = val, k = indC[i], l = i;
    }
    double p = A[n * k + l];
    if (std::fabs(p) <= eps) break;
    double y = (W[l] - W[k]) * 0.5;
    double t = std::fabs(y) + hyp(p, y);
    double s = hyp(p, t);
    double c = t / s;
    s = p / s; t = (p / t) * p;
    if (y < 0) s = -s, t = -t;
    A[n * k + l] = 0;
    W[k] -= t;
    W[l] += t;
    double a0, b0;
#define ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
    for (i = 0; i < k; i++) ROT(A[n * i + k], A[n * i + l]);
    for (i = k + 1; i < l; i++) ROT(A[n * k + i], A[n * i + l]);
    for (i = l + 1; i < n; i++) ROT(A[n * k + i], A[n * l + i]);
    for (i = 0; i < n; i++) ROT(V[n * k + i], V[n * l + i]);
#undef ROT
    for (j = 0; j < 2; j++) {
      int idx = j == 0 ? k : l;
      if (idx < n - 1) {
        for (m = idx + 1, mv = std::fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
          double val = std::fabs(A[n * idx + i]);
          if (mv < val) mv = val, m = i;
        }
        indR[idx] = m;
      }
      if (idx > 0) {
        for (m = 0, mv = std::fabs(A[idx]), i = 1; i < idx; i++) {
          double val = std::fabs(A[n * i + idx]);
          if (mv < val) mv = val, m = i;
        }
        indC[idx] = m;
      }
    }
  }
  for (k = 0; k < n - 1; k++) {
    m = k;
    for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
    if (k != m) {
      std::swap(W[m], W[k]);
      for (i = 0; i < n; i++) std::swap(V[n * m + i], V[n * k + i]);
    }
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* HomographyEstimatorCallback::runKernel -- normalised DLT; src=M, dst=m, H maps M -> m              */
extern "C" int evo_dlt(const float* M, const float* m, int count, double* Hout) {
  double LtL[9][9], W[9], V[9][9];
  double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
  for (int i = 0; i < count; i++) {
    cmx += m[2 * i]; cmy += m[2 * i + 1];
    cMx += M[2 * i]; cMy += M[2 * i + 1];
  }
  cmx /= count; cmy /= count; cMx /= count; cMy /= count;
  for (int i = 0; i < count; i++) {
    smx += std::fabs(m[2 * i] - cmx); smy += std::fabs(m[2 * i + 1] - cmy);
    sMx += std::fabs(M[2 * i] - cMx); sMy += std::fabs(M[2 * i + 1] - cMy);
  }
  if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON ||
      std::fabs(sMy) < DBL_EPSILON)
    return 0;
  smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
  double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
  double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
  memset(LtL, 0, sizeof(LtL));
  for (int i = 0; i < count; i++) {
    double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
    double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
    double Lx[] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
    double Ly[] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
    for (int j = 0; j < 9; j++)
      for (int k = j; k < 9; k++) LtL[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
  }
  for (int j = 0; j < 9; j++)
    for (int k = 0; k < j; k++) LtL[j][k] = LtL[k][j];
  evo_jacobi(&LtL[0][0], 9, W, &V[0][0]);
  const double* H0 = V[8];
  double Ht[9], H1[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      Ht[3 * r + c] = invHnorm[3 * r] * H0[c] + invHnorm[3 * r + 1] * H0[3 + c] + invHnorm[3 * r + 2] * H0[6 + c];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      H1[3 * r + c] = Ht[3 * r] * Hnorm2[c] + Ht[3 * r + 1] * Hnorm2[3 + c] + Ht[3 * r + 2] * Hnorm2[6 + c];
  double inv = 1. / H1[8];
  for (int i = 0; i < 9; i++) Hout[i] = H1[i] * inv;
  return 1;
}

namespace {

struct Rng {  // cv::RNG, multiply-with-carry
  uint64_t state;
  Rng() : state(0xFFFFFFFFFFFFFFFFull) {}
  unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

bool have_collinear(const float* p, int count) {
  int i = count - 1;
  for (int j = 0; j < i; j++) {
    double dx1 = p[2 * j] - p[2 * i];
    double dy1 = p[2 * j + 1] - p[2 * i + 1];
    for (int k = 0; k < j; k++) {
      double dx2 = p[2 * k] - p[2 * i];
      double dy2 = p[2 * k + 1] - p[2 * i + 1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <=
          FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
        return true;
    }
  }
  return false;
}

double det3(const float* p, const int* t) {
  double a00 = p[2 * t[0]], a01 = p[2 * t[0] + 1], a02 = 1., a10 = p[2 * t[1]], a11 = p[2 * t[1] + 1], a12 = 1.,
         a20 = p[2 * t[2]], a21 = p[2 * t[2] + 1], a22 = 1.;
  return a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
}

bool check_subset(const float* ms1, const float* ms2, int count) {
  if (have_collinear(ms1, count) || have_collinear(ms2, count)) return false;
  if (count == 4) {
    static const int tt[][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; i++) negative += det3(ms1, tt[i]) * det3(ms2, tt[i]) < 0;
    if (negative != 0 && negative != 4) return false;
  }
  return true;
}

bool get_subset(const float* m1, const float* m2, int count, float* ms1, float* ms2, Rng& rng, int maxAttempts) {
  int idx[4], i = 0, j, iters = 0;
  for (; iters < maxAttempts; iters++) {
    for (i = 0; i < 4 && iters < maxAttempts;) {
      int idx_i = 0;
      for (;;) {
        idx_i = idx[i] = rng.uniform(0, count);
        for (j = 0; j < i; j++) if (idx_i == idx[j]) break;
        if (j == i) break;
      }
      ms1[2 * i] = m1[2 * idx_i]; ms1[2 * i + 1] = m1[2 * idx_i + 1];
      ms2[2 * i] = m2[2 * idx_i]; ms2[2 * i + 1] = m2[2 * idx_i + 1];
      i++;
    }
    if (i == 4 && !check_subset(ms1, ms2, i)) continue;
    break;
  }
  return i == 4 && iters < maxAttempts;
}

int find_inliers(const float* M, const float* m, int count, const double* H, float t, uint8_t* mask) {
  float Hf[8];
  for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
  int good = 0;
  for (int i = 0; i < count; i++) {
    float Mx = M[2 * i], My = M[2 * i + 1];
    float ww = 1.f / ((Hf[6] * Mx + Hf[7] * My) + 1.f);
    float dx = ((Hf[0] * Mx + Hf[1] * My) + Hf[2]) * ww - m[2 * i];
    float dy = ((Hf[3] * Mx + Hf[4] * My) + Hf[5]) * ww - m[2 * i + 1];
    float err = dx * dx + dy * dy;
    int f = err <= t;
    mask[i] = (uint8_t)f;
    good += f;
  }
  return good;
}

int update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = std::max(p, 0.); p = std::min(p, 1.);
  ep = std::max(ep, 0.); ep = std::min(ep, 1.);
  double num = std::max(1. - p, DBL_MIN);
  double denom = 1. - std::pow(1. - ep, modelPoints);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : round_d(num / denom);
}

/* cv::solve / cv::invert with DECOMP_EIGEN on a symmetric n x n system: Jacobi + back-substitution */
void eig_solve(const double* Ain, int n, const double* b, int nb, double* x) {
  std::vector<double> A(Ain, Ain + n * n), W(n), V(n * n);
  evo_jacobi(A.data(), n, W.data(), V.data());
  double threshold = 0;
  for (int i = 0; i < n; i++) threshold += W[i];
  threshold *= DBL_EPSILON * 2;
  for (int i = 0; i < n * nb; i++) x[i] = 0;
  for (int i = 0; i < n; i++) {
    double wi = W[i];
    if (std::fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    const double* u = &V[i * n];
    if (nb == 1) {
      double s = 0;
      for (int j = 0; j < n; j++) s += u[j] * b[j];
      s *= wi;
      for (int j = 0; j < n; j++) x[j] = x[j] + s * u[j];
    } else {  // b == identity (invert): buffer = wi * u, x += v (x) buffer
      for (int j = 0; j < n; j++) {
        double s = u[j] * wi;
        for (int r = 0; r < n; r++) x[r * nb + j] = x[r * nb + j] + u[r] * s;
      }
    }
  }
}

/* HomographyRefineCallback::compute */
void refine_compute(const float* M, const float* m, int count, const double* h, double* err, double* J) {
  for (int i = 0; i < count; i++) {
    double Mx = M[2 * i], My = M[2 * i + 1];
    double ww = h[6] * Mx + h[7] * My + 1.;
    ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
    double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
    double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
    err[2 * i] = xi - m[2 * i];
    err[2 * i + 1] = yi - m[2 * i + 1];
    if (J) {
      double* Jp = J + (size_t)i * 16;
      Jp[0] = Mx * ww; Jp[1] = My * ww; Jp[2] = ww; Jp[3] = Jp[4] = Jp[5] = 0.;
      Jp[6] = -Mx * ww * xi; Jp[7] = -My * ww * xi;
      Jp[8] = Jp[9] = Jp[10] = 0.; Jp[11] = Mx * ww; Jp[12] = My * ww; Jp[13] = ww;
      Jp[14] = -Mx * ww * yi; Jp[15] = -My * ww * yi;
    }
  }
}

double norm_l2sqr(const double* a, int n) {
  double s = 0; int i = 0;
  for (; i <= n - 4; i += 4) s += a[i] * a[i] + a[i + 1] * a[i + 1] + a[i + 2] * a[i + 2] + a[i + 3] * a[i + 3];
  for (; i < n; i++) s += a[i] * a[i];
  return s;
}
double norm_inf(const double* a, int n) { double s = 0; for (int i = 0; i < n; i++) s = std::max(s, std::fabs(a[i])); return s; }
double dot8(const double* a, const double* b) {
  double r = 0;
  for (int i = 0; i <= 8 - 4; i += 4) r += a[i] * b[i] + a[i + 1] * b[i + 1] + a[i + 2] * b[i + 2] + a[i + 3] * b[i + 3];
  return r;
}
/* A = J^T J (upper, sequential over rows, then mirrored), v = J^T r (four interleaved partial sums) */
void normal_eqs(const double* J, const double* r, int rows, double* A, double* v) {
  for (int i = 0; i < 8; i++)
    for (int j = i; j < 8; j++) {
      double s = 0;
      for (int k = 0; k < rows; k++) s += J[k * 8 + i] * J[k * 8 + j];
      A[i * 8 + j] = s; A[j * 8 + i] = s;
    }
  for (int i = 0; i < 8; i++) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0; int k = 0;
    for (; k <= rows - 4; k += 4) {
      s0 += J[k * 8 + i] * r[k]; s1 += J[(k + 1) * 8 + i] * r[k + 1];
      s2 += J[(k + 2) * 8 + i] * r[k + 2]; s3 += J[(k + 3) * 8 + i] * r[k + 3];
    }
    for (; k < rows; k++) s0 += J[k * 8 + i] * r[k];
    v[i] = (s0 + s1 + s2 + s3) * 1.0;
  }
}

/* createLMSolver(cb, 10)->run(H8): Levenberg-Marquardt as in calib3d/levmarq.cpp */
int lm_refine(const float* M, const float* m, int count, double* H) {
  const int lx = 8, maxIters = 10;
  const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
  std::vector<double> r(2 * count), rd(2 * count), J((size_t)2 * count * 8);
  double x[8], xd[8], A[64], Ap[64], v[8], d[8], temp_d[8], D[8];
  for (int i = 0; i < 8; i++) x[i] = H[i];
  refine_compute(M, m, count, x, r.data(), J.data());
  double S = norm_l2sqr(r.data(), 2 * count);
  normal_eqs(J.data(), r.data(), 2 * count, A, v);
  for (int i = 0; i < 8; i++) D[i] = A[i * 8 + i];
  const double Rlo = 0.25, Rhi = 0.75;
  double lambda = 1, lc = 0.75;
  int iter = 0;
  for (;;) {
    memcpy(Ap, A, sizeof(A));
    for (int i = 0; i < lx; i++) Ap[i * 8 + i] += lambda * D[i];
    eig_solve(Ap, 8, v, 1, d);
    for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
    refine_compute(M, m, count, xd, rd.data(), nullptr);
    double Sd = norm_l2sqr(rd.data(), 2 * count);
    for (int i = 0; i < 8; i++) {  // temp_d = -A*d + 2*v
      const double* a = A + i * 8;
      double s0 = a[0] * d[0] + a[4] * d[4], s1 = a[1] * d[1] + a[5] * d[5], s2 = a[2] * d[2] + a[6] * d[6],
             s3 = a[3] * d[3] + a[7] * d[7];
      temp_d[i] = (s0 + s1 + s2 + s3) * -1.0 + v[i] * 2.0;
    }
    double dS = dot8(d, temp_d);
    double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
    if (R > Rhi) {
      lambda *= 0.5;
      if (lambda < lc) lambda = 0;
    } else if (R < Rlo) {
      double t = dot8(d, v);
      double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
      nu = std::min(std::max(nu, 2.), 10.);
      if (lambda == 0) {
        double I[64];
        eig_solve(A, 8, nullptr, 8, I);
        double maxval = DBL_EPSILON;
        for (int i = 0; i < lx; i++) maxval = std::max(maxval, std::fabs(I[i * 8 + i]));
        lambda = lc = 1. / maxval;
        nu *= 0.5;
      }
      lambda *= nu;
    }
    if (Sd < S) {
      S = Sd;
      for (int i = 0; i < 8; i++) std::swap(x[i], xd[i]);
      refine_compute(M, m, count, x, r.data(), J.data());
      normal_eqs(J.data(), r.data(), 2 * count, A, v);
    }
    iter++;
    bool proceed = iter < maxIters && norm_inf(d, 8) >= epsx && norm_inf(r.data(), 2 * count) >= epsf;
    if (!proceed) break;
  }
  for (int i = 0; i < 8; i++) H[i] = x[i];
  return iter;
}

}  // namespace

/* force_max != 0: the iteration bound is never lowered (RANSACUpdateNumIters is not called), i.e. exactly
 * max(maxItersArg, 1) accepted samples are evaluated -- the fixed-iteration workload of BASELINE configs[2]. */
extern "C" int evo_find_homography_ex(const float* a, const float* b, int n, double thr, int maxItersArg, double conf,
                                      int force_max, double* H, uint8_t* mask, int* info) {
  int it_run = 0, best = 0, lm_it = 0;
  bool result = false;
  if (info) info[0] = info[1] = info[2] = 0;
  for (int i = 0; i < n; i++) mask[i] = 0;
  if (thr <= 0) thr = 3;
  if (n < 4) return 0;
  if (n == 4) {
    if (evo_dlt(a, b, 4, H) <= 0) return 0;
    for (int i = 0; i < 4; i++) mask[i] = 1;
    if (info) info[1] = 4;
    return 1;
  }
  {
    Rng rng;
    int niters = std::max(maxItersArg, 1), maxGood = 0;
    float t = (float)(thr * thr);
    std::vector<uint8_t> cur(n);
    double model[9], bestModel[9];
    float ms1[8], ms2[8];
    for (int iter = 0; iter < niters; iter++) {
      bool found = get_subset(a, b, n, ms1, ms2, rng, 10000);
      if (!found) { if (iter == 0) return 0; break; }
      it_run = iter + 1;
      if (evo_dlt(ms1, ms2, 4, model) <= 0) continue;
      int good = find_inliers(a, b, n, model, t, cur.data());
      if (good > std::max(maxGood, 3)) {
        memcpy(mask, cur.data(), n);
        memcpy(bestModel, model, sizeof(model));
        maxGood = good;
        if (!force_max) niters = update_num_iters(conf, (double)(n - good) / n, 4, niters);
      }
    }
    best = maxGood;
    if (maxGood > 0) { memcpy(H, bestModel, sizeof(bestModel)); result = true; }
  }
  if (info) { info[0] = it_run; info[1] = best; }
  if (!result) { for (int i = 0; i < n; i++) mask[i] = 0; return 0; }
  // refit on the inliers + Levenberg-Marquardt (findHomography, fundam.cpp)
  std::vector<float> sa, sb;
  for (int i = 0; i < n; i++)
    if (mask[i]) { sa.push_back(a[2 * i]); sa.push_back(a[2 * i + 1]); sb.push_back(b[2 * i]); sb.push_back(b[2 * i + 1]); }
  int ni = (int)sa.size() / 2;
  if (ni > 0) {
    evo_dlt(sa.data(), sb.data(), ni, H);  // return value ignored, as in the reference operator
    lm_it = lm_refine(sa.data(), sb.data(), ni, H);
  }
  if (info) info[2] = lm_it;
  return 1;
}

extern "C" int evo_find_homography(const float* a, const float* b, int n, double thr, int maxItersArg, double conf,
                                   double* H, uint8_t* mask, int* info) {
  return evo_find_homography_ex(a, b, n, thr, maxItersArg, conf, 0, H, mask, info);
}

/* np.dot(H, (x, y, 1)): the summation order of numpy's BLAS matrix-vector kernel as captured in
 * tests/golden/glue_goldens.json ("hv"): fma(h0, x, h1*y) + h2.                                     */
static inline void hdot(const double* H, double x, double y, double* tx, double* ty, double* tw) {
  *tx = std::fma(H[0], x, H[1] * y) + H[2];
  *ty = std::fma(H[3], x, H[4] * y) + H[5];
  *tw = std::fma(H[6], x, H[7] * y) + H[8];
}

/* ------------------------------------------------------------------------------------------------ */
/* find_point_displacement (utils.py:316-324) + get_largest_group_points (utils.py:279-285)          */
extern "C" int evo_static_filter(const double* H, const float* a, const float* b, int n, float* oa, float* ob) {
  std::vector<long> r(n);
  std::vector<long> order;            // insertion order of the displacement keys
  std::unordered_map<long, int> cnt;
  for (int i = 0; i < n; i++) {
    double x = a[2 * i], y = a[2 * i + 1];
    double tx, ty, tw;
    hdot(H, x, y, &tx, &ty, &tw);
    double dx = tx / tw - (double)b[2 * i], dy = ty / tw - (double)b[2 * i + 1];
    double dist = std::sqrt(dx * dx + dy * dy);
    long k = lrint(dist);  // Python round(): half to even
    r[i] = k;
    if (!cnt.count(k)) { cnt[k] = 0; order.push_back(k); }
    cnt[k]++;
  }
  if (n == 0) return 0;
  long best = order[0]; int bl = 0;
  for (long k : order) if (cnt[k] > bl) { bl = cnt[k]; best = k; }
  int m = 0;
  for (int i = 0; i < n; i++)
    if (r[i] == best) { oa[2 * m] = a[2 * i]; oa[2 * m + 1] = a[2 * i + 1]; ob[2 * m] = b[2 * i]; ob[2 * m + 1] = b[2 * i + 1]; m++; }
  return m;
}

/* homography_transformation (utils.py:89-92) on a float32 point -> float64 pair */
static inline void htransform(const double* H, double x, double y, double* ox, double* oy) {
  double nx, ny, nw;
  hdot(H, x, y, &nx, &ny, &nw);
  *ox = nx / nw; *oy = ny / nw;
}

/* compute_homography (utils.py:351-362) */
extern "C" int evo_compute_homography_ex(const float* a, const float* b, int n, const double* Hsup, int force_max,
                                         double* H) {
  std::vector<float> fa(a, a + 2 * n), fb(b, b + 2 * n);
  if (Hsup) {
    for (int i = 0; i < n; i++) {
      double x, y;
      htransform(Hsup, a[2 * i], a[2 * i + 1], &x, &y); fa[2 * i] = (float)x; fa[2 * i + 1] = (float)y;
      htransform(Hsup, b[2 * i], b[2 * i + 1], &x, &y); fb[2 * i] = (float)x; fb[2 * i + 1] = (float)y;
    }
  }
  std::vector<uint8_t> mask(n > 0 ? n : 1);
  int found = evo_find_homography_ex(fa.data(), fb.data(), n, 3.0, 2000, 0.995, force_max, H, mask.data(), nullptr);
  int s = 0;
  for (int i = 0; i < n; i++) s += mask[i];
  if ((double)s < 0.7 * (double)n) return EVO_LOW_INLIER_RATIO;
  if (!found) return EVO_NO_FINAL_H;
  return EVO_OK;
}

extern "C" int evo_compute_homography(const float* a, const float* b, int n, const double* Hsup, double* H) {
  return evo_compute_homography_ex(a, b, n, Hsup, 0, H);
}

/* matrix_superposition (utils.py:139-145) */
extern "C" void evo_matrix_superposition(const double* H, const double* Hsup, int first, double* out) {
  if (first) { memcpy(out, H, 9 * sizeof(double)); return; }
  double P[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)  // np.dot(3x3, 3x3): forward FMA chain (pinned by glue_goldens.json "sup_false")
      P[3 * r + c] = std::fma(H[3 * r + 2], Hsup[6 + c], std::fma(H[3 * r + 1], Hsup[3 + c], H[3 * r] * Hsup[c]));
  for (int i = 0; i < 9; i++) out[i] = P[i] / P[8];
}

/* KeyPoints.match_static_kps (matching.py:131-163); a = self (current frame), b = acceding (previous) */
extern "C" int evo_match_static_ex(const float* xy_a, const uint8_t* desc_a, int na, const float* xy_b,
                                   const uint8_t* desc_b, int nb, int force_max, float* oa, float* ob, int* out_n) {
  *out_n = 0;
  if (na == 0 || nb == 0) return EVO_NO_DESCRIPTORS;
  std::vector<int32_t> idx(2 * na), mq(na), mt(na);
  std::vector<uint32_t> d2(2 * na);
  evo_knn2_l2(desc_a, na, desc_b, nb, idx.data(), d2.data());
  int m = evo_ratio_unique(idx.data(), d2.data(), na, 0.5, mq.data(), mt.data());
  if (m < 4) return EVO_FEW_MATCHES;
  std::vector<float> pa(2 * m), pb(2 * m), ua(2 * m), ub(2 * m);
  for (int i = 0; i < m; i++) {
    pa[2 * i] = xy_a[2 * mq[i]]; pa[2 * i + 1] = xy_a[2 * mq[i] + 1];
    pb[2 * i] = xy_b[2 * mt[i]]; pb[2 * i + 1] = xy_b[2 * mt[i] + 1];
  }
  int u = evo_remove_double(pa.data(), pb.data(), m, ua.data(), ub.data());
  double H[9];
  std::vector<uint8_t> mask(u);
  if (!evo_find_homography_ex(ua.data(), ub.data(), u, 3.0, 2000, 0.995, force_max, H, mask.data(), nullptr))
    return EVO_NO_PROVISIONAL_H;
  *out_n = evo_static_filter(H, ua.data(), ub.data(), u, oa, ob);
  return EVO_OK;
}
extern "C" int evo_match_static(const float* xy_a, const uint8_t* desc_a, int na, const float* xy_b,
                                const uint8_t* desc_b, int nb, float* oa, float* ob, int* out_n) {
  return evo_match_static_ex(xy_a, desc_a, na, xy_b, desc_b, nb, 0, oa, ob, out_n);
}

namespace {
struct Feat { std::vector<float> xy; std::vector<uint8_t> desc; int n = 0; };
void detect(const uint8_t* gray, int w, int h, int nfeatures, Feat& f) {
  int cap = nfeatures * 2 + 4096;
  f.xy.resize(2 * cap); f.desc.resize((size_t)32 * cap);
  std::vector<int> oc(cap), lx(cap), ly(cap); std::vector<float> rs(cap), an(cap);
  f.n = evo_orb_detect(gray, w, h, nfeatures, f.xy.data(), f.desc.data(), oc.data(), lx.data(), ly.data(), rs.data(),
                       an.data(), cap);
}
int pair_from_feats(const Feat& cur, const Feat& prev, const double* Hsup, double* H, int force_max = 0) {
  std::vector<float> oa(2 * std::max(cur.n, 1)), ob(2 * std::max(cur.n, 1)), ua(oa.size()), ub(oa.size());
  int n = 0;
  int st = evo_match_static_ex(cur.xy.data(), cur.desc.data(), cur.n, prev.xy.data(), prev.desc.data(), prev.n,
                               force_max, oa.data(), ob.data(), &n);
  if (st != EVO_OK) return st;
  // frame_processing.py:102-104: remove_double_matching across feature types (single type here)
  int u = evo_remove_double(oa.data(), ob.data(), n, ua.data(), ub.data());
  return evo_compute_homography_ex(ua.data(), ub.data(), u, Hsup, force_max, H);
}
}  // namespace

extern "C" int evo_pair_gray(const uint8_t* cur, const uint8_t* prev, int w, int h, int nfeatures, const double* Hsup,
                             double* H) {
  Feat fc, fp;
  detect(cur, w, h, nfeatures, fc);
  detect(prev, w, h, nfeatures, fp);
  return pair_from_feats(fc, fp, Hsup, H);
}

extern "C" void evo_pairs_gray_batch(const uint8_t* frames, int npairs, int w, int h, int nfeatures, int threads,
                                     double* H, int* status) {
  size_t fs = (size_t)w * h;
  auto work = [&](int t) {
    for (int p = t; p < npairs; p += threads)
      status[p] = evo_pair_gray(frames + (2 * (size_t)p + 1) * fs, frames + 2 * (size_t)p * fs, w, h, nfeatures, nullptr,
                                H + 9 * (size_t)p);
  };
  if (threads <= 1) { threads = 1; work(0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(work, t);
  for (auto& x : th) x.join();
}

extern "C" int evo_stream_gray_ex(const uint8_t* frames, int nframes, int w, int h, int nfeatures, int force_max,
                                  double* H, int* status) {
  size_t fs = (size_t)w * h;
  Feat prev, cur;
  detect(frames, w, h, nfeatures, prev);
  double Hsup[9], Hprev[9];
  bool first = true, have_prev = false;
  for (int k = 1; k < nframes; k++) {
    detect(frames + (size_t)k * fs, w, h, nfeatures, cur);
    double* Hk = H + 9 * (size_t)(k - 1);
    int st = pair_from_feats(cur, prev, first ? nullptr : Hsup, Hk, force_max);
    status[k - 1] = st;
    if (st != EVO_OK) {
      if (!have_prev) return k - 1;  // reference: None.tolist() raises on a failing first pair
      memcpy(Hk, Hprev, sizeof(Hprev));
    }
    double S[9];
    evo_matrix_superposition(Hk, Hsup, first, S);
    memcpy(Hsup, S, sizeof(S));
    memcpy(Hprev, Hk, sizeof(Hprev));
    have_prev = true; first = false;
    std::swap(prev, cur);
  }
  return -1;
}
extern "C" int evo_stream_gray(const uint8_t* frames, int nframes, int w, int h, int nfeatures, double* H, int* status) {
  return evo_stream_gray_ex(frames, nframes, w, h, nfeatures, 0, H, status);
}
