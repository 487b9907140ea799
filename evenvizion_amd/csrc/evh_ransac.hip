// evh_ransac.hip -- RANSAC + DLT + Levenberg-Marquardt homography estimation and the reference's point filters
// around it, for gfx950.  Replaces cv2.findHomography(a, b, cv2.RANSAC, 3.0) (reference call sites
// evenvizion/processing/matching.py:156-157 and utils.py:356-358) and the glue find_point_displacement /
// get_largest_group_points (utils.py:258-325), compute_homography (utils.py:328-363), matrix_superposition
// (utils.py:118-145).
//
// One wavefront (64 lanes) per frame pair.  RANSAC hypotheses are evaluated four at a time -- one 4-point
// hypothesis per 16-lane group: the sample sequence of the fixed-seed multiply-with-carry generator is advanced
// by every lane identically, each group keeps its own quadruple, solves its normalised DLT (9x9 Jacobi
// eigen-solver, f64, cooperative inside the group) and counts its inliers; a sequential replay in sample order
// then applies "strictly more inliers wins" and the adaptive iteration bound exactly like a serial RANSAC.
// All floating-point sums keep a fixed order (-ffp-contract=off); f64 throughout the solves, f32 for the
// reprojection error, as the operator being replaced.
#include "evh_internal.h"
#include "evh_ransac.h"
#include <float.h>
#include <math.h>

namespace {

#define WSYNC() __syncthreads()
#define NL 64          // lanes of the wavefront
#define NG 4           // lane groups of 16: one 4-point hypothesis (or one refit / LM matrix) per group
#define GL 16          // lanes per group
#define NC NG          // LDS matrix columns (one per group)
// upper-triangle index of (i, j), i <= j, of an N x N symmetric matrix (the eigen-solver only touches i <= j)
__device__ __forceinline__ int tri_index(int N, int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }
#define TRI(N, i, j) tri_index((N), (i), (j))

struct RansacLds {
  double A[45 * NC];   // per-group symmetric 9x9 (or 8x8), upper triangle, element-major: A[TRI(i,j)*NC + group]
  double V[81 * NC];   // per-group eigenvectors (rows), element-major
  double W[9 * NC];
  int indR[9 * NC];
  int indC[9 * NC];
  double bestH[9];
  double H[9];         // result of the last single-problem DLT / LM
  double x[8], xd[8], v[8], d[8], D[8], tmpd[8], A8[64], Ap[64], Inv[64];
  double sc[8];        // scalars: S, Sd, ...
  int ib[8];           // ints: proceed flags, counts
};

// Cross-lane moves inside a 16-lane group as DPP register moves (a VALU op) instead of ds_bpermute round trips.
// The four controls pair every lane with: its xor-1 / xor-2 neighbour (quad permutes), the mirrored lane of its
// 8-lane half, the mirrored lane of its 16-lane row -- applied in that order a commutative/associative combine
// leaves every lane of the group with the result over all 16 lanes.
#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int group_sum16(int v) {
  v += dpp_i<DPP_XOR1>(v); v += dpp_i<DPP_XOR2>(v); v += dpp_i<DPP_HALF_MIRROR>(v); v += dpp_i<DPP_ROW_MIRROR>(v);
  return v;
}
// (value, index) -> larger value, smaller index among equal values
template <int CTRL>
__device__ __forceinline__ void argmax_step(double& v, int& i) {
  const double ov = dpp_d<CTRL>(v);
  const int oi = dpp_i<CTRL>(i);
  const bool take = ov > v || (ov == v && oi < i);
  v = take ? ov : v; i = take ? oi : i;
}

__device__ __forceinline__ double hyp(double a, double b) {
  a = fabs(a); b = fabs(b);
  if (a > b) { b /= a; return a * sqrt(1 + b * b); }
  if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
  return 0;
}

// Symmetric eigen-solver (Jacobi with largest-pivot selection, eigenvalues sorted descending, eigenvectors = rows
// of V), one matrix per 16-lane group, up to four groups of a wavefront at once.  Arithmetic and pivot order are
// those of the serial algorithm (the first maximum in the scan order R0..R(N-2), C1..C(N-1)); inside a group the
// pivot search is a 16-lane reduction, the 2N-2 element rotations run one per lane and the four index rescans on
// four lanes; only the c/s/t scalar chain is serial.  Must be called by all 64 lanes; `active` is group-uniform.
template <int N>
__device__ void jacobi_group(RansacLds& S, int lane, bool active) {
  const int g = lane >> 4, gl = lane & 15;
#define A0(i, j) S.A[TRI(N, i, j) * NC + g]
#define V0(i, j) S.V[((i) * N + (j)) * NC + g]
#define W0(i) S.W[(i) * NC + g]
#define IR0(i) S.indR[(i) * NC + g]
#define IC0(i) S.indC[(i) * NC + g]
  const double eps = DBL_EPSILON;
  if (active) {
    for (int e = gl; e < N * N; e += GL) { const int i = e / N, j = e - i * N; V0(i, j) = i == j ? 1.0 : 0.0; }
    if (gl < N) W0(gl) = A0(gl, gl);
    // initial indR (lanes 0..N-2 -> k) and indC (lanes 8.. -> k = 1..N-1): each lane scans its own row / column
    if (gl < N - 1) {
      const int k = gl;
      int m = k + 1; double mv = fabs(A0(k, m));
      for (int i = k + 2; i < N; i++) { double val = fabs(A0(k, i)); if (mv < val) mv = val, m = i; }
      IR0(k) = m;
    } else if (gl >= 8 && gl < 8 + N - 1) {
      const int k = gl - 8 + 1;
      int m = 0; double mv = fabs(A0(0, k));
      for (int i = 1; i < k; i++) { double val = fabs(A0(i, k)); if (mv < val) mv = val, m = i; }
      IC0(k) = m;
    }
  }
  WSYNC();
  bool done = !active;
  const int maxIters = N * N * 30;
  for (int iters = 0; iters < maxIters; iters++) {
    if (__ballot(!done) == 0ull) break;
    // ---- pivot: lane j < N-1 holds |A[j][indR[j]]|, lane 8+(i-1) holds |A[indC[i]][i]| (i = 1..N-1); lane order =
    //      scan order, so "first maximum" = smallest lane among the equal maxima
    double val = -1.0;
    if (!done) {
      if (gl < N - 1) val = fabs(A0(gl, IR0(gl)));
      else if (gl >= 8 && gl < 8 + N - 1) { const int i = gl - 8 + 1; val = fabs(A0(IC0(i), i)); }
    }
    double mx = val;
    int who = gl;              // first maximum in scan order = smallest lane among the equal maxima
    argmax_step<DPP_XOR1>(mx, who); argmax_step<DPP_XOR2>(mx, who);
    argmax_step<DPP_HALF_MIRROR>(mx, who); argmax_step<DPP_ROW_MIRROR>(mx, who);
    int k = 0, l = 1;
    double c = 1, sn = 0, t = 0;
    if (!done) {
      if (who < 8) { k = who; l = IR0(who); }
      else { l = who - 8 + 1; k = IC0(l); }
      const double p = A0(k, l);
      if (fabs(p) <= eps) done = true;
      else {
        const double y = (W0(l) - W0(k)) * 0.5;
        t = fabs(y) + hyp(p, y);
        sn = hyp(p, t);
        c = t / sn;
        sn = p / sn; t = (p / t) * p;
        if (y < 0) sn = -sn, t = -t;
      }
    }
    WSYNC();   // every lane has read A[k][l], W[k], W[l] before they change
    if (!done) {
      if (gl == 0) { A0(k, l) = 0; W0(k) -= t; W0(l) += t; }
      // rotations: lane r < N rotates the V pair of column r; lane N+j rotates the A pair of the j-th index != k, l
      if (gl < N) {
        const double a0 = V0(k, gl), b0 = V0(l, gl);
        V0(k, gl) = a0 * c - b0 * sn; V0(l, gl) = a0 * sn + b0 * c;
      } else if (gl - N < N - 2) {
        int i = gl - N;
        if (i >= k) i++;
        if (i >= l) i++;
        double* p0 = i < k ? &A0(i, k) : &A0(k, i);
        double* p1 = i < l ? &A0(i, l) : &A0(l, i);
        const double a0 = *p0, b0 = *p1;
        *p0 = a0 * c - b0 * sn; *p1 = a0 * sn + b0 * c;
      }
    }
    WSYNC();
    // ---- rescan indR / indC of the two touched indices: four scans, one per quad of the group (quad 0: row k,
    //      1: column k, 2: row l, 3: column l); a lane takes elements sl and sl+4 of its scan, the quad combines with
    //      "larger value, smaller index on ties" = the first maximum of the serial strict-< scan
    {
      const int sc = gl >> 2, sl = gl & 3;
      const int idx = sc < 2 ? k : l;
      const bool rowscan = (sc & 1) == 0;
      const bool want = !done && (rowscan ? idx < N - 1 : idx > 0);
      double mv = -1.0; int m = 0x7FFFFFFF;
      if (want) {
        const int first = rowscan ? idx + 1 : 0, end = rowscan ? N : idx;     // elements [first, end)
        const int i1 = first + sl, i2 = i1 + 4;
        if (i1 < end) { mv = fabs(rowscan ? A0(idx, i1) : A0(i1, idx)); m = i1; }
        if (i2 < end) { const double v2 = fabs(rowscan ? A0(idx, i2) : A0(i2, idx)); if (mv < v2) mv = v2, m = i2; }
      }
      argmax_step<DPP_XOR1>(mv, m); argmax_step<DPP_XOR2>(mv, m);
      if (want && sl == 0) { if (rowscan) IR0(idx) = m; else IC0(idx) = m; }
    }
    WSYNC();
  }
  WSYNC();
  // ---- sort eigenvalues (descending) with their eigenvector rows: selection sort
  for (int k = 0; k < N - 1; k++) {
    int m = k;
    if (active) for (int i = k + 1; i < N; i++) if (W0(m) < W0(i)) m = i;
    WSYNC();
    if (active && k != m) {
      if (gl == 0) { double tw = W0(m); W0(m) = W0(k); W0(k) = tw; }
      if (gl >= 1 && gl <= N) { const int i = gl - 1; double tv = V0(m, i); V0(m, i) = V0(k, i); V0(k, i) = tv; }
    }
    WSYNC();
  }
#undef A0
#undef V0
#undef W0
#undef IR0
#undef IC0
}

// de-normalise the smallest-eigenvalue eigenvector (row 8 of V, column `col`) into H (runKernel's tail)
__device__ __forceinline__ void dlt_finish(RansacLds& S, int col, double cmx, double cmy, double smx, double smy,
                                           double cMx, double cMy, double sMx, double sMy, double* H) {
  double H0[9];
#pragma unroll
  for (int i = 0; i < 9; i++) H0[i] = S.V[((8) * 9 + i) * NC + col];
  const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
  const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
  double Ht[9], H1[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++)
      Ht[3 * r + c] = (invHnorm[3 * r] * H0[c] + invHnorm[3 * r + 1] * H0[3 + c]) + invHnorm[3 * r + 2] * H0[6 + c];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++)
      H1[3 * r + c] = (Ht[3 * r] * Hnorm2[c] + Ht[3 * r + 1] * Hnorm2[3 + c]) + Ht[3 * r + 2] * Hnorm2[6 + c];
  double inv = 1. / H1[8];
#pragma unroll
  for (int i = 0; i < 9; i++) H[i] = H1[i] * inv;
}

// one entry (j, k) of L^T L contributed by a normalised correspondence (x, y) <- (X, Y)
__device__ __forceinline__ double ltl_term(int j, int k, double x, double y, double X, double Y) {
  const double nxX = -x * X, nxY = -x * Y, nyX = -y * X, nyY = -y * Y;
  // Lx = {X, Y, 1, 0, 0, 0, -xX, -xY, -x}; Ly = {0, 0, 0, X, Y, 1, -yX, -yY, -y}
#define LXS(q) ((q) == 0 ? X : (q) == 1 ? Y : (q) == 2 ? 1.0 : (q) < 6 ? 0.0 : (q) == 6 ? nxX : (q) == 7 ? nxY : -x)
#define LYS(q) ((q) < 3 ? 0.0 : (q) == 3 ? X : (q) == 4 ? Y : (q) == 5 ? 1.0 : (q) == 6 ? nyX : (q) == 7 ? nyY : -y)
  return LXS(j) * LXS(k) + LYS(j) * LYS(k);
#undef LXS
#undef LYS
}

// normalised DLT of each group's own 4 correspondences (M -> m): every lane of a group holds the same 4 rows.
// `valid` is group-uniform; returns (group-uniform) whether a model was produced; H valid on every lane of the group.
__device__ bool dlt4_group(RansacLds& S, int lane, bool valid, const float* Mx, const float* My, const float* mx,
                           const float* my, double* H) {
  const int g = lane >> 4, gl = lane & 15;
  double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { cmx += mx[i]; cmy += my[i]; cMx += Mx[i]; cMy += My[i]; }
  cmx /= 4; cmy /= 4; cMx /= 4; cMy /= 4;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    smx += fabs(mx[i] - cmx); smy += fabs(my[i] - cmy);
    sMx += fabs(Mx[i] - cMx); sMy += fabs(My[i] - cMy);
  }
  const bool ok = valid && !(fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON ||
                             fabs(sMy) < DBL_EPSILON);
  if (ok) {
    smx = 4 / smx; smy = 4 / smy; sMx = 4 / sMx; sMy = 4 / sMy;
    for (int e = gl; e < 45; e += GL) {          // L^T L upper triangle, entry e <-> (j, k), rows summed in order
      int j = 0, r = e;
      while (r >= 9 - j) { r -= 9 - j; j++; }
      const int k = j + r;
      double acc = 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
        acc += ltl_term(j, k, (mx[i] - cmx) * smx, (my[i] - cmy) * smy, (Mx[i] - cMx) * sMx, (My[i] - cMy) * sMy);
      S.A[TRI(9, j, k) * NC + g] = acc;
    }
  }
  WSYNC();
  jacobi_group<9>(S, lane, ok);
  if (ok) dlt_finish(S, g, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
  return ok;
}

struct Rng {  // multiply-with-carry generator, seeded with all ones for every findHomography call
  unsigned long long state;
  __device__ Rng() : state(0xFFFFFFFFFFFFFFFFull) {}
  __device__ unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
    return (unsigned)state;
  }
};

__device__ __forceinline__ bool have_collinear4(const float* px, const float* py) {
  const int i = 3;
  for (int j = 0; j < i; j++) {
    double dx1 = px[j] - px[i];
    double dy1 = py[j] - py[i];
    for (int k = 0; k < j; k++) {
      double dx2 = px[k] - px[i];
      double dy2 = py[k] - py[i];
      if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
    }
  }
  return false;
}

__device__ __forceinline__ double det3(const float* px, const float* py, int t0, int t1, int t2) {
  double a00 = px[t0], a01 = py[t0], a02 = 1., a10 = px[t1], a11 = py[t1], a12 = 1., a20 = px[t2], a21 = py[t2],
         a22 = 1.;
  return a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
}

__device__ bool check_subset4(const float* Mx, const float* My, const float* mx, const float* my) {
  if (have_collinear4(Mx, My) || have_collinear4(mx, my)) return false;
  const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
  int negative = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    negative += (det3(Mx, My, tt[i][0], tt[i][1], tt[i][2]) * det3(mx, my, tt[i][0], tt[i][1], tt[i][2]) < 0) ? 1 : 0;
  return negative == 0 || negative == 4;
}

__device__ __forceinline__ bool is_inlier(const float* Hf, float Mx, float My, float mx, float my, float t) {
  float ww = 1.f / ((Hf[6] * Mx + Hf[7] * My) + 1.f);
  float dx = ((Hf[0] * Mx + Hf[1] * My) + Hf[2]) * ww - mx;
  float dy = ((Hf[3] * Mx + Hf[4] * My) + Hf[5]) * ww - my;
  float err = dx * dx + dy * dy;
  return err <= t;
}

__device__ int update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = fmax(p, 0.); p = fmin(p, 1.);
  ep = fmax(ep, 0.); ep = fmin(ep, 1.);
  double num = fmax(1. - p, DBL_MIN);
  double denom = 1. - pow(1. - ep, (double)modelPoints);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)__builtin_rint(num / denom);
}

__device__ __forceinline__ int wave_sum(int v) {
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

// ---- single-problem normalised DLT on `count` rows (ax,ay,bx,by): sums in row order, one lane per sum -------------
__device__ bool dlt_rows(RansacLds& S, int lane, const float* rows, int count, double* Hout /* LDS */) {
  // centroids: lanes 0..3 own cm.x, cm.y, cM.x, cM.y  (m = b columns, M = a columns)
  double acc = 0;
  if (lane < 4) {
    const int col = lane == 0 ? 2 : lane == 1 ? 3 : lane == 2 ? 0 : 1;
    for (int i = 0; i < count; i++) acc += rows[4 * i + col];
    acc /= count;
  }
  const double cmx = __shfl(acc, 0), cmy = __shfl(acc, 1), cMx = __shfl(acc, 2), cMy = __shfl(acc, 3);
  double dev = 0;
  if (lane < 4) {
    const int col = lane == 0 ? 2 : lane == 1 ? 3 : lane == 2 ? 0 : 1;
    const double cc = lane == 0 ? cmx : lane == 1 ? cmy : lane == 2 ? cMx : cMy;
    for (int i = 0; i < count; i++) dev += fabs(rows[4 * i + col] - cc);
  }
  double smx = __shfl(dev, 0), smy = __shfl(dev, 1), sMx = __shfl(dev, 2), sMy = __shfl(dev, 3);
  if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
    return false;
  smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
  // LtL upper triangle: lane e <-> entry (j,k), sequential over rows
  if (lane < 45) {
    int j = 0, e = lane;
    while (e >= 9 - j) { e -= 9 - j; j++; }
    const int k = j + e;
    double s = 0;
    for (int i = 0; i < count; i++) {
      double x = (rows[4 * i + 2] - cmx) * smx, y = (rows[4 * i + 3] - cmy) * smy;
      double X = (rows[4 * i] - cMx) * sMx, Y = (rows[4 * i + 1] - cMy) * sMy;
      double nxX = -x * X, nxY = -x * Y, nyX = -y * X, nyY = -y * Y;
      // Lx = {X, Y, 1, 0, 0, 0, -xX, -xY, -x}; Ly = {0, 0, 0, X, Y, 1, -yX, -yY, -y}
#define LXS(q) ((q) == 0 ? X : (q) == 1 ? Y : (q) == 2 ? 1.0 : (q) < 6 ? 0.0 : (q) == 6 ? nxX : (q) == 7 ? nxY : -x)
#define LYS(q) ((q) < 3 ? 0.0 : (q) == 3 ? X : (q) == 4 ? Y : (q) == 5 ? 1.0 : (q) == 6 ? nyX : (q) == 7 ? nyY : -y)
      s += LXS(j) * LXS(k) + LYS(j) * LYS(k);
#undef LXS
#undef LYS
    }
    S.A[TRI(9, j, k) * NC + 0] = s;  // column 0
  }
  WSYNC();
  jacobi_group<9>(S, lane, lane < GL);
  if (lane == 0) {
    double H[9];
    dlt_finish(S, 0, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
    for (int i = 0; i < 9; i++) Hout[i] = H[i];
  }
  WSYNC();
  return true;
}

// ---- symmetric solve / inverse through the eigen-decomposition (cv::solve / cv::invert, DECOMP_EIGEN) -------------
// all 64 lanes; Ain / b / x live in LDS.  Back-substitution keeps the serial summation orders: lane i forms
// s_i = (sum_j u_i[j] b[j]) / w_i, lane j accumulates x[j] += s_i u_i[j] over i ascending.
__device__ void eig_solve8_wave(RansacLds& S, int lane, const double* Ain /*LDS 64*/, const double* b /*LDS 8 or null*/,
                                double* x /*LDS 8 or 64*/) {
  const int N = 8;
  if (lane < 36) {
    int i = 0, e = lane;
    while (e >= 8 - i) { e -= 8 - i; i++; }
    const int j = i + e;
    S.A[TRI(8, i, j) * NC] = Ain[i * 8 + j];
  }
  WSYNC();
  jacobi_group<8>(S, lane, lane < GL);
  double threshold = 0;
  for (int i = 0; i < 8; i++) threshold += S.W[i * NC];
  threshold *= DBL_EPSILON * 2;
  if (b) {
    // s_i on lane i (0 for skipped eigenvalues is NOT equivalent to skipping: keep a flag)
    double si = 0; bool use = false;
    if (lane < 8) {
      double wi = S.W[lane * NC];
      if (!(fabs(wi) <= threshold)) {
        use = true;
        wi = 1 / wi;
        double acc = 0;
        for (int j = 0; j < 8; j++) acc += S.V[(lane * N + j) * NC] * b[j];
        si = acc * wi;
      }
    }
    double xj = 0;
    for (int i = 0; i < 8; i++) {
      const double s_i = __shfl(si, i);
      const int u_i = __shfl((int)use, i);
      if (u_i && lane < 8) xj = xj + s_i * S.V[(i * N + lane) * NC];
    }
    if (lane < 8) x[lane] = xj;
  } else {
    // inverse: x[r][j] += u_i[r] * (u_i[j] / w_i) over i ascending; lane = r*8 + j
    const int r = lane >> 3, j = lane & 7;
    double acc = 0;
    for (int i = 0; i < 8; i++) {
      double wi = S.W[i * NC];
      if (fabs(wi) <= threshold) continue;
      wi = 1 / wi;
      const double sj = S.V[(i * N + j) * NC] * wi;
      acc = acc + S.V[(i * N + r) * NC] * sj;
    }
    x[r * 8 + j] = acc;
  }
  WSYNC();
}

// per-point residual pieces of the refinement callback: lm[4i..] = {ww, xi, yi}; returns nothing, all lanes help
__device__ void lm_points(const float* rows, int count, const double* h /*LDS*/, double* lm, int lane) {
  const double h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7];
  for (int i = lane; i < count; i += NL) {
    double Mx = rows[4 * i], My = rows[4 * i + 1];
    double ww = h6 * Mx + h7 * My + 1.;
    ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
    double xi = (h0 * Mx + h1 * My + h2) * ww;
    double yi = (h3 * Mx + h4 * My + h5) * ww;
    lm[4 * i] = ww; lm[4 * i + 1] = xi; lm[4 * i + 2] = yi;
  }
}

// sum of squared residuals in the fixed "groups of four rows" order; lane 0
__device__ double lm_norm_l2sqr(const float* rows, const double* lm, int count) {
  double s = 0;
  int i = 0;
  for (; i + 1 < count; i += 2) {
    double v0 = lm[4 * i + 1] - rows[4 * i + 2], v1 = lm[4 * i + 2] - rows[4 * i + 3];
    double v2 = lm[4 * i + 5] - rows[4 * i + 6], v3 = lm[4 * i + 6] - rows[4 * i + 7];
    s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
  }
  if (i < count) {
    double v0 = lm[4 * i + 1] - rows[4 * i + 2], v1 = lm[4 * i + 2] - rows[4 * i + 3];
    s += v0 * v0;
    s += v1 * v1;
  }
  return s;
}

__device__ __forceinline__ double jsel(int q, double a0, double a1, double a2, double a6, double a7) {
  // x-row of the Jacobian: {a0, a1, a2, 0, 0, 0, a6, a7}
  return q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : q < 6 ? 0.0 : q == 6 ? a6 : a7;
}
__device__ __forceinline__ double jsely(int q, double a0, double a1, double a2, double a6, double a7) {
  // y-row: {0, 0, 0, a0, a1, a2, a6, a7}
  return q < 3 ? 0.0 : q == 3 ? a0 : q == 4 ? a1 : q == 5 ? a2 : q == 6 ? a6 : a7;
}

// A = J^T J (lanes 0..35, one upper-triangle entry each, rows in order) and v = J^T r (lanes 36..43, four
// interleaved partial sums); results to LDS A8 (mirrored) and v.
__device__ void lm_normal_eqs(RansacLds& S, const float* rows, const double* lm, int count, int lane) {
  if (lane < 36) {
    int i = 0, e = lane;
    while (e >= 8 - i) { e -= 8 - i; i++; }
    const int j = i + e;
    double s = 0;
    for (int p = 0; p < count; p++) {
      double Mx = rows[4 * p], My = rows[4 * p + 1];
      double ww = lm[4 * p], xi = lm[4 * p + 1], yi = lm[4 * p + 2];
      double a0 = Mx * ww, a1 = My * ww;
      double x6 = -Mx * ww * xi, x7 = -My * ww * xi, y6 = -Mx * ww * yi, y7 = -My * ww * yi;
      s += jsel(i, a0, a1, ww, x6, x7) * jsel(j, a0, a1, ww, x6, x7);
      s += jsely(i, a0, a1, ww, y6, y7) * jsely(j, a0, a1, ww, y6, y7);
    }
    S.A8[i * 8 + j] = s; S.A8[j * 8 + i] = s;
  } else if (lane < 44) {
    const int i = lane - 36;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int p = 0;
    for (; p + 1 < count; p += 2) {
      {
        double Mx = rows[4 * p], My = rows[4 * p + 1];
        double ww = lm[4 * p], xi = lm[4 * p + 1], yi = lm[4 * p + 2];
        double a0 = Mx * ww, a1 = My * ww;
        s0 += jsel(i, a0, a1, ww, -Mx * ww * xi, -My * ww * xi) * (xi - rows[4 * p + 2]);
        s1 += jsely(i, a0, a1, ww, -Mx * ww * yi, -My * ww * yi) * (yi - rows[4 * p + 3]);
      }
      {
        const int q = p + 1;
        double Mx = rows[4 * q], My = rows[4 * q + 1];
        double ww = lm[4 * q], xi = lm[4 * q + 1], yi = lm[4 * q + 2];
        double a0 = Mx * ww, a1 = My * ww;
        s2 += jsel(i, a0, a1, ww, -Mx * ww * xi, -My * ww * xi) * (xi - rows[4 * q + 2]);
        s3 += jsely(i, a0, a1, ww, -Mx * ww * yi, -My * ww * yi) * (yi - rows[4 * q + 3]);
      }
    }
    if (p < count) {
      double Mx = rows[4 * p], My = rows[4 * p + 1];
      double ww = lm[4 * p], xi = lm[4 * p + 1], yi = lm[4 * p + 2];
      double a0 = Mx * ww, a1 = My * ww;
      s0 += jsel(i, a0, a1, ww, -Mx * ww * xi, -My * ww * xi) * (xi - rows[4 * p + 2]);
      s0 += jsely(i, a0, a1, ww, -Mx * ww * yi, -My * ww * yi) * (yi - rows[4 * p + 3]);
    }
    S.v[i] = (s0 + s1 + s2 + s3) * 1.0;
  }
  WSYNC();
}

__device__ __forceinline__ double dot8(const double* a, const double* b) {
  double r = 0;
  r += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  r += a[4] * b[4] + a[5] * b[5] + a[6] * b[6] + a[7] * b[7];
  return r;
}

// Levenberg-Marquardt refinement of S.H[0..7] over `count` rows (<= 10 iterations). Returns iterations.
__device__ int lm_refine(RansacLds& S, int lane, const float* rows, int count, double* lm) {
  const int maxIters = 10;
  const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
  if (lane < 8) S.x[lane] = S.H[lane];
  WSYNC();
  lm_points(rows, count, S.x, lm, lane);
  __threadfence_block();
  WSYNC();
  if (lane == 0) S.sc[0] = lm_norm_l2sqr(rows, lm, count);  // S
  lm_normal_eqs(S, rows, lm, count, lane);
  if (lane < 8) S.D[lane] = S.A8[lane * 8 + lane];
  if (lane == 0) { S.sc[2] = 1; S.sc[3] = 0.75; }  // lambda, lc
  WSYNC();
  int iter = 0;
  for (;;) {
    {
      const int i = lane >> 3, j = lane & 7;                 // Ap = A + lambda * diag(D)
      S.Ap[lane] = i == j ? S.A8[lane] + S.sc[2] * S.D[i] : S.A8[lane];
    }
    WSYNC();
    eig_solve8_wave(S, lane, S.Ap, S.v, S.d);
    if (lane < 8) S.xd[lane] = S.x[lane] - S.d[lane];
    WSYNC();
    lm_points(rows, count, S.xd, lm, lane);
    __threadfence_block();
    WSYNC();
    // trial residual -> Sd, gain ratio R; lane 0 decides, the (rare) inverse is done by the whole wave
    if (lane == 0) {
      const double Rlo = 0.25, Rhi = 0.75;
      double Sc = S.sc[0];
      double Sd = lm_norm_l2sqr(rows, lm, count);
      for (int i = 0; i < 8; i++) {  // tmpd = -A*d + 2*v  (four interleaved partial sums per row)
        const double* a = S.A8 + i * 8;
        const double* d = S.d;
        double s0 = a[0] * d[0] + a[4] * d[4], s1 = a[1] * d[1] + a[5] * d[5], s2 = a[2] * d[2] + a[6] * d[6],
               s3 = a[3] * d[3] + a[7] * d[7];
        S.tmpd[i] = (s0 + s1 + s2 + s3) * -1.0 + S.v[i] * 2.0;
      }
      double dS = dot8(S.d, S.tmpd);
      double R = (Sc - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
      double lambda = S.sc[2], lc = S.sc[3];
      int need_inv = 0;
      double nu = 0;
      if (R > Rhi) {
        lambda *= 0.5;
        if (lambda < lc) lambda = 0;
      } else if (R < Rlo) {
        double t = dot8(S.d, S.v);
        nu = (Sd - Sc) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
        nu = fmin(fmax(nu, 2.), 10.);
        if (lambda == 0) need_inv = 1;
        else lambda *= nu;
      }
      S.sc[2] = lambda; S.sc[3] = lc; S.sc[4] = nu; S.sc[5] = Sd;
      S.ib[1] = need_inv;
    }
    WSYNC();
    if (S.ib[1]) {
      eig_solve8_wave(S, lane, S.A8, nullptr, S.Inv);
      if (lane == 0) {
        double maxval = DBL_EPSILON;
        for (int i = 0; i < 8; i++) maxval = fmax(maxval, fabs(S.Inv[i * 8 + i]));
        const double lam = 1. / maxval;
        S.sc[3] = lam;                       // lc
        S.sc[2] = lam * (S.sc[4] * 0.5);     // lambda = lc; nu *= 0.5; lambda *= nu
      }
      WSYNC();
    }
    if (lane == 0) {
      const double Sc = S.sc[0], Sd = S.sc[5];
      S.ib[0] = Sd < Sc ? 1 : 0;
      if (Sd < Sc) {
        S.sc[0] = Sd;
        for (int i = 0; i < 8; i++) { double t = S.x[i]; S.x[i] = S.xd[i]; S.xd[i] = t; }
      }
    }
    WSYNC();
    const bool accepted = S.ib[0] != 0;
    // residuals / Jacobian at the accepted point (lm currently holds the trial point's pieces = accepted x if taken)
    if (accepted) {
      lm_normal_eqs(S, rows, lm, count, lane);
    } else {
      lm_points(rows, count, S.x, lm, lane);  // restore the pieces of the kept point for the norms below
      __threadfence_block();
      WSYNC();
    }
    iter++;
    // norm(r, INF) of the accepted residual, norm(d, INF)
    double rmax = 0;
    for (int i = lane; i < count; i += NL) {
      rmax = fmax(rmax, fabs(lm[4 * i + 1] - rows[4 * i + 2]));
      rmax = fmax(rmax, fabs(lm[4 * i + 2] - rows[4 * i + 3]));
    }
    for (int s = 32; s > 0; s >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, s));
    double dmax = 0;
    for (int i = 0; i < 8; i++) dmax = fmax(dmax, fabs(S.d[i]));
    const bool proceed = iter < maxIters && dmax >= epsx && rmax >= epsf;
    WSYNC();
    if (!proceed) break;
  }
  if (lane < 8) S.H[lane] = S.x[lane];
  WSYNC();
  return iter;
}

// ---- cv2.findHomography(a, b, RANSAC, thr) on `n` rows; result in S.H (LDS), mask[n] in global. --------------------
// scratch: crow = float rows [n][4] for the compacted inliers, lm = double [n][4].
__device__ bool find_homography_wave(RansacLds& S, int lane, const float* rows, int n, double thr, int maxItersArg,
                                     double conf, int force_max, uint8_t* mask, float* crow, double* lm, int* info) {
  if (info && lane < 3) info[lane] = 0;
  for (int i = lane; i < n; i += NL) mask[i] = 0;
  if (thr <= 0) thr = 3;
  if (n < 4) return false;
  if (n == 4) {
    bool ok = dlt_rows(S, lane, rows, 4, S.H);
    if (!ok) return false;
    if (lane < 4) mask[lane] = 1;
    if (info && lane == 0) info[1] = 4;
    return true;
  }
  const float t = (float)(thr * thr);
  Rng rng;
  int niters = max(maxItersArg, 1), maxGood = 0, iter = 0, run = 0;
  bool stop = false, any_found = false;
  while (!stop && iter < niters) {
    // every lane advances the generator identically through NG quadruples; the lanes of group h keep quadruple #h
    const int grp = lane >> 4;
    int my[4] = {0, 1, 2, 3};
    for (int h = 0; h < NG; h++) {
      int q[4];
      for (int i = 0; i < 4;) {
        int idx_i;
        for (;;) {
          idx_i = q[i] = (int)(rng.next() % (unsigned)n);
          int j = 0;
          for (; j < i; j++) if (idx_i == q[j]) break;
          if (j == i) break;
        }
        i++;
      }
      if (h == grp) { my[0] = q[0]; my[1] = q[1]; my[2] = q[2]; my[3] = q[3]; }
    }
    float Mx[4], My[4], mx[4], my_[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float4 r = *reinterpret_cast<const float4*>(rows + 4 * my[i]);
      Mx[i] = r.x; My[i] = r.y; mx[i] = r.z; my_[i] = r.w;
    }
    const bool valid = check_subset4(Mx, My, mx, my_);
    double H[9];
    const bool ok = dlt4_group(S, lane, valid, Mx, My, mx, my_, H);
    int good = 0;
    if (ok) {   // the 16 lanes of the group split the points; integer count, order-free
      float Hf[8];
#pragma unroll
      for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
      for (int i = lane & 15; i < n; i += GL) {
        const float4 r = *reinterpret_cast<const float4*>(rows + 4 * i);
        good += is_inlier(Hf, r.x, r.y, r.z, r.w, t) ? 1 : 0;
      }
    }
    good = group_sum16(good);
    const unsigned long long vmask = __ballot(valid), okmask = __ballot(ok);
    // sequential replay in sample order
    for (int h = 0; h < NG; h++) {
      if (!((vmask >> (GL * h)) & 1ull)) {  // rejected sample: counts towards the 10000-attempt bound of one draw
        if (++run >= 10000) { stop = true; break; }
        continue;
      }
      run = 0;
      if (iter >= niters) { stop = true; break; }
      iter++;
      any_found = true;
      if (!((okmask >> (GL * h)) & 1ull)) continue;
      const int g = __shfl(good, GL * h);
      if (g > max(maxGood, 3)) {
        maxGood = g;
        if (lane == GL * h) {
#pragma unroll
          for (int i = 0; i < 9; i++) S.bestH[i] = H[i];
        }
        if (!force_max) niters = update_num_iters(conf, (double)(n - g) / n, 4, niters);
      }
    }
  }
  WSYNC();
  (void)any_found;
  if (info && lane == 0) { info[0] = iter; info[1] = maxGood; }
  if (maxGood <= 0) return false;
  // inlier mask of the winning hypothesis + ordered compaction of its inliers
  float Hf[8];
#pragma unroll
  for (int i = 0; i < 8; i++) Hf[i] = (float)S.bestH[i];
  int ni = 0;
  for (int c0 = 0; c0 < n; c0 += NL) {
    const int i = c0 + lane;
    bool in = false;
    float4 r = make_float4(0, 0, 0, 0);
    if (i < n) {
      r = *reinterpret_cast<const float4*>(rows + 4 * i);
      in = is_inlier(Hf, r.x, r.y, r.z, r.w, t);
      mask[i] = in ? 1 : 0;
    }
    const unsigned long long m = __ballot(in);
    if (in) *reinterpret_cast<float4*>(crow + 4 * (ni + __popcll(m & ((1ull << lane) - 1ull)))) = r;
    ni += __popcll(m);
  }
  __threadfence_block();
  WSYNC();
  if (lane < 9) S.H[lane] = S.bestH[lane];
  WSYNC();
  if (ni > 0) {
    dlt_rows(S, lane, crow, ni, S.H);  // keeps the RANSAC model when the refit is degenerate
    int it = lm_refine(S, lane, crow, ni, lm);
    if (info && lane == 0) info[2] = it;
  }
  return true;
}

// np.dot(H, (x, y, 1)) in the summation order pinned by the reference-glue fixtures: fma(h0, x, h1*y) + h2
__device__ __forceinline__ void hdot(const double* H, double x, double y, double* tx, double* ty, double* tw) {
  *tx = fma(H[0], x, H[1] * y) + H[2];
  *ty = fma(H[3], x, H[4] * y) + H[5];
  *tw = fma(H[6], x, H[7] * y) + H[8];
}

// find_point_displacement + get_largest_group_points; rbin = int scratch [n]; returns kept count
__device__ int static_filter_wave(int lane, const double* H /*LDS or regs-uniform*/, const float* rows, int n, int* rbin,
                                  float* out) {
  for (int i = lane; i < n; i += NL) {
    double tx, ty, tw;
    hdot(H, (double)rows[4 * i], (double)rows[4 * i + 1], &tx, &ty, &tw);
    double dx = tx / tw - (double)rows[4 * i + 2], dy = ty / tw - (double)rows[4 * i + 3];
    double dist = sqrt(dx * dx + dy * dy);
    rbin[i] = (int)__builtin_rint(dist);  // Python round(): half to even
  }
  __threadfence_block();
  WSYNC();
  // most populated bin; ties -> the bin whose first member comes first
  unsigned long long bestkey = 0;
  for (int i = lane; i < n; i += NL) {
    const int r = rbin[i];
    bool first = true;
    int cnt = 0;
    for (int j = 0; j < n; j++) {
      const int rj = rbin[j];
      if (rj == r) { cnt++; if (j < i) first = false; }
    }
    if (first) {
      unsigned long long key = ((unsigned long long)cnt << 32) | (unsigned)(0x7FFFFFFF - i);
      bestkey = key > bestkey ? key : bestkey;
    }
  }
  for (int s = 32; s > 0; s >>= 1) {
    unsigned long long o = __shfl_xor(bestkey, s);
    bestkey = o > bestkey ? o : bestkey;
  }
  if (n == 0) return 0;
  const int ibest = 0x7FFFFFFF - (int)(bestkey & 0xFFFFFFFFull);
  const int rbest = rbin[ibest];
  int m = 0;
  for (int c0 = 0; c0 < n; c0 += NL) {
    const int i = c0 + lane;
    const bool f = i < n && rbin[i] == rbest;
    const unsigned long long b = __ballot(f);
    if (f) *reinterpret_cast<float4*>(out + 4 * (m + __popcll(b & ((1ull << lane) - 1ull)))) =
        *reinterpret_cast<const float4*>(rows + 4 * i);
    m += __popcll(b);
  }
  return m;
}

// compute_homography (utils.py:351-362): optional pre-transform by Hsup (f64 -> f32), RANSAC #2, 0.7 gate.
__device__ int compute_homography_wave(RansacLds& S, int lane, const float* rows, int n, const double* Hsup /*LDS|null*/,
                                       const EvhRansacArgs& A, uint8_t* mask, float* trow, float* crow, double* lm,
                                       int* info) {
  const float* use = rows;
  if (Hsup) {
    for (int i = lane; i < n; i += NL) {
      double tx, ty, tw;
      hdot(Hsup, (double)rows[4 * i], (double)rows[4 * i + 1], &tx, &ty, &tw);
      float ax = (float)(tx / tw), ay = (float)(ty / tw);
      hdot(Hsup, (double)rows[4 * i + 2], (double)rows[4 * i + 3], &tx, &ty, &tw);
      float bx = (float)(tx / tw), by = (float)(ty / tw);
      *reinterpret_cast<float4*>(trow + 4 * i) = make_float4(ax, ay, bx, by);
    }
    __threadfence_block();
    WSYNC();
    use = trow;
  }
  const bool found = find_homography_wave(S, lane, use, n, A.thr, A.max_iters, A.conf, A.force_max, mask, crow, lm, info);
  __threadfence_block();
  WSYNC();
  int s = 0;
  for (int i = lane; i < n; i += NL) s += mask[i];
  s = wave_sum(s);
  if ((double)s < 0.7 * (double)n) return EVH_PAIR_LOW_INLIER_RATIO;
  if (!found) return EVH_PAIR_NO_FINAL_H;
  return EVH_PAIR_OK;
}

__shared__ RansacLds g_lds;

// generic single-problem entry (evh_find_homography_ransac)
__global__ __launch_bounds__(64) void k_find_homography(EvhRansacArgs A) {
  RansacLds& S = g_lds;
  const int lane = threadIdx.x;
  const int n = A.n_fixed;
  bool found = find_homography_wave(S, lane, A.pts, n, A.thr, A.max_iters, A.conf, A.force_max, A.mask, A.crow, A.lm,
                                    A.info);
  WSYNC();
  if (lane < 9) A.H[lane] = found ? S.H[lane] : 0.0;
  if (lane == 0) A.found[0] = found ? 1 : 0;
}

// generic static filter entry
__global__ __launch_bounds__(64) void k_static_filter(const double* H, const float* rows, int n, int* rbin, float* out,
                                                      int* count) {
  __shared__ double Hs[9];
  const int lane = threadIdx.x;
  if (lane < 9) Hs[lane] = H[lane];
  WSYNC();
  int m = static_filter_wave(lane, Hs, rows, n, rbin, out);
  if (lane == 0) count[0] = m;
}

// phase 1 of a pair: RANSAC #1 on the matched rows, then the static-point filter (matching.py:152-163)
__global__ __launch_bounds__(64) void k_ransac_static(EvhRansacArgs A) {
  RansacLds& S = g_lds;
  const int p = blockIdx.x, lane = threadIdx.x;
  if (A.status[p] != EVH_PAIR_OK) { if (lane == 0) A.npts2[p] = 0; return; }
  const int n = A.npts[p];
  const float* rows = A.pts + (int64_t)p * A.row_stride * 4;
  float* out = A.pts2 + (int64_t)p * A.row_stride * 4;
  uint8_t* mask = A.mask + (int64_t)p * A.row_stride;
  float* crow = A.crow + (int64_t)p * A.row_stride * 4;
  double* lm = A.lm + (int64_t)p * A.row_stride * 4;
  int* info = A.info ? A.info + 8 * p : nullptr;
  bool found = find_homography_wave(S, lane, rows, n, A.thr, A.max_iters, A.conf, A.force_max, mask, crow, lm, info);
  WSYNC();
  if (!found) {
    if (lane == 0) { A.status[p] = EVH_PAIR_NO_PROVISIONAL_H; A.npts2[p] = 0; }
    return;
  }
  if (A.H1 && lane < 9) A.H1[9 * p + lane] = S.H[lane];
  int* rbin = reinterpret_cast<int*>(lm);  // LM scratch is free again
  int m = static_filter_wave(lane, S.H, rows, n, rbin, out);
  if (lane == 0) A.npts2[p] = m;
}

// phase 2: compute_homography.  Independent pairs: one wave per pair, Hsup = None.
__global__ __launch_bounds__(64) void k_ransac_final_pairs(EvhRansacArgs A) {
  RansacLds& S = g_lds;
  const int p = blockIdx.x, lane = threadIdx.x;
  int st = A.status[p];
  if (st == EVH_PAIR_OK) {
    const int n = A.npts2[p];
    const float* rows = A.pts2 + (int64_t)p * A.row_stride * 4;
    st = compute_homography_wave(S, lane, rows, n, nullptr, A, A.mask + (int64_t)p * A.row_stride,
                                 A.pts + (int64_t)p * A.row_stride * 4 /* matched rows are dead: reuse as scratch */,
                                 A.crow + (int64_t)p * A.row_stride * 4, A.lm + (int64_t)p * A.row_stride * 4,
                                 A.info ? A.info + 8 * p + 4 : nullptr);
  }
  WSYNC();
  if (lane < 9) A.H[9 * p + lane] = st == EVH_PAIR_OK ? S.H[lane] : 0.0;
  if (lane == 0) A.out_status[p] = st;
}

// phase 2, stream semantics (video_processing.py:83-105): sequential scan over the pairs of one stream with the
// running superposition; a failed pair repeats the previous H (none_H_processing=True).
__global__ __launch_bounds__(64) void k_ransac_final_stream(EvhRansacArgs A, int npairs, int pitch) {
  // one wavefront per stream: block s scans the npairs pairs whose per-pair slots start at s * pitch (several streams
  // of one batch sit `pitch` pair slots apart); H / status are written compactly at s * npairs + p
  RansacLds& S = g_lds;
  __shared__ double Hsup[9], Hprev[9], Hcur[9];
  __shared__ int have_prev;
  const int lane = threadIdx.x, s = blockIdx.x;
  const double* Hsup0 = A.Hsup0 ? A.Hsup0 + 18 * s : nullptr;
  const double* Hprev0 = A.Hprev0 ? A.Hprev0 + 18 * s : nullptr;
  const int64_t slot0 = (int64_t)s * pitch;                 // first pair slot of this stream; also its scratch slot
  double* Hout = A.H + (int64_t)9 * s * npairs;
  int* stout = A.out_status + (int64_t)s * npairs;
  if (lane == 0) have_prev = Hprev0 ? 1 : 0;
  if (lane < 9 && Hsup0) Hsup[lane] = Hsup0[lane];
  if (lane < 9 && Hprev0) Hprev[lane] = Hprev0[lane];
  WSYNC();
  bool first = Hsup0 == nullptr;
  for (int p = 0; p < npairs; p++) {
    int st = A.status[slot0 + p];
    if (st == EVH_PAIR_OK) {
      const int n = A.npts2[slot0 + p];
      const float* rows = A.pts2 + (slot0 + p) * A.row_stride * 4;
      // the scan is sequential: one pair's worth of scratch (the stream's first slot) serves all its pairs
      st = compute_homography_wave(S, lane, rows, n, first ? nullptr : Hsup, A, A.mask + slot0 * A.row_stride,
                                   A.pts + slot0 * A.row_stride * 4, A.crow + slot0 * A.row_stride * 4,
                                   A.lm + slot0 * A.row_stride * 4, A.info ? A.info + 8 * (slot0 + p) + 4 : nullptr);
    }
    WSYNC();
    if (lane == 0) stout[p] = st;
    if (st != EVH_PAIR_OK && !have_prev) {
      // the reference raises here (None.tolist()); mark the pair and stop the scan
      if (lane < 9) Hout[9 * p + lane] = __longlong_as_double(0x7FF8000000000000ll);
      for (int q = p + 1 + lane; q < npairs; q += NL) { stout[q] = st; }
      for (int q = p + 1; q < npairs; q++) if (lane < 9) Hout[9 * q + lane] = __longlong_as_double(0x7FF8000000000000ll);
      return;
    }
    if (lane < 9) Hcur[lane] = st == EVH_PAIR_OK ? S.H[lane] : Hprev[lane];
    WSYNC();
    if (lane < 9) { Hout[9 * p + lane] = Hcur[lane]; Hprev[lane] = Hcur[lane]; }
    // matrix_superposition (utils.py:139-145); np.dot(3x3,3x3) = forward FMA chain (pinned by fixtures)
    double P = 0;
    if (!first && lane < 9) {
      const int r = lane / 3, c = lane - 3 * r;
      P = fma(Hcur[3 * r + 2], Hsup[6 + c], fma(Hcur[3 * r + 1], Hsup[3 + c], Hcur[3 * r] * Hsup[c]));
    }
    const double P8 = __shfl(P, 8);
    WSYNC();
    if (lane < 9) Hsup[lane] = first ? Hcur[lane] : P / P8;
    if (lane == 0) have_prev = 1;
    first = false;
    WSYNC();
  }
  if (A.state_out && lane < 9) { A.state_out[18 * s + lane] = Hsup[lane]; A.state_out[18 * s + 9 + lane] = Hprev[lane]; }
}

}  // namespace

int evh_launch_find_homography(evh_ctx* c, const EvhRansacArgs& A) {
  hipLaunchKernelGGL(k_find_homography, dim3(1), dim3(64), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
int evh_launch_static_filter(evh_ctx* c, const double* d_H, const float* d_rows, int n, int* d_rbin, float* d_out,
                             int* d_count) {
  hipLaunchKernelGGL(k_static_filter, dim3(1), dim3(64), 0, c->stream, d_H, d_rows, n, d_rbin, d_out, d_count);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
int evh_launch_ransac_static(evh_ctx* c, const EvhRansacArgs& A, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  hipLaunchKernelGGL(k_ransac_static, dim3(npairs), dim3(64), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
int evh_launch_ransac_final(evh_ctx* c, const EvhRansacArgs& A, int npairs, int nstreams, int pitch) {
  if (npairs <= 0) return EVH_SUCCESS;
  // nstreams == 0: independent pairs; otherwise nstreams sequential scans of npairs pairs each, `pitch` pair slots apart
  if (nstreams > 0) hipLaunchKernelGGL(k_ransac_final_stream, dim3(nstreams), dim3(64), 0, c->stream, A, npairs, pitch);
  else hipLaunchKernelGGL(k_ransac_final_pairs, dim3(npairs), dim3(64), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
