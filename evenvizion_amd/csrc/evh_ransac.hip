// evh_ransac.hip -- RANSAC + DLT + Levenberg-Marquardt homography estimation and the reference's point filters
// around it, for gfx950.  Replaces cv2.findHomography(a, b, cv2.RANSAC, 3.0) (reference call sites
// evenvizion/processing/matching.py:156-157 and utils.py:356-358) and the glue find_point_displacement /
// get_largest_group_points (utils.py:258-325), compute_homography (utils.py:328-363), matrix_superposition
// (utils.py:118-145).
//
// Shape of the computation.  One workgroup of NW wavefronts per frame pair (or per stream in the sequential scan).
// * RANSAC hypotheses are evaluated a chunk at a time: every 16-lane row of every wave owns one 4-point hypothesis
//   (4 * NW per chunk), or -- when the caller forces the iteration count -- every LANE does (64 * NW per chunk).  The
//   sample sequence of the fixed-seed multiply-with-carry generator is advanced by every lane identically (scalar
//   code), each owner keeps its own quadruple, solves its normalised DLT (9x9 Jacobi eigen-solver, f64) and counts its
//   inliers; a sequential replay in sample order then applies "strictly more inliers wins" and the adaptive iteration
//   bound exactly like a serial RANSAC (over-evaluated hypotheses are simply ignored).
// * The eigen-solver is the latency-critical piece: measured on this part (tools/ubench/lat.hip) a dependent f64 divide
//   costs 69 cycles, a square root 106, an LDS round trip 65-125 and a lone wave issues one instruction per ~5 cycles.
//   Three forms of the same arithmetic (same pivot order, bit-identical results):
//     jacobi_rows  -- one matrix per 16-lane row, four per wave (hypothesis chunks): pivot candidates in registers,
//                     pivot search and the four index rescans as DPP integer reductions on the raw bits of |a_ij|,
//                     rotated values rescanned before they leave registers;
//     jacobi_one   -- one matrix for the whole wave (refit, LM solves): pivot through SGPRs, W in registers, the two
//                     rescans side by side, results exchanged by v_permlane16_swap;
//     jacobi_lanes9 -- one matrix per lane (fixed-iteration mode): the serial algorithm, A and W in LDS as
//                     [element][lane], V in an L2-resident global scratch.
//   The c/s/t scalars of a rotation use the exact divide / square-root instruction sequences without the range
//   scaling they cannot need here (one check per solve; out-of-range input takes a plain-arithmetic sweep).
// * Sums over points (centroids, L^T L, J^T J, J^T r, residual norms) keep the serial summation order of the operator
//   being replaced: the points go through a 64-point LDS tile (all lanes compute their own point's terms), then one
//   lane per output entry adds the tile's terms in point order.
// All floating-point sums keep a fixed order (-ffp-contract=off); f64 throughout the solves, f32 for the
// reprojection error, as the operator being replaced.
#include "evh_internal.h"
#include "evh_ransac.h"
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

namespace {

#define NL 64          // lanes of a wavefront
#define NG 4           // 16-lane rows per wavefront: one eigen-problem (one RANSAC hypothesis) per row
#define GL 16          // lanes per row
#define MS 9           // row stride (doubles) of the matrices held in LDS, for N = 8 and N = 9
#define TS 11          // stride (doubles) of one point's terms in the LDS tile
#define TT (NL + 2)    // term-major tiles ([term][point], the single-problem stages): doubles between two terms' rows; 528
                       // bytes, so that 16-byte reads of different terms fall on different bank slots
typedef double d2_t __attribute__((ext_vector_type(2)));
#define HB 2048        // displacement-histogram bins held in LDS (larger displacements take the quadratic path)

// Ordering point between a cross-lane write and read of LDS / global scratch INSIDE one wavefront.  DS (and VMEM)
// operations of one wave execute in order, so no wait is needed: this only pins the compiler's ordering.
#define WSYNC()                                                \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  } while (0)

// ---- cross-lane reductions inside a 16-lane row (or one of its 8-lane halves) as DPP-fused integer min / max ------
#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
template <int CTRL>
__device__ __forceinline__ unsigned dmax(unsigned v) {
  return max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned dmin(unsigned v) {
  return min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dadd(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned rmax8(unsigned v) { return dmax<DPP_HALF_MIRROR>(dmax<DPP_XOR2>(dmax<DPP_XOR1>(v))); }
__device__ __forceinline__ unsigned rmin8(unsigned v) { return dmin<DPP_HALF_MIRROR>(dmin<DPP_XOR2>(dmin<DPP_XOR1>(v))); }
__device__ __forceinline__ unsigned rmax16(unsigned v) { return dmax<DPP_ROW_MIRROR>(rmax8(v)); }
__device__ __forceinline__ unsigned rmin16(unsigned v) { return dmin<DPP_ROW_MIRROR>(rmin8(v)); }
__device__ __forceinline__ int rsum16(int v) {
  return dadd<DPP_ROW_MIRROR>(dadd<DPP_HALF_MIRROR>(dadd<DPP_XOR2>(dadd<DPP_XOR1>(v))));
}
// c ? a : b as one v_cndmask_b32 (the optimiser otherwise turns the candidate updates into exec-mask branches)
__device__ __forceinline__ unsigned vsel(bool c, unsigned a, unsigned b) {
  unsigned r;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ unsigned hi32(double v) { return (unsigned)__double2hiint(v); }
__device__ __forceinline__ unsigned lo32(double v) { return (unsigned)__double2loint(v); }
__device__ __forceinline__ double mk64(unsigned hi, unsigned lo) { return __hiloint2double((int)hi, (int)lo); }

// ---- exact f64 divide / square root without the range scaling (same instruction sequences the compiler emits for
//      `/` and sqrt(), minus v_div_scale / v_div_fixup / v_ldexp): bit-identical whenever no intermediate leaves the
//      normal range, which the callers guarantee (and check) ------------------------------------------------------
struct Recip { double den, r; };
__device__ __forceinline__ Recip recip_refined(double den) {
  double r = __builtin_amdgcn_rcp(den);
  double e = fma(-den, r, 1.0);
  r = fma(r, e, r);
  e = fma(-den, r, 1.0);
  r = fma(r, e, r);
  return Recip{den, r};
}
__device__ __forceinline__ double div_by(double num, const Recip& R) {
  const double q = num * R.r;
  const double rem = fma(-R.den, q, num);
  return fma(rem, R.r, q);
}
__device__ __forceinline__ double sqrt_1_2(double x) {  // x in [1, 2]
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  return g;
}

__device__ __forceinline__ double hyp(double a, double b) {
  a = fabs(a); b = fabs(b);
  if (a > b) { b /= a; return a * sqrt(1 + b * b); }
  if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
  return 0;
}
// the rotation scalars of one Jacobi step, plain form (reference order of operations)
struct Cst { double c, s, t; };
__device__ __forceinline__ Cst rotation_scalars_plain(double p, double wk, double wl) {
  const double y = (wl - wk) * 0.5;
  double tt = fabs(y) + hyp(p, y);
  double sn = hyp(p, tt);
  Cst r;
  r.c = tt / sn;
  sn = p / sn; tt = (p / tt) * p;
  if (y < 0) sn = -sn, tt = -tt;
  r.s = sn; r.t = tt;
  return r;
}
// the same values with the short sequences: |p| > DBL_EPSILON is given, so hyp(p, y) >= |p| > 0, t >= |p| and the
// second hyp() always takes its "b >= a" branch; (p / t) * p == (|p| / t) * |p| because IEEE division and
// multiplication are sign-symmetric.  The caller has checked that no operand can be too large for the unscaled
// sequences (small ones are harmless: a quotient that loses its last bits is one whose square vanishes against 1).
__device__ __forceinline__ void rotation_scalars(double p, double wk, double wl, double& c, double& s, double& t) {
  const double y = (wl - wk) * 0.5;
  const double ap = fabs(p), ay = fabs(y);
  const bool pg = ap > ay;
  const double hi = pg ? ap : ay;
  const double lo = pg ? ay : ap;             // (a quotient too small for the short divide also vanishes against 1)
  const double q = div_by(lo, recip_refined(hi));
  const double h = hi * sqrt_1_2(1.0 + q * q);
  const double tt = ay + h;
  const double q2 = div_by(ap, recip_refined(tt));
  const double sn = tt * sqrt_1_2(1.0 + q2 * q2);
  const Recip rs = recip_refined(sn);
  c = div_by(tt, rs);
  double ss = div_by(p, rs);
  double t2 = q2 * ap;
  if (y < 0) ss = -ss, t2 = -t2;
  s = ss; t = t2;
}

struct RowMat {          // one eigen-problem: A full symmetric, V eigenvectors as rows, W eigenvalues, sort order
  double A[MS * MS];
  double V[MS * MS];     // directly behind A: jacobi_one addresses both through one element index
  double W[MS];
  int ord[12];           // ord[p] = index of the p-th largest eigenvalue (selection-sort order of the reference)
};
static_assert(offsetof(RowMat, V) == sizeof(double) * MS * MS, "V must follow A");

// Order of the eigenvalues (descending) into M.ord, rows of a wave in parallel; `active` is row-uniform.  Distinct
// values: position = number of larger ones.  Equal values (degenerate input): replay the reference's selection sort
// on the index list.
template <int N>
__device__ __forceinline__ void eig_order(RowMat& M, int lane, bool active) {
  const int gl = lane & 15;
  double* Wd = M.W;
  int gt = 0, eq = 0;
  if (active && gl < N) {
    const double w = Wd[gl];
#pragma unroll
    for (int i = 0; i < N; i++) { const double wi = Wd[i]; gt += wi > w ? 1 : 0; eq += wi == w ? 1 : 0; }
  }
  const unsigned long long tie = __ballot(active && gl < N && eq != 1);   // != 1 also catches NaN
  const bool row_tie = ((tie >> (lane & 48)) & 0xFFFFull) != 0ull;
  if (active && gl < N && !row_tie) M.ord[gt] = gl;
  if (active && row_tie && gl == 0) {
    for (int i = 0; i < N; i++) M.ord[i] = i;
    for (int a = 0; a < N - 1; a++) {
      int mm = a;
      for (int i = a + 1; i < N; i++) if (Wd[M.ord[mm]] < Wd[M.ord[i]]) mm = i;
      const int tmp = M.ord[a]; M.ord[a] = M.ord[mm]; M.ord[mm] = tmp;
    }
  }
  WSYNC();
}

// Symmetric eigen-solver (Jacobi with largest-pivot selection; eigenvalues sorted descending through M.ord), one
// matrix per 16-lane row, up to four rows of a wavefront at once.  Arithmetic and pivot order are those of the
// serial algorithm (first maximum in the scan order R0..R(N-2), C1..C(N-1); indR / indC rescanned only for the two
// rotated indices).  Lane roles inside a row (half = lanes 0-7 / 8-15, m = lane & 7):
//   half 0: rotation index m,     owner of the column candidate C(m+1) = A[indC[m+1]][m+1], column rescans (i < K)
//   half 1: rotation index m + 1, owner of the row candidate R(m) = A[m][indR[m]],         row rescans (j > K)
// A is held as a full symmetric matrix (both mirrors written), so A[own][cidx] addresses either kind of candidate.
// The diagonal of A is dead after W is taken from it (lanes k and l park their unused products there).
// Must be called by all 64 lanes; `active` is row-uniform.
template <int N>
__device__ __forceinline__ int jacobi_rows(RowMat& M, int lane, bool active) {
  const int gl = lane & 15, half = gl >> 3, m = gl & 7;
  const int idx = half ? m + 1 : m;
  const bool idx_ok = idx < N;
  const int idx_c = idx_ok ? idx : 0;
  const int own = half ? m : m + 1;
  const bool own_ok = m <= N - 2;
  const int own_c = own_ok ? own : 1;
  const unsigned prio = half ? m : 8 + m;
  const int vc = gl < N ? gl : 0;
  const int sgn = half ? 1 : -1, sidx = idx_ok ? sgn * idx : -64;
  double* A = M.A;
  double* V = M.V;
  double* Wd = M.W;
  unsigned amax = 0;                            // largest high word of |a_ij|: decides once whether the short divide /
  if (active) {                                 // square-root sequences are safe for the whole solve (see below)
    for (int e = gl; e < N * N; e += GL) {
      const int i = e / N, j = e - i * N;
      V[i * MS + j] = i == j ? 1.0 : 0.0;
      amax = max(amax, hi32(A[i * MS + j]) & 0x7FFFFFFFu);
    }
    if (gl < N) Wd[gl] = A[gl * MS + gl];
  }
  // Rotations preserve the Frobenius norm, so with every |a_ij| < 2^300 all later entries, eigenvalue estimates and
  // the hypotenuses built from them stay below 2^310: no operand of the unscaled sequences can leave their range.
  // Anything larger (or NaN) sends the whole wave through the plain `/` and sqrt() forms.
  const bool plain = __ballot(active && rmax16(amax) >= 0x52B00000u) != 0ull;
  // initial indR[own] / indC[own]: first maximum of the row right of / the column above the diagonal
  int cidx = half ? own_c + 1 : 0;
  double cval = 0;
  if (active && own_ok) {
    double mv = -1.0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const bool in = half ? j > own : j < own;
      const double v = fabs(A[own * MS + j]);
      if (in && mv < v) mv = v, cidx = j;
    }
    cval = A[own * MS + cidx];
  }
  WSYNC();
  bool act = active;
  const int maxIters = N * N * 30;
  int iters = 0;
  auto sweep = [&](auto plain_tag) {
  constexpr bool PLAIN = decltype(plain_tag)::value;
  for (; iters < maxIters; iters++) {
    if (__ballot(act) == 0ull) break;
    // ---- pivot: first maximum of |candidate| over the row's 2N-2 owners, in the order R0.., C1..
    const bool cv = act && own_ok;
    const unsigned ch = cv ? hi32(cval) & 0x7FFFFFFFu : 0u, cl = cv ? lo32(cval) : 0u;
    const unsigned mh = rmax16(ch);
    const unsigned ml = rmax16(ch == mh ? cl : 0u);
    const bool win = cv && ch == mh && cl == ml;
    const unsigned pack = (prio << 12) | ((hi32(cval) >> 31) << 8) | ((half ? own : cidx) << 4) | (half ? cidx : own);
    const unsigned pk = rmin16(win ? pack : 0xFFFFFFFFu);
    const double pabs = mk64(mh, ml);
    if (pabs <= DBL_EPSILON) act = false;
    const int k = act ? (pk >> 4) & 15 : 0, l = act ? pk & 15 : 1;
    const double p = (pk >> 8) & 1 ? -pabs : pabs;
    // ---- operands of this rotation (independent of c, s: issued before the scalar chain)
    const double wk = Wd[k], wl = Wd[l];
    const double a0 = A[idx_c * MS + k], b0 = A[idx_c * MS + l];
    const double va = V[k * MS + vc], vb = V[l * MS + vc];
    double c = 1, s = 0, t = 0;
    if (PLAIN) { const Cst r = rotation_scalars_plain(p, wk, wl); c = r.c; s = r.s; t = r.t; }
    else rotation_scalars(p, wk, wl, c, s, t);
    double u = a0 * c - b0 * s, v = a0 * s + b0 * c;
    if (idx == l) u = 0;                        // A[k][l] = 0
    if (idx == k) v = 0;
    const double nva = va * c - vb * s, nvb = va * s + vb * c;
    if (act) {
      if (idx_ok) { A[idx * MS + k] = u; A[k * MS + idx] = u; A[idx * MS + l] = v; A[l * MS + idx] = v; }
      if (gl < N) { V[k * MS + gl] = nva; V[l * MS + gl] = nvb; }
      if (gl == 0) { Wd[k] = wk - t; Wd[l] = wl + t; }
    }
    WSYNC();
    // ---- candidates: every owner re-reads its element (its index may be stale, its value never is) ...
    const double fresh = A[own_c * MS + cidx];
    // ---- ... and the owners of k and l rescan: half 0 the column above, half 1 the row right of the diagonal,
    //      straight from the rotated values in registers (u = new A[idx][k], v = new A[idx][l])
    const bool inu = sidx - sgn * k > 0;        // half 1: idx > k, half 0: idx < k (never for a lane without index)
    const bool inv = sidx - sgn * l > 0;
    const unsigned uh = inu ? hi32(u) & 0x7FFFFFFFu : 0u, ul = inu ? lo32(u) : 0u;
    const unsigned vh = inv ? hi32(v) & 0x7FFFFFFFu : 0u, vl = inv ? lo32(v) : 0u;
    const unsigned muh = rmax8(uh), mvh = rmax8(vh);
    const unsigned mul_ = rmax8(uh == muh ? ul : 0u), mvl = rmax8(vh == mvh ? vl : 0u);
    const unsigned pu = rmin8(inu && uh == muh && ul == mul_ ? (unsigned)(idx << 1) | (hi32(u) >> 31) : 0xFFFFFFFFu);
    const unsigned pv = rmin8(inv && vh == mvh && vl == mvl ? (unsigned)(idx << 1) | (hi32(v) >> 31) : 0xFFFFFFFFu);
    {
      const bool tk = act && own == k, tl = act && own == l;
      cidx = (int)vsel(tk, (pu >> 1) & 15, vsel(tl, (pv >> 1) & 15, (unsigned)cidx));
      const unsigned nh = vsel(tk, muh | (pu << 31), vsel(tl, mvh | (pv << 31), hi32(fresh)));
      const unsigned nl = vsel(tk, mul_, vsel(tl, mvl, lo32(fresh)));
      cval = mk64(nh, nl);
    }
  }
  };
  if (plain) sweep(std::true_type{}); else sweep(std::false_type{});
  WSYNC();
  eig_order<N>(M, lane, active);
  return iters;
}

// The same solver for ONE matrix served by the whole wavefront (refit, LM solves): the pivot (k, l, p) and the
// eigenvalues it needs travel through SGPRs (v_readfirstlane / v_readlane), the rotations of A and V are one
// instruction stream (rows 0-1 of the wave rotate A pairs, row 2 the V pairs), the rescans for k and for l run side
// by side (row 0 / row 1) and row 1's results reach the candidate owners in row 0 by v_permlane16_swap.  W lives in
// registers (lane j holds W[j]) and is stored to M.W at the end.  Same arithmetic, same pivot order, same result.
template <int N>
__device__ __forceinline__ int jacobi_one(RowMat& M, int lane) {
  const int row = lane >> 4, gl = lane & 15, half = gl >> 3, m = gl & 7;
  const int idx = half ? m + 1 : m;
  const bool a_lane = row < 2 && idx < N;
  const bool a_writer = row == 0 && idx < N;
  const bool v_lane = row == 2 && gl < N;
  const int own = half ? m : m + 1;
  const bool own_ok = row == 0 && m <= N - 2;
  const int own_c = m <= N - 2 ? own : 1;
  const unsigned prio = half ? m : 8 + m;
  const int sgn = half ? 1 : -1, sidx = a_lane ? sgn * idx : -64;
  const int zidx = a_lane ? idx : 99;
  // element index of this lane's pair inside M.A (V behind it): e0 = ebase + k * emult, e1 = ebase + l * emult
  const int ebase = a_lane ? idx * MS : v_lane ? MS * MS + gl : 0;
  const int emult = a_lane ? 1 : v_lane ? MS : 0;
  double* D = M.A;
  unsigned amax = 0;
  for (int e = lane; e < N * N; e += NL) {
    const int i = e / N, j = e - i * N;
    M.V[i * MS + j] = i == j ? 1.0 : 0.0;
    amax = max(amax, hi32(D[i * MS + j]) & 0x7FFFFFFFu);
  }
  const bool plain = __ballot(amax >= 0x52B00000u) != 0ull;     // see jacobi_rows
  double wreg = lane < N ? D[lane * MS + lane] : 0.0;
  int cidx = half ? own_c + 1 : 0;
  double cval = 0;
  if (own_ok) {
    double mv = -1.0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const bool in = half ? j > own : j < own;
      const double v = fabs(D[own * MS + j]);
      if (in && mv < v) mv = v, cidx = j;
    }
    cval = D[own * MS + cidx];
  }
  WSYNC();
  const int maxIters = N * N * 30;
  int iters = 0;
  auto sweep = [&](auto plain_tag) {
  constexpr bool PLAIN = decltype(plain_tag)::value;
  for (; iters < maxIters; iters++) {
    // ---- pivot (row 0 holds the candidates; the other rows reduce zeros)
    const unsigned ch = own_ok ? hi32(cval) & 0x7FFFFFFFu : 0u, cl = own_ok ? lo32(cval) : 0u;
    const unsigned mh = rmax16(ch);
    const unsigned ml = rmax16(ch == mh ? cl : 0u);
    const bool win = own_ok && ch == mh && cl == ml;
    const unsigned pack = (prio << 12) | ((hi32(cval) >> 31) << 8) | ((half ? own : cidx) << 4) | (half ? cidx : own);
    const unsigned pk = rmin16(win ? pack : 0xFFFFFFFFu);
    const unsigned spk = (unsigned)__builtin_amdgcn_readfirstlane((int)pk);
    const double pabs = mk64((unsigned)__builtin_amdgcn_readfirstlane((int)mh), (unsigned)__builtin_amdgcn_readfirstlane((int)ml));
    if (pabs <= DBL_EPSILON) break;
    const int k = (spk >> 4) & 15, l = spk & 15;
    const double p = (spk >> 8) & 1 ? -pabs : pabs;
    const double wk = mk64((unsigned)__builtin_amdgcn_readlane((int)hi32(wreg), k), (unsigned)__builtin_amdgcn_readlane((int)lo32(wreg), k));
    const double wl = mk64((unsigned)__builtin_amdgcn_readlane((int)hi32(wreg), l), (unsigned)__builtin_amdgcn_readlane((int)lo32(wreg), l));
    const int e0 = ebase + k * emult, e1 = ebase + l * emult;
    const double a0 = D[e0], b0 = D[e1];
    double c = 1, s = 0, t = 0;
    if (PLAIN) { const Cst r = rotation_scalars_plain(p, wk, wl); c = r.c; s = r.s; t = r.t; }
    else rotation_scalars(p, wk, wl, c, s, t);
    double x0 = a0 * c - b0 * s, x1 = a0 * s + b0 * c;
    if (zidx == l) x0 = 0;                      // A[k][l] = 0
    if (zidx == k) x1 = 0;
    if (a_writer || v_lane) { D[e0] = x0; D[e1] = x1; }
    if (a_writer) { D[k * MS + idx] = x0; D[l * MS + idx] = x1; }
    {
      const double wm = wreg - t, wp = wreg + t;
      wreg = lane == k ? wm : lane == l ? wp : wreg;
    }
    WSYNC();
    const double fresh = D[own_c * MS + cidx];
    // ---- rescans: row 0 for k on the first components, row 1 for l on the second ones
    const double xs = row == 1 ? x1 : x0;
    const int Ks = row == 1 ? l : k;
    const bool inr = sidx - sgn * Ks > 0;
    const unsigned xh = inr ? hi32(xs) & 0x7FFFFFFFu : 0u, xl = inr ? lo32(xs) : 0u;
    const unsigned m8h = rmax8(xh);
    const unsigned m8l = rmax8(xh == m8h ? xl : 0u);
    const unsigned pw = rmin8(inr && xh == m8h && xl == m8l ? (unsigned)(idx << 1) | (hi32(xs) >> 31) : 0xFFFFFFFFu);
    const unsigned o_pw = __builtin_amdgcn_permlane16_swap(pw, pw, false, false)[1];     // row 0 <- row 1
    const unsigned o_h = __builtin_amdgcn_permlane16_swap(m8h, m8h, false, false)[1];
    const unsigned o_l = __builtin_amdgcn_permlane16_swap(m8l, m8l, false, false)[1];
    {
      const bool tk = own == k, tl = own == l;
      cidx = (int)vsel(tk, (pw >> 1) & 15, vsel(tl, (o_pw >> 1) & 15, (unsigned)cidx));
      const unsigned nh = vsel(tk, m8h | (pw << 31), vsel(tl, o_h | (o_pw << 31), hi32(fresh)));
      const unsigned nl = vsel(tk, m8l, vsel(tl, o_l, lo32(fresh)));
      cval = mk64(nh, nl);
    }
  }
  };
  if (plain) sweep(std::true_type{}); else sweep(std::false_type{});
  if (lane < N) M.W[lane] = wreg;
  WSYNC();
  eig_order<N>(M, lane, lane < GL);
  return iters;
}

// de-normalise the smallest-eigenvalue eigenvector into H (runKernel's tail)
__device__ __forceinline__ void dlt_finish_from(const double* H0, double cmx, double cmy, double smx, double smy, double cMx,
                                                double cMy, double sMx, double sMy, double* H) {
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
__device__ __forceinline__ void dlt_finish(const RowMat& M, double cmx, double cmy, double smx, double smy, double cMx,
                                           double cMy, double sMx, double sMy, double* H) {
  double H0[9];
  const int r8 = M.ord[8];
#pragma unroll
  for (int i = 0; i < 9; i++) H0[i] = M.V[r8 * MS + i];
  dlt_finish_from(H0, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
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
// upper-triangle entry number e (0..44, row-major) of a 9x9 -> (j, k), j <= k
__device__ __forceinline__ void tri9(int e, int& j, int& k) {
  j = 0;
  while (e >= 9 - j) { e -= 9 - j; j++; }
  k = j + e;
}
__device__ __forceinline__ void tri8(int e, int& i, int& j) {
  i = 0;
  while (e >= 8 - i) { e -= 8 - i; i++; }
  j = i + e;
}

// normalised DLT of each row's own 4 correspondences (M -> m): every lane of a row holds the same 4 points.
// `valid` is row-uniform; returns (row-uniform) whether a model was produced; H valid on every lane of the row.
__device__ __forceinline__ bool dlt4_rows(RowMat& M, int lane, bool valid, const float* Mx, const float* My,
                                          const float* mx, const float* my, double* H) {
  const int gl = lane & 15;
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
    for (int e = gl; e < 45; e += GL) {          // L^T L upper triangle, entry e <-> (j, k), points summed in order
      int j, k;
      tri9(e, j, k);
      double acc = 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
        acc += ltl_term(j, k, (mx[i] - cmx) * smx, (my[i] - cmy) * smy, (Mx[i] - cMx) * sMx, (My[i] - cMy) * sMy);
      M.A[j * MS + k] = acc;
      M.A[k * MS + j] = acc;
    }
  }
  WSYNC();
  jacobi_rows<9>(M, lane, ok);
  if (ok) dlt_finish(M, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
  return ok;
}


// ---- one 9x9 eigen-problem PER LANE (fixed-iteration RANSAC: thousands of hypotheses, throughput matters, latency does
// not): the serial algorithm as it stands, 64 problems side by side.  A and W of every lane live in LDS as
// [element][lane] (512 bytes between elements: whatever element a lane picks, it stays on its own banks): 0..35 upper
// off-diagonal of A (row i starts at i(17-i)/2), 36..44 W.  V (81 elements per lane, written and read only by its own
// lane, never on the pivot's critical path) lives in a global scratch of the same [element][lane] shape -- it stays in
// L2 -- so that four waves fit a compute unit instead of two; its loads are issued before the rotation scalars.  indR / indC are
// one register per index.  The rescans of indR / indC for the two rotated indices run inside the
// rotation loop on the freshly rotated values (ascending index, strict '<': the first maximum, as the reference).
#define LM_ELEMS 45
#define LM_W 36
#define LV_ELEMS 81
__device__ __forceinline__ int lm_arow(int i) { return (i * (17 - i)) >> 1; }          // first element of row i (j = i+1)
__device__ __forceinline__ int lm_a(int i, int j) { return lm_arow(i) + j - i - 1; }     // i < j
// L = this lane's column: element e at L[e * 64].  A (upper) and W (diagonal) hold the input.  Returns the row of V
// that belongs to the smallest eigenvalue under the reference's selection sort.
__device__ __forceinline__ int jacobi_lanes9(double* L, double* Vg, bool active) {
  const int N = 9;
#define EL(e) L[(e) * NL]
#define VL(e) Vg[(e) * NL]
  if (active) {
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
      for (int j = 0; j < N; j++) VL(i * N + j) = i == j ? 1.0 : 0.0;
  }
  int indR[9], indC[9];                // one register each (static index in every loop below)
#pragma unroll
  for (int k = 0; k < N; k++) { indR[k] = k < N - 1 ? k + 1 : 0; indC[k] = 0; }
  if (active) {
#pragma unroll
    for (int k = 0; k < N; k++) {
      if (k < N - 1) {
        int m = k + 1; double mv = fabs(EL(lm_a(k, k + 1)));
#pragma unroll
        for (int i = k + 2; i < N; i++) { const double v = fabs(EL(lm_a(k, i))); if (mv < v) mv = v, m = i; }
        indR[k] = m;
      }
      if (k > 0) {
        int m = 0; double mv = fabs(EL(lm_a(0, k)));
#pragma unroll
        for (int i = 1; i < k; i++) { const double v = fabs(EL(lm_a(i, k))); if (mv < v) mv = v, m = i; }
        indC[k] = m;
      }
    }
  }
  // the short divide / square-root sequences need every |a_ij| < 2^300 (see jacobi_rows); one lane out of range sends
  // the wave through the plain forms
  unsigned amax = 0;
  if (active) {
#pragma unroll
    for (int e = 0; e < LM_ELEMS; e++) amax = max(amax, hi32(EL(e)) & 0x7FFFFFFFu);
  }
  const bool plain = __ballot(active && amax >= 0x52B00000u) != 0ull;
  bool act = active;
  for (int iters = 0; iters < N * N * 30; iters++) {
    if (__ballot(act) == 0ull) break;
    // pivot: rows 0..7 through indR, then columns 1..8 through indC; first maximum.  All sixteen candidates are
    // loaded first (one LDS latency), then compared in order.
    int ci[16]; double cvv[16];
#pragma unroll
    for (int i = 0; i < N - 1; i++) { ci[i] = indR[i]; cvv[i] = EL(lm_arow(i) + ci[i] - i - 1); }
#pragma unroll
    for (int i = 1; i < N; i++) { ci[7 + i] = indC[i]; cvv[7 + i] = EL(lm_arow(ci[7 + i]) + i - ci[7 + i] - 1); }
    int k = 0, l = ci[0];
    double p = cvv[0], mv = fabs(p);
#pragma unroll
    for (int i = 1; i < N - 1; i++) {
      const bool b = mv < fabs(cvv[i]);
      mv = b ? fabs(cvv[i]) : mv; p = b ? cvv[i] : p; k = b ? i : k; l = b ? ci[i] : l;
    }
#pragma unroll
    for (int i = 1; i < N; i++) {
      const bool b = mv < fabs(cvv[7 + i]);
      mv = b ? fabs(cvv[7 + i]) : mv; p = b ? cvv[7 + i] : p; k = b ? ci[7 + i] : k; l = b ? i : l;
    }
    if (mv <= DBL_EPSILON) act = false;
    k = act ? k : 0; l = act ? l : 1;
    const int rowk = lm_arow(k), rowl = lm_arow(l);
    // operands of the rotation: the pairs of A (dummy element 0 for i == k, l) and of V, loaded before the scalars
    int e1[9], e2[9];
    double a0[9], b0[9], va[9], vb[9];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const bool rot = i != k && i != l;
      e1[i] = rot ? (i < k ? lm_arow(i) + k - i - 1 : rowk + i - k - 1) : 0;
      e2[i] = rot ? (i < l ? lm_arow(i) + l - i - 1 : rowl + i - l - 1) : 0;
      a0[i] = EL(e1[i]); b0[i] = EL(e2[i]);
      va[i] = VL(k * N + i); vb[i] = VL(l * N + i);
    }
    const double wk = EL(LM_W + k), wl = EL(LM_W + l);
    double c = 1, sn = 0, t = 0;
    if (plain) { const Cst r = rotation_scalars_plain(p, wk, wl); c = r.c; sn = r.s; t = r.t; }
    else rotation_scalars(p, wk, wl, c, sn, t);
    // rotate; the rescans of indR / indC for k and l run on the fresh values: row k right of the diagonal holds
    // A[k][l] = 0 at index l, column l above the diagonal holds it at index k
    int mRk = 0, mCk = 0, mRl = 0, mCl = 0;
    double vRk = -1, vCk = -1, vRl = -1, vCl = -1;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const bool rot = i != k && i != l;
      const double u = rot ? a0[i] * c - b0[i] * sn : 0.0, v = rot ? a0[i] * sn + b0[i] * c : 0.0;
      if (act && rot) { EL(e1[i]) = u; EL(e2[i]) = v; }
      const double au = fabs(u), av = fabs(v);
      const bool rk = i > k && vRk < au, ck = i < k && vCk < au, rl = i > l && vRl < av, cl = i < l && vCl < av;
      vRk = rk ? au : vRk; mRk = rk ? i : mRk;
      vCk = ck ? au : vCk; mCk = ck ? i : mCk;
      vRl = rl ? av : vRl; mRl = rl ? i : mRl;
      vCl = cl ? av : vCl; mCl = cl ? i : mCl;
    }
    if (act) {
      EL(rowk + l - k - 1) = 0.0;
      EL(LM_W + k) = wk - t; EL(LM_W + l) = wl + t;
#pragma unroll
      for (int i = 0; i < N; i++) {
        VL(k * N + i) = va[i] * c - vb[i] * sn;
        VL(l * N + i) = va[i] * sn + vb[i] * c;
      }
    }
    // indR / indC of the two rotated indices (row N-1 has no indR, column 0 no indC: those registers are never read)
#pragma unroll
    for (int i = 0; i < N; i++) {
      indR[i] = act && i == k ? mRk : act && i == l ? mRl : indR[i];
      indC[i] = act && i == k ? mCk : act && i == l ? mCl : indC[i];
    }
  }
  // selection sort (descending) on the index list: only the row that ends last is needed
  double w[9]; int pm[9];
#pragma unroll
  for (int i = 0; i < N; i++) { w[i] = active ? EL(LM_W + i) : 0.0; pm[i] = i; }
#pragma unroll
  for (int a = 0; a < N - 1; a++) {
    double bv = w[a]; int bi = a, bp = pm[a];
#pragma unroll
    for (int i = a + 1; i < N; i++) if (bv < w[i]) bv = w[i], bi = i, bp = pm[i];
#pragma unroll
    for (int i = a + 1; i < N; i++) if (i == bi) { w[i] = w[a]; pm[i] = pm[a]; }
    w[a] = bv; pm[a] = bp;
  }
  return pm[N - 1];
#undef EL
#undef VL
}

// normalised DLT of this LANE's own 4 correspondences (M -> m); returns whether a model was produced
__device__ __forceinline__ bool dlt4_lane(double* L, double* Vg, bool valid, const float* Mx, const float* My, const float* mx,
                                          const float* my, double* H) {
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
    double x[4], y[4], X[4], Y[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      x[i] = (mx[i] - cmx) * smx; y[i] = (my[i] - cmy) * smy; X[i] = (Mx[i] - cMx) * sMx; Y[i] = (My[i] - cMy) * sMy;
    }
#pragma unroll
    for (int j = 0; j < 9; j++)
#pragma unroll
      for (int k = j; k < 9; k++) {
        double acc = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) acc += ltl_term(j, k, x[i], y[i], X[i], Y[i]);
        L[(j == k ? LM_W + j : lm_a(j, k)) * NL] = acc;
      }
  }
  const int r8 = jacobi_lanes9(L, Vg, ok);
  if (ok) {
    double H0[9];
#pragma unroll
    for (int i = 0; i < 9; i++) H0[i] = Vg[(r8 * 9 + i) * NL];
    dlt_finish_from(H0, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
  }
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
// x % n for 32-bit x through the 64-bit reciprocal M = ceil(2^64 / n): floor(x * M / 2^64) == x / n for every 32-bit x
struct FastMod {
  unsigned n, mlo, mhi;
  __device__ explicit FastMod(unsigned n_) : n(n_) {
    const unsigned long long M = 0xFFFFFFFFFFFFFFFFull / n_ + 1ull;
    mlo = (unsigned)M; mhi = (unsigned)(M >> 32);
  }
  __device__ __forceinline__ unsigned mod(unsigned x) const {
    const unsigned long long t = (unsigned long long)x * mhi + __umulhi(x, mlo);
    return x - (unsigned)(t >> 32) * n;
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

__device__ __forceinline__ bool check_subset4(const float* Mx, const float* My, const float* mx, const float* my) {
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

// optional in-kernel cycle accounting (EVH_RANSAC_PROF=1): slots of A.prof, accumulated by thread 0
enum { PF_CALLS = 0, PF_HYP, PF_CHUNKS, PF_COMPACT, PF_REFIT, PF_LM, PF_LM_ITERS, PF_SOLVE8, PF_EVAL, PF_TOTAL, PF_ROT9,
       PF_ROT8, PF_SETUP, PF_RNG, PF_COUNT, PF_BARRIER, PF_REPLAY, PF_MW_W0, PF_MW_W1, PF_MW_W2, PF_MW_W3, PF_MW_WAIT, PF_MW_STEPS, PF_NSLOTS };
__device__ __forceinline__ unsigned long long pf_now() { return __builtin_readcyclecounter(); }
// inside loops: s_memtime is a scalar memory instruction -- its result is waited for with lgkmcnt(0), which also drains every
// LDS read in flight -- so the counter is read only when the accounting is on (prof is wave-uniform)
__device__ __forceinline__ unsigned long long pf_now_if(const unsigned long long* prof) { return prof ? __builtin_readcyclecounter() : 0ull; }
__device__ __forceinline__ void pf_add(unsigned long long* prof, int slot, unsigned long long v) {
  if (prof && threadIdx.x == 0) atomicAdd(prof + slot, v);
}
__device__ __forceinline__ void pf_add_wave(unsigned long long* prof, int slot, unsigned long long v) {   // lane 0 of any wave
  if (prof && (threadIdx.x & 63) == 0) atomicAdd(prof + slot, v);
}

struct SolveLds {        // scratch of the single-problem stages (refit, LM): used by wave 0 only
  double bestH[9];
  double H[9];           // result of the last single-problem DLT / LM
  double x[8], xd[8], v[8], d[8], D[8], tmpd[8], A8[64], Ap[64], Inv[64];
  double sc[8];          // scalars: S, Sd, lambda, lc, nu, rmax ...
  int ib[8];             // ints: proceed flags, counts
  int fast;              // EvhRansacArgs::fast_solver (set by the kernels before any solve)
  alignas(16) double T[NL * TS];     // the 64-point tile (16-byte LDS accesses: ds_read_b128 costs a quarter of the 8-byte forms)
  double P2[NL / 2 + 2]; // lm_eval: squared residuals of a tile, summed per pair of points (+ the two terms of an odd last point)
};

// ---- tolerance mode (EVH_SOLVER_FAST) of the refit's sums: lane-strided partial sums + a butterfly instead of the point-order
// chains (see lm_eval_fast).  With a = (X, Y, 1): Lx = (a, 0, -x a), Ly = (0, a, -y a), so L^T L needs sum a_i a_j, sum x a_i a_j,
// sum y a_i a_j and sum (x^2 + y^2) a_i a_j -- 24 sums instead of 45 chains.  The eigen-solve and the de-normalisation are shared.
__device__ __forceinline__ double wave_sum_f64_(double v) {
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
  return v;
}
__device__ void fast_solve8(int lane, const double* A, const double* b, double* x, int* ok);
struct SolveLds;
__device__ __forceinline__ bool dlt_rows_fast(SolveLds& S, RowMat& M, int lane, const float* rows, int count, double* Hout /* LDS */,
                                              unsigned long long* prof) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int i = lane; i < count; i += NL) {
    const float4 r = *reinterpret_cast<const float4*>(rows + 4 * i);
    c0 += r.z; c1 += r.w; c2 += r.x; c3 += r.y;
  }
  const double cmx = wave_sum_f64_(c0) / count, cmy = wave_sum_f64_(c1) / count, cMx = wave_sum_f64_(c2) / count, cMy = wave_sum_f64_(c3) / count;
  c0 = c1 = c2 = c3 = 0;
  for (int i = lane; i < count; i += NL) {
    const float4 r = *reinterpret_cast<const float4*>(rows + 4 * i);
    c0 += fabs(r.z - cmx); c1 += fabs(r.w - cmy); c2 += fabs(r.x - cMx); c3 += fabs(r.y - cMy);
  }
  double smx = wave_sum_f64_(c0), smy = wave_sum_f64_(c1), sMx = wave_sum_f64_(c2), sMy = wave_sum_f64_(c3);
  if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return false;
  smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
  double aa[6] = {0, 0, 0, 0, 0, 0}, xa[6] = {0, 0, 0, 0, 0, 0}, ya[6] = {0, 0, 0, 0, 0, 0}, qa[6] = {0, 0, 0, 0, 0, 0};
  for (int i = lane; i < count; i += NL) {
    const float4 r = *reinterpret_cast<const float4*>(rows + 4 * i);
    const double x = (r.z - cmx) * smx, y = (r.w - cmy) * smy;
    const double X = (r.x - cMx) * sMx, Y = (r.y - cMy) * sMy;
    const double p[6] = {X * X, X * Y, X, Y * Y, Y, 1.0};           // a_i a_j for (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    const double q = x * x + y * y;
#pragma unroll
    for (int k = 0; k < 6; k++) { aa[k] += p[k]; xa[k] += x * p[k]; ya[k] += y * p[k]; qa[k] += q * p[k]; }
  }
#pragma unroll
  for (int k = 0; k < 6; k++) { aa[k] = wave_sum_f64_(aa[k]); xa[k] = wave_sum_f64_(xa[k]); ya[k] = wave_sum_f64_(ya[k]); qa[k] = wave_sum_f64_(qa[k]); }
  if (lane == 0) {
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) M.A[i * MS + j] = 0.0;
    const int ui[6] = {0, 0, 0, 1, 1, 2}, uj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int i = ui[k], j = uj[k];
      M.A[i * MS + j] = aa[k]; M.A[j * MS + i] = aa[k];
      M.A[(3 + i) * MS + 3 + j] = aa[k]; M.A[(3 + j) * MS + 3 + i] = aa[k];
      M.A[(6 + i) * MS + 6 + j] = qa[k]; M.A[(6 + j) * MS + 6 + i] = qa[k];
      M.A[i * MS + 6 + j] = -xa[k]; M.A[j * MS + 6 + i] = -xa[k]; M.A[(6 + j) * MS + i] = -xa[k]; M.A[(6 + i) * MS + j] = -xa[k];
      M.A[(3 + i) * MS + 6 + j] = -ya[k]; M.A[(3 + j) * MS + 6 + i] = -ya[k]; M.A[(6 + j) * MS + 3 + i] = -ya[k]; M.A[(6 + i) * MS + 3 + j] = -ya[k];
    }
  }
  WSYNC();
  // The refit only seeds the LM refinement, so in this mode the smallest eigenvector of L^T L (117 Jacobi rotations) gives way
  // to the inhomogeneous least-squares solution with h33 = 1 in the normalised frame: one 8x8 LDL^T.  A pivot that is not
  // positive, or a solution that is not finite, falls back to the eigen-solve.
  if (lane == 0) {
    for (int i = 0; i < 8; i++) {
      for (int j = 0; j < 8; j++) S.Ap[i * 8 + j] = M.A[i * MS + j];
      S.tmpd[i] = -M.A[i * MS + 8];
    }
  }
  WSYNC();
  fast_solve8(lane, S.Ap, S.tmpd, S.d, &S.ib[2]);
  WSYNC();
  bool direct = S.ib[2] != 0;
  if (direct) {
    double mx = 0;
    for (int i = 0; i < 8; i++) mx = fmax(mx, fabs(S.d[i]));
    direct = mx < 1e12;                                   // (NaN compares false)
  }
  if (direct) {
    if (lane == 0) {
      double H0[9], H[9];
      for (int i = 0; i < 8; i++) H0[i] = S.d[i];
      H0[8] = 1.0;
      dlt_finish_from(H0, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
      for (int i = 0; i < 9; i++) Hout[i] = H[i];
    }
    WSYNC();
    return true;
  }
  pf_add(prof, PF_ROT9, jacobi_one<9>(M, lane));
  if (lane == 0) {
    double H[9];
    dlt_finish(M, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
    for (int i = 0; i < 9; i++) Hout[i] = H[i];
  }
  WSYNC();
  return true;
}

// ---- single-problem normalised DLT on `count` rows (ax,ay,bx,by): sums in row order, one lane per sum; wave 0 ------
__device__ __forceinline__ bool dlt_rows(SolveLds& S, RowMat& M, int lane, const float* rows, int count, double* Hout /* LDS */,
                                         unsigned long long* prof = nullptr) {
  if (S.fast && count > 4) return dlt_rows_fast(S, M, lane, rows, count, Hout, prof);
  double* T = S.T;
  // centroids: lanes 0..3 own cm.x, cm.y, cM.x, cM.y  (m = b columns, M = a columns)
  double acc = 0;
  const float4 first_rows = lane < count ? *reinterpret_cast<const float4*>(rows + 4 * lane) : make_float4(0, 0, 0, 0);
  float4 rnext = first_rows;
  for (int c0 = 0; c0 < count; c0 += NL) {
    const int i = c0 + lane;
    const float4 r = rnext;                       // requested one tile ahead
    if (i + NL < count) rnext = *reinterpret_cast<const float4*>(rows + 4 * (i + NL));
    if (i < count) {
      T[lane] = r.z; T[TT + lane] = r.w; T[2 * TT + lane] = r.x; T[3 * TT + lane] = r.y;
    }
    WSYNC();
    const int cnt = min(NL, count - c0);
    if (lane < 4) {
      int j = 0;
      for (; j + 16 <= cnt; j += 16) {
        d2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const d2_t*>(T + lane * TT + j + 2 * u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) { acc += v[u].x; acc += v[u].y; }
      }
      for (; j < cnt; j++) acc += T[lane * TT + j];
    }
    WSYNC();
  }
  if (lane < 4) acc /= count;
  const double cmx = __shfl(acc, 0), cmy = __shfl(acc, 1), cMx = __shfl(acc, 2), cMy = __shfl(acc, 3);
  double dev = 0;
  const double mycen = lane == 0 ? cmx : lane == 1 ? cmy : lane == 2 ? cMx : cMy;
  rnext = first_rows;
  for (int c0 = 0; c0 < count; c0 += NL) {
    const int i = c0 + lane;
    const float4 r = rnext;                       // requested one tile ahead
    if (i + NL < count) rnext = *reinterpret_cast<const float4*>(rows + 4 * (i + NL));
    if (i < count) {
      T[lane] = r.z; T[TT + lane] = r.w; T[2 * TT + lane] = r.x; T[3 * TT + lane] = r.y;
    }
    WSYNC();
    const int cnt = min(NL, count - c0);
    if (lane < 4) {
      int j = 0;
      for (; j + 16 <= cnt; j += 16) {
        d2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const d2_t*>(T + lane * TT + j + 2 * u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) { dev += fabs(v[u].x - mycen); dev += fabs(v[u].y - mycen); }
      }
      for (; j < cnt; j++) dev += fabs(T[lane * TT + j] - mycen);
    }
    WSYNC();
  }
  double smx = __shfl(dev, 0), smy = __shfl(dev, 1), sMx = __shfl(dev, 2), sMy = __shfl(dev, 3);
  if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
    return false;
  smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
  // L^T L upper triangle: lane e <-> entry (j,k), sequential over the points.  A point's terms in the tile:
  // 0:X 1:Y 2:1 3:0 4:-xX 5:-xY 6:-x 7:-yX 8:-yY 9:-y ; Lx = {0,1,2,3,3,3,4,5,6}, Ly = {3,3,3,0,1,2,7,8,9}
  int ej = 0, ek = 0;
  if (lane < 45) tri9(lane, ej, ek);
  const int lxj = ej < 3 ? ej : ej < 6 ? 3 : ej - 2, lxk = ek < 3 ? ek : ek < 6 ? 3 : ek - 2;
  const int lyj = ej < 3 ? 3 : ej < 6 ? ej - 3 : ej + 1, lyk = ek < 3 ? 3 : ek < 6 ? ek - 3 : ek + 1;
  double s = 0;
  rnext = first_rows;
  for (int c0 = 0; c0 < count; c0 += NL) {
    const int i = c0 + lane;
    const float4 r = rnext;
    if (i + NL < count) rnext = *reinterpret_cast<const float4*>(rows + 4 * (i + NL));
    if (i < count) {
      const double x = (r.z - cmx) * smx, y = (r.w - cmy) * smy;
      const double X = (r.x - cMx) * sMx, Y = (r.y - cMy) * sMy;
      double* t = T + lane;                                 // term j of this point at t[j * TT]
      t[0] = X; t[TT] = Y; t[2 * TT] = 1.0; t[3 * TT] = 0.0; t[4 * TT] = -x * X; t[5 * TT] = -x * Y; t[6 * TT] = -x;
      t[7 * TT] = -y * X; t[8 * TT] = -y * Y; t[9 * TT] = -y;
    }
    WSYNC();
    const int cnt = min(NL, count - c0);
    if (lane < 45) {
      // four points' operands requested before the first product: the compiler otherwise waits for the LDS after every
      // point (measured on lm_eval_mw's loops: one round trip per read group, 3x the time)
      int j = 0;
      for (; j + 8 <= cnt; j += 8) {                        // eight points: four 16-byte reads per operand row
        d2_t a[4], b[4], c[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          a[u] = *reinterpret_cast<const d2_t*>(T + lxj * TT + j + 2 * u); b[u] = *reinterpret_cast<const d2_t*>(T + lxk * TT + j + 2 * u);
          c[u] = *reinterpret_cast<const d2_t*>(T + lyj * TT + j + 2 * u); d[u] = *reinterpret_cast<const d2_t*>(T + lyk * TT + j + 2 * u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) { s += a[u].x * b[u].x + c[u].x * d[u].x; s += a[u].y * b[u].y + c[u].y * d[u].y; }
      }
      for (; j < cnt; j++) s += T[lxj * TT + j] * T[lxk * TT + j] + T[lyj * TT + j] * T[lyk * TT + j];
    }
    WSYNC();
  }
  if (lane < 45) { M.A[ej * MS + ek] = s; M.A[ek * MS + ej] = s; }
  WSYNC();
  pf_add(prof, PF_ROT9, jacobi_one<9>(M, lane));
  if (lane == 0) {
    double H[9];
    dlt_finish(M, cmx, cmy, smx, smy, cMx, cMy, sMx, sMy, H);
    for (int i = 0; i < 9; i++) Hout[i] = H[i];
  }
  WSYNC();
  return true;
}

// ---- symmetric solve / inverse through the eigen-decomposition (cv::solve / cv::invert, DECOMP_EIGEN) -------------
// all 64 lanes of wave 0; Ain / b / x live in LDS.  Back-substitution keeps the serial summation orders: lane i forms
// s_i = (sum_j u_i[j] b[j]) / w_i, lane j accumulates x[j] += s_i u_i[j] over i ascending.
__device__ __forceinline__ void eig_solve8_wave(RowMat& M, int lane, const double* Ain /*LDS 64*/, const double* b /*LDS 8 or null*/,
                                double* x /*LDS 8 or 64*/, unsigned long long* prof = nullptr) {
  const unsigned long long pt0 = pf_now_if(prof);
  {
    const int i = lane >> 3, j = lane & 7;
    M.A[i * MS + j] = Ain[min(i, j) * 8 + max(i, j)];
  }
  WSYNC();
  pf_add(prof, PF_ROT8, jacobi_one<8>(M, lane));
  double threshold = 0;
  for (int i = 0; i < 8; i++) threshold += M.W[M.ord[i]];
  threshold *= DBL_EPSILON * 2;
  if (b) {
    // s_i on lane i (0 for skipped eigenvalues is NOT equivalent to skipping: keep a flag)
    double si = 0; bool use = false;
    if (lane < 8) {
      const int r = M.ord[lane];
      double wi = M.W[r];
      if (!(fabs(wi) <= threshold)) {
        use = true;
        wi = 1 / wi;
        double acc = 0;
        for (int j = 0; j < 8; j++) acc += M.V[r * MS + j] * b[j];
        si = acc * wi;
      }
    }
    double xj = 0;
    for (int i = 0; i < 8; i++) {
      const double s_i = __shfl(si, i);
      const int u_i = __shfl((int)use, i);
      if (u_i && lane < 8) xj = xj + s_i * M.V[M.ord[i] * MS + lane];
    }
    if (lane < 8) x[lane] = xj;
  } else {
    // inverse: x[r][j] += u_i[r] * (u_i[j] / w_i) over i ascending; lane = r*8 + j
    const int r = lane >> 3, j = lane & 7;
    double acc = 0;
    for (int i = 0; i < 8; i++) {
      const int ri = M.ord[i];
      double wi = M.W[ri];
      if (fabs(wi) <= threshold) continue;
      wi = 1 / wi;
      const double sj = M.V[ri * MS + j] * wi;
      acc = acc + M.V[ri * MS + r] * sj;
    }
    x[r * 8 + j] = acc;
  }
  WSYNC();
  pf_add(prof, PF_SOLVE8, pf_now_if(prof) - pt0);
}

// acc + v[0] + v[1] + ... + v[n-1] in that order, the words requested sixteen at a time (the compiler alone waits for the
// LDS after every single read of such a chain)
__device__ __forceinline__ double add_in_order(double acc, const double* v, int n) {
  int g = 0;
  for (; g + 16 <= n; g += 16) {
    double w[16];
#pragma unroll
    for (int u = 0; u < 16; u++) w[u] = v[g + u];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; u++) acc += w[u];
  }
  for (; g < n; g++) acc += v[g];
  return acc;
}

// One pass of the refinement callback over the rows at parameters h[0..7] (LDS): S.sc[slotS] = sum of squared
// residuals (groups of four, as cv::norm), S.sc[slotR] = max |residual|; with J also S.A8 = J^T J (mirrored) and
// S.v = J^T r (four interleaved partial sums).  Wave 0, all 64 lanes.  A point's terms in the tile:
// 0:Mx*ww 1:My*ww 2:ww 3:0 4:-Mx*ww*xi 5:-My*ww*xi 6:-Mx*ww*yi 7:-My*ww*yi 8:xi-mx 9:yi-my
//   x-row of J = {0,1,2,3,3,3,4,5}, y-row = {3,3,3,0,1,2,6,7}
__device__ __forceinline__ void lm_eval(SolveLds& S, int lane, const float* rows, int count, const double* h, bool withJ, int slotS,
                        int slotR) {
  double* T = S.T;
  const double h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7];
  int ei = 0, ej = 0;
  if (lane < 36) tri8(lane, ei, ej);
  const int jxi = ei < 3 ? ei : ei < 6 ? 3 : ei - 2, jxj = ej < 3 ? ej : ej < 6 ? 3 : ej - 2;
  const int jyi = ei < 3 ? 3 : ei < 6 ? ei - 3 : ei, jyj = ej < 3 ? 3 : ej < 6 ? ej - 3 : ej;
  const int vi = lane - 36;                      // lanes 36..43: J^T r entry vi
  const int vx = vi < 3 ? vi : vi < 6 ? 3 : vi - 2, vy = vi < 3 ? 3 : vi < 6 ? vi - 3 : vi;
  double s = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, nrm = 0, rmax = 0;
  // role of this lane in the sums: 1 = one J^T J entry (lanes 0..35), 2 = one J^T r entry (36..43), 3 = the squared norm (44)
  const int role = withJ && lane < 36 ? 1 : withJ && lane < 44 ? 2 : lane == 44 ? 3 : 0;
  const int pa = role == 1 ? jxi : role == 2 ? vx : 8, pb = role == 1 ? jxj : 8;
  const int pc = role == 1 ? jyi : role == 2 ? vy : 9, pd = role == 1 ? jyj : 9;
  // the rows of the NEXT tile are requested before this tile is worked on (a tile used to start with a full memory round trip)
  float4 rnext = lane < count ? *reinterpret_cast<const float4*>(rows + 4 * lane) : make_float4(0, 0, 0, 0);
  for (int c0 = 0; c0 < count; c0 += NL) {
    const int i = c0 + lane;
    double q0 = 0, q1 = 0;
    const float4 r = rnext;
    if (i + NL < count) rnext = *reinterpret_cast<const float4*>(rows + 4 * (i + NL));
    if (i < count) {
      const double Mx = r.x, My = r.y;
      double ww = h6 * Mx + h7 * My + 1.;
      ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
      const double xi = (h0 * Mx + h1 * My + h2) * ww;
      const double yi = (h3 * Mx + h4 * My + h5) * ww;
      const double rx = xi - r.z, ry = yi - r.w;
      double* t = T + lane * TS;
      t[8] = rx; t[9] = ry;
      if (withJ) {
        t[0] = Mx * ww; t[1] = My * ww; t[2] = ww; t[3] = 0.0;
        t[4] = -Mx * ww * xi; t[5] = -My * ww * xi; t[6] = -Mx * ww * yi; t[7] = -My * ww * yi;
      }
      rmax = fmax(rmax, fabs(rx));
      rmax = fmax(rmax, fabs(ry));
      q0 = rx * rx; q1 = ry * ry;
    }
    // the squared norm goes in groups of four like cv::norm: ((rx_j^2 + ry_j^2) + rx_{j+1}^2) + ry_{j+1}^2 per PAIR of
    // points.  The pair sums are formed in parallel (the even lane takes its neighbour's squares), one lane then adds the
    // <= 32 of them in order -- 48 instructions per tile instead of a 512-instruction walk by one lane
    const int cnt = min(NL, count - c0);
    {
      const double n0 = __shfl_down(q0, 1), n1 = __shfl_down(q1, 1);
      if (!(lane & 1)) {
        if (lane + 1 < cnt) S.P2[lane >> 1] = ((q0 + q1) + n0) + n1;
        else if (lane < cnt) { S.P2[NL / 2] = q0; S.P2[NL / 2 + 1] = q1; }      // odd last point: two separate additions
      }
    }
    WSYNC();
    // One instruction stream for the two kinds of matrix sums (round 3; with the norm they were three divergent
    // branches, i.e. three passes of the wave over the tile): every lane forms the same four products per PAIR of points
    // from its own four term indices (pa, pb, pc, pd) -- J^T J entry: (jxi, jxj, jyi, jyj), J^T r entry: (vx, 8, vy, 9) --
    // and only the additions differ: one running sum in point order / four interleaved partial sums.
    if (role == 1 || role == 2) {
      int j = 0;
      for (; j + 3 < cnt; j += 4) {               // two pairs of points per trip, their sixteen operands requested together
        const double* t = T + j * TS;
        double a[4], b[4], c[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = t[u * TS + pa]; b[u] = t[u * TS + pb]; c[u] = t[u * TS + pc]; d[u] = t[u * TS + pd]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
          const double P0 = a[u] * b[u], P1 = c[u] * d[u], P2 = a[u + 1] * b[u + 1], P3 = c[u + 1] * d[u + 1];
          if (role == 1) { s += P0; s += P1; s += P2; s += P3; }
          else { s0 += P0; s1 += P1; s2 += P2; s3 += P3; }
        }
      }
      for (; j + 1 < cnt; j += 2) {
        const double* t = T + j * TS;
        const double P0 = t[pa] * t[pb], P1 = t[pc] * t[pd], P2 = t[TS + pa] * t[TS + pb], P3 = t[TS + pc] * t[TS + pd];
        if (role == 1) { s += P0; s += P1; s += P2; s += P3; }
        else { s0 += P0; s1 += P1; s2 += P2; s3 += P3; }
      }
      if (j < cnt) {                              // only at the very end (tiles hold an even number of points)
        const double* t = T + j * TS;
        const double P0 = t[pa] * t[pb], P1 = t[pc] * t[pd];
        if (role == 1) { s += P0; s += P1; }
        else { s0 += P0; s0 += P1; }
      }
    } else if (role == 3) {
      nrm = add_in_order(nrm, S.P2, cnt >> 1);
      if (cnt & 1) { nrm += S.P2[NL / 2]; nrm += S.P2[NL / 2 + 1]; }
    }
    WSYNC();
  }
  for (int sft = 32; sft > 0; sft >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, sft));
  if (withJ && lane < 36) { S.A8[ei * 8 + ej] = s; S.A8[ej * 8 + ei] = s; }
  if (withJ && lane >= 36 && lane < 44) S.v[vi] = (s0 + s1 + s2 + s3) * 1.0;
  if (lane == 44) { S.sc[slotS] = nrm; S.sc[slotR] = rmax; }
  WSYNC();
}

// ---- tolerance mode (EVH_SOLVER_FAST): the same sums WITHOUT the operator's point order.  Every lane of wave 0 takes the
// points lane, lane + 64, ... and keeps its own partial sums in registers, a butterfly over the wave adds them: the 2N-long
// dependent chains of the exact form (8.4 cycles per addition, 0.7 M cycles per pair on the default detector list) become
// N / 64 independent steps and a 6-level tree.  J's rows are (t0 t1 t2 0 0 0 t4 t5) and (0 0 0 t0 t1 t2 t6 t7): 21 distinct
// entries of J^T J (the (3..5, 3..5) block repeats the (0..2, 0..2) block), 8 of J^T r, the squared norm, max |r|.
// Results differ from the exact form in the last digits (tests/test_gpu_parity.py::test_fast_solver_mode states the bars).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
  return v;
}
__device__ __forceinline__ void lm_eval_fast(SolveLds& S, int lane, const float* rows, int count, const double* h, bool withJ, int slotS,
                                             int slotR) {
  const double h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7];
  double xx[6] = {0, 0, 0, 0, 0, 0};        // sum t_i t_j, (i, j) = (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
  double xh[6] = {0, 0, 0, 0, 0, 0};        // sum t_i t4, t_i t5        (rows 0..2 against columns 6, 7)
  double yh[6] = {0, 0, 0, 0, 0, 0};        // sum t_i t6, t_i t7        (rows 3..5 against columns 6, 7)
  double hh[3] = {0, 0, 0};                 // (6,6) (6,7) (7,7)
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double nrm = 0, rmax = 0;
  for (int i = lane; i < count; i += NL) {
    const float4 r = *reinterpret_cast<const float4*>(rows + 4 * i);
    const double Mx = r.x, My = r.y;
    double ww = h6 * Mx + h7 * My + 1.;
    ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
    const double xi = (h0 * Mx + h1 * My + h2) * ww;
    const double yi = (h3 * Mx + h4 * My + h5) * ww;
    const double rx = xi - r.z, ry = yi - r.w;
    rmax = fmax(rmax, fmax(fabs(rx), fabs(ry)));
    nrm += rx * rx + ry * ry;
    if (withJ) {
      const double t0 = Mx * ww, t1 = My * ww, t2 = ww;
      const double t4 = -t0 * xi, t5 = -t1 * xi, t6 = -t0 * yi, t7 = -t1 * yi;
      xx[0] += t0 * t0; xx[1] += t0 * t1; xx[2] += t0 * t2; xx[3] += t1 * t1; xx[4] += t1 * t2; xx[5] += t2 * t2;
      xh[0] += t0 * t4; xh[1] += t0 * t5; xh[2] += t1 * t4; xh[3] += t1 * t5; xh[4] += t2 * t4; xh[5] += t2 * t5;
      yh[0] += t0 * t6; yh[1] += t0 * t7; yh[2] += t1 * t6; yh[3] += t1 * t7; yh[4] += t2 * t6; yh[5] += t2 * t7;
      hh[0] += t4 * t4 + t6 * t6; hh[1] += t4 * t5 + t6 * t7; hh[2] += t5 * t5 + t7 * t7;
      v[0] += t0 * rx; v[1] += t1 * rx; v[2] += t2 * rx; v[3] += t0 * ry; v[4] += t1 * ry; v[5] += t2 * ry;
      v[6] += t4 * rx + t6 * ry; v[7] += t5 * rx + t7 * ry;
    }
  }
  nrm = wave_sum_f64(nrm);
  for (int sft = 32; sft > 0; sft >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, sft));
  if (withJ) {
#pragma unroll
    for (int k = 0; k < 6; k++) { xx[k] = wave_sum_f64(xx[k]); xh[k] = wave_sum_f64(xh[k]); yh[k] = wave_sum_f64(yh[k]); }
#pragma unroll
    for (int k = 0; k < 3; k++) hh[k] = wave_sum_f64(hh[k]);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = wave_sum_f64(v[k]);
    if (lane == 0) {
      double* A = S.A8;
      for (int k = 0; k < 64; k++) A[k] = 0.0;
      const int ui[6] = {0, 0, 0, 1, 1, 2}, uj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int k = 0; k < 6; k++) {
        A[ui[k] * 8 + uj[k]] = xx[k]; A[uj[k] * 8 + ui[k]] = xx[k];
        A[(3 + ui[k]) * 8 + 3 + uj[k]] = xx[k]; A[(3 + uj[k]) * 8 + 3 + ui[k]] = xx[k];
      }
#pragma unroll
      for (int i = 0; i < 3; i++) {
        A[i * 8 + 6] = xh[2 * i]; A[6 * 8 + i] = xh[2 * i]; A[i * 8 + 7] = xh[2 * i + 1]; A[7 * 8 + i] = xh[2 * i + 1];
        A[(3 + i) * 8 + 6] = yh[2 * i]; A[6 * 8 + 3 + i] = yh[2 * i]; A[(3 + i) * 8 + 7] = yh[2 * i + 1]; A[7 * 8 + 3 + i] = yh[2 * i + 1];
      }
      A[6 * 8 + 6] = hh[0]; A[6 * 8 + 7] = hh[1]; A[7 * 8 + 6] = hh[1]; A[7 * 8 + 7] = hh[2];
#pragma unroll
      for (int k = 0; k < 8; k++) S.v[k] = v[k];
    }
  }
  if (lane == 0) { S.sc[slotS] = nrm; S.sc[slotR] = rmax; }
  WSYNC();
}

__device__ __forceinline__ double dot8(const double* a, const double* b) {
  double r = 0;
  r += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  r += a[4] * b[4] + a[5] * b[5] + a[6] * b[6] + a[7] * b[7];
  return r;
}

// Levenberg-Marquardt refinement of S.H[0..7] over `count` rows (<= 10 iterations). Returns iterations.  Wave 0.
// S.sc: 0 = S, 1 = rmax of the kept point, 2 = lambda, 3 = lc, 4 = nu, 5 = Sd, 6 = rmax of the trial point
// evalJ(): the pass WITH the Jacobian at S.x into S.sc[0], S.sc[1], S.A8, S.v -- lm_eval by this wave alone, or the
// four-wave form (lm_eval_mw) where the workgroup has helper waves.
// ---- tolerance mode of the LM refinement (EVH_SOLVER_FAST).  cv::solve(Ap, v, d, DECOMP_EIG) costs ~100 dependent Jacobi
// rotations of ~1 400 cycles each; the same 8x8 symmetric positive definite system by LDL^T in one lane is ~3 000 cycles.
// These systems are graded over ~14 orders of magnitude (raw pixel coordinates: smallest eigenvalue ~5 x the eigen-solve's
// truncation threshold), so along the weakest direction the two solvers differ in the leading digits of the step and, LM being
// cut after 10 iterations, H ends up to ~6e-4 px (corners) away from OpenCV's -- an opt-in mode (include/evhip.h); the RANSAC
// draw, the inlier masks and the refit are untouched.  Returns false when a pivot is not positive (the
// caller then takes the exact path), so nothing is ever solved with a factorisation that does not exist.
__device__ bool ldl8_factor(const double* A /*LDS, symmetric 8x8*/, double (&Lm)[28], double (&Dinv)[8]) {
  double Dd[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    double dj = A[j * 8 + j];
#pragma unroll
    for (int k = 0; k < j; k++) { const double l = Lm[(j * (j - 1)) / 2 + k]; dj -= l * l * Dd[k]; }
    if (!(dj > 0)) return false;
    Dd[j] = dj;
    Dinv[j] = 1.0 / dj;
#pragma unroll
    for (int i = j + 1; i < 8; i++) {
      double t = A[i * 8 + j];
#pragma unroll
      for (int k = 0; k < j; k++) t -= Lm[(i * (i - 1)) / 2 + k] * Lm[(j * (j - 1)) / 2 + k] * Dd[k];
      Lm[(i * (i - 1)) / 2 + j] = t * Dinv[j];
    }
  }
  return true;
}
__device__ void ldl8_solve(const double (&Lm)[28], const double (&Dinv)[8], const double (&b)[8], double (&x)[8]) {
  double y[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double t = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) t -= Lm[(i * (i - 1)) / 2 + k] * y[k];
    y[i] = t;
  }
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    double t = y[i] * Dinv[i];
#pragma unroll
    for (int k = i + 1; k < 8; k++) t -= Lm[(k * (k - 1)) / 2 + i] * x[k];
    x[i] = t;
  }
}
// lane 0: x = A^-1 b (b != null) or x[0] = max_i |(A^-1)_ii| (b == null); flag in ok (LDS int)
__device__ void fast_solve8(int lane, const double* A, const double* b, double* x, int* ok) {
  if (lane == 0) {
    double Lm[28], Dinv[8];
    bool good = ldl8_factor(A, Lm, Dinv);
    if (good) {
      if (b) {
        double bb[8], xx[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bb[i] = b[i];
        ldl8_solve(Lm, Dinv, bb, xx);
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = xx[i];
      } else {
        double mv = 0;
        for (int c = 0; c < 8; c++) {
          double e[8], col[8];
#pragma unroll
          for (int i = 0; i < 8; i++) e[i] = i == c ? 1.0 : 0.0;
          ldl8_solve(Lm, Dinv, e, col);
          double dc = 0;
#pragma unroll
          for (int i = 0; i < 8; i++) dc = i == c ? col[i] : dc;
          mv = fmax(mv, fabs(dc));
        }
        x[0] = mv;
      }
    }
    *ok = good ? 1 : 0;
  }
}

template <typename EvalJ, typename EvalN>
__device__ __forceinline__ int lm_refine(SolveLds& S, RowMat& M, int lane, const float* rows, int count,
                                         unsigned long long* prof, EvalJ evalJ, EvalN evalN /* residuals at S.xd -> sc[5], sc[6] */) {
  const int maxIters = 10;
  const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
  if (lane < 8) S.x[lane] = S.H[lane];
  WSYNC();
  unsigned long long pe = pf_now_if(prof);
  evalJ();
  pf_add(prof, PF_EVAL, pf_now_if(prof) - pe);
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
    bool solved = false;
    if (S.fast) {
      const unsigned long long pt0 = pf_now_if(prof);
      fast_solve8(lane, S.Ap, S.v, S.d, &S.ib[2]);
      WSYNC();
      solved = S.ib[2] != 0;
      pf_add(prof, PF_SOLVE8, pf_now_if(prof) - pt0);
    }
    if (!solved) eig_solve8_wave(M, lane, S.Ap, S.v, S.d, prof);
    if (lane < 8) S.xd[lane] = S.x[lane] - S.d[lane];
    WSYNC();
    pe = pf_now_if(prof);
    evalN();
    pf_add(prof, PF_EVAL, pf_now_if(prof) - pe);
    // trial residual -> Sd, gain ratio R; lane 0 decides, the (rare) inverse is done by the whole wave
    if (lane == 0) {
      const double Rlo = 0.25, Rhi = 0.75;
      double Sc = S.sc[0];
      double Sd = S.sc[5];
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
      S.sc[2] = lambda; S.sc[3] = lc; S.sc[4] = nu;
      S.ib[1] = need_inv;
    }
    WSYNC();
    if (S.ib[1]) {
      bool inverted = false;
      if (S.fast) {
        fast_solve8(lane, S.A8, nullptr, S.Inv, &S.ib[2]);      // S.Inv[0] = max |diag(A^-1)|
        WSYNC();
        inverted = S.ib[2] != 0;
        if (inverted && lane == 0) { const double mv = S.Inv[0]; for (int i = 0; i < 8; i++) S.Inv[i * 8 + i] = mv; }
        WSYNC();
      }
      if (!inverted) eig_solve8_wave(M, lane, S.A8, nullptr, S.Inv, prof);
      if (lane == 0) {
        double maxval = DBL_EPSILON;
        for (int i = 0; i < 8; i++) maxval = fmax(maxval, fabs(S.Inv[i * 8 + i]));
        const double lam = 1. / maxval;
        S.sc[3] = lam;                       // lc
        S.sc[2] = lam * (S.sc[4] * 0.5);     // lambda = lc; nu *= 0.5; lambda *= nu
      }
      WSYNC();
    }
    const bool accepted = S.sc[5] < S.sc[0];
    WSYNC();
    if (accepted) {
      if (lane < 8) { const double t = S.x[lane]; S.x[lane] = S.xd[lane]; S.xd[lane] = t; }
      WSYNC();
      pe = pf_now_if(prof);
      evalJ();                                          // residuals / Jacobian at the accepted point (S = Sd again)
      pf_add(prof, PF_EVAL, pf_now_if(prof) - pe);
    }
    iter++;
    // norm(r, INF) of the kept residual, norm(d, INF)
    const double rmax = S.sc[1];
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

// LANES: the hypothesis phase gives every LANE its own hypothesis (jacobi_lanes9) instead of every 16-lane row -- the
// fixed-iteration mode, where thousands of hypotheses are evaluated and throughput counts; 23 KB of LDS per wave.
template <int NW, bool LANES>
struct alignas(16) BlockLds {
  RowMat m[LANES ? 1 : NW][NG];
  SolveLds s;
  int hyp[2][LANES ? NW * NL : NW * NG];    // per hypothesis of a chunk: valid << 31 | model << 30 | inlier count (double-buffered)
  double Hsup[9], Hprev[9], Hcur[9];
  int have_prev, gate;
  union {
    struct { unsigned hist[HB]; } h;                       // static filter: population of every displacement bin
    double lmat[LANES ? NW * LM_ELEMS * NL : 1];            // the per-lane matrices (dead when the static filter runs)
  } u;
  unsigned long long red[NW];
};

// ---- lm_eval with the Jacobian, all four waves of the workgroup (round 3).  The sums of J^T J and J^T r are strictly
// sequential over the points (the operator's order), but only their ADDITIONS are: the products are formed ahead by
// another wave.  And most of the products are structural zeros: the x-row of J is (t0 t1 t2 0 0 0 t4 t5), the y-row
// (0 0 0 t0 t1 t2 t6 t7), so of the 36 entries (i <= j) of J^T J  9 have no nonzero product at all, 24 have ONE per
// point (x or y) and only (6,6), (6,7), (7,7) have both.  Adding +-0.0 to a running sum that started at +0.0 never
// changes a bit of it (x + +-0 = x for x != 0, and +0 + -0 = +0), so the zero products are neither formed nor added:
// 46 products per point instead of 88, 16 dependent additions per step instead of 32 for the 24 single entries.
// (Finite terms assumed: 0 * inf would be NaN.  The terms are products of the rows, 1/w and the current parameters;
// parameters that large have already failed the residual tests.)
// Steps of 16 points, one workgroup barrier per step, everything double-buffered:
//   wave 1  terms of the next 64-point tile (every fourth step; its rows are requested one step ahead), max |r| and the
//           pair sums of the squared residuals of that tile
//   wave 2  the 46 products (lane = product) for the 16 points of the NEXT step: all lanes read the same point's terms
//           (10 words: no bank conflict) and write prod[point][46]: 24 singles, then (x, y) of 11 pair entries
//   wave 0  this step's additions of the 24 single entries (lane = entry, one running sum in point order) and, every
//           fourth step, the squared norm (lane 44)
//   wave 3  this step's additions of the pair entries: (6,6), (6,7), (7,7) of J^T J (s += x; s += y) and the 8 entries of
//           J^T r (four interleaved partial sums; their structural zeros are formed and added like any other value)
// Same operations in the same order on every sum.  LDS: the buffers live in what is dead during the refinement -- the
// hypothesis matrices of the other rows / waves and the static filter's histogram.
#define MW_SUB 16                        // points per step
#define MW_NS 24                         // single entries of J^T J
#define MW_NP 11                         // pair entries: 3 of J^T J + 8 of J^T r
#define MW_NPR (MW_NS + 2 * MW_NP)       // products per point
#define MW_PSTR (MW_SUB + 2)              // doubles between two products' rows: [product][point], 144 bytes -> ds_*_b128 of
                                         // neighbouring lanes fall on different bank slots
#define MW_PROD (MW_PSTR * MW_NPR)       // doubles per product buffer
#define MW_TSTR (NL + 2)                 // doubles between two terms' rows of a tile: [term][point], 528 bytes (same reason)
typedef double mw_d2 __attribute__((ext_vector_type(2)));
#ifndef MW_MIN_ROWS
#define MW_MIN_ROWS 512                   // fewer inlier rows: wave 0 alone (measured break-even ~300 rows)
#endif
__device__ __forceinline__ int mw_jx(int k) { return k < 3 ? k : k < 6 ? 3 : k - 2; }   // term index of J's x-row, column k
__device__ __forceinline__ int mw_jy(int k) { return k < 3 ? 3 : k < 6 ? k - 3 : k; }   // ... y-row
// the `which`-th entry (i <= j, tri8 order) of the given kind: 1 = x only, 2 = y only (both kinds enumerated together as
// "single"), 0 = none; returns false when there is no such entry
__device__ __forceinline__ bool mw_entry(int which, bool single, int& ei, int& ej, bool& yrow) {
  int cnt = 0;
  bool found = false;
  for (int e = 0; e < 36; e++) {
    int i, j;
    tri8(e, i, j);
    const bool cx = !(i >= 3 && i < 6) && !(j >= 3 && j < 6), cy = i >= 3 && j >= 3;
    const bool is_single = cx != cy, is_none = !cx && !cy;
    if (single ? is_single : is_none) {
      if (cnt == which) { ei = i; ej = j; yrow = cy; found = true; }
      cnt++;
    }
  }
  return found;
}
template <int NW, bool LANES>
__device__ __forceinline__ void lm_eval_mw(BlockLds<NW, LANES>& B, int wave, int lane, const float* rows, int count,
                                           unsigned long long* prof) {
  static_assert(NW == 4 && !LANES, "helper waves: the four-wave row form only");
  static_assert(sizeof(RowMat) * (NW * NG - 1) >= sizeof(double) * (MW_PROD + NL * TS) + 16, "buffer 0 + second tile");
  static_assert(sizeof(B.u) >= sizeof(double) * MW_PROD + 16, "buffer 1");
  static_assert(NL * TS >= 10 * MW_TSTR, "a tile of terms, term-major");
  static_assert(alignof(BlockLds<NW, LANES>) >= 16, "16-byte LDS accesses below");
  SolveLds& S = B.s;
  // (selects, not arrays of pointers: an indexed pointer array loses the LDS address space and turns every access into
  // a FLAT instruction -- measured 3x slower)
  // every buffer starts on a 16-byte boundary (the struct is 16-byte aligned; an odd multiple of 8 is skipped by one double)
  typedef BlockLds<NW, LANES> BL;
  double* const prod0 = reinterpret_cast<double*>(&B.m[0][1]) + ((offsetof(BL, m) + sizeof(RowMat)) % 16 ? 1 : 0);
  double* const prod1 = reinterpret_cast<double*>(&B.u) + (offsetof(BL, u) % 16 ? 1 : 0);
  double* const Tb0 = S.T + ((offsetof(BL, s) + offsetof(SolveLds, T)) % 16 ? 1 : 0);
  double* const Tb1 = prod0 + MW_PROD;
  static_assert(MW_PROD % 2 == 0 && MW_PSTR % 2 == 0 && MW_TSTR % 2 == 0 && MW_SUB % 2 == 0, "16-byte rows");
#define MW_PRODBUF(i) (((i) & 1) ? prod1 : prod0)
#define MW_TERMBUF(i) (((i) & 1) ? Tb1 : Tb0)
  const int nsub = (count + MW_SUB - 1) / MW_SUB;
  double h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0;
  if (wave == 1) { h0 = S.x[0]; h1 = S.x[1]; h2 = S.x[2]; h3 = S.x[3]; h4 = S.x[4]; h5 = S.x[5]; h6 = S.x[6]; h7 = S.x[7]; }
  // wave 2: the two term indices of this lane's product.  wave 0: where this lane's sum goes in A8 (lanes 24..32: the
  // entries that are zero by structure).  wave 3: lanes 0..2 = (6,6), (6,7), (7,7); lanes 3..10 = J^T r entry lane - 3.
  int ia = 3, ib = 3, a8i = -1, a8j = -1;
  if (wave == 2) {
    if (lane < MW_NS) {
      int ei = 0, ej = 0; bool yrow = false;
      mw_entry(lane, true, ei, ej, yrow);
      ia = yrow ? mw_jy(ei) : mw_jx(ei); ib = yrow ? mw_jy(ej) : mw_jx(ej);
    } else if (lane < MW_NPR) {
      const int u = (lane - MW_NS) >> 1;
      const bool yrow = (lane - MW_NS) & 1;
      if (u < 3) {
        const int ei = u == 2 ? 7 : 6, ej = u == 0 ? 6 : 7;
        ia = yrow ? mw_jy(ei) : mw_jx(ei); ib = yrow ? mw_jy(ej) : mw_jx(ej);
      } else {
        ia = yrow ? mw_jy(u - 3) : mw_jx(u - 3); ib = yrow ? 9 : 8;
      }
    }
  } else if (wave == 0) {
    bool yrow = false;
    if (lane < MW_NS) mw_entry(lane, true, a8i, a8j, yrow);
    else if (lane < MW_NS + 9) mw_entry(lane - MW_NS, false, a8i, a8j, yrow);
  } else if (wave == 3 && lane < 3) {
    a8i = lane == 2 ? 7 : 6; a8j = lane == 0 ? 6 : 7;
  }
  double s = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, nrm = 0, rmax = 0;
  float4 rnext = make_float4(0, 0, 0, 0);
  unsigned long long pf_busy = 0, pf_wait = 0;       // cycle accounting: summed in registers, one atomic per pass
  for (int k = -2; k < nsub; k++) {
    const unsigned long long pm0 = pf_now_if(prof);
    if (wave == 1) {
      // rows of the tile whose terms are due at the next step (or now, for the first tile)
      if (k == -2 || ((k + 3) & 3) == 0) {
        const int i = ((k + 3) >> 2) * NL + lane;
        if (i < count) rnext = *reinterpret_cast<const float4*>(rows + 4 * i);
      }
      if (((k + 2) & 3) == 0 && ((k + 2) >> 2) * NL < count) {
        const int n = (k + 2) >> 2, c0 = n * NL, i = c0 + lane;
        double q0 = 0, q1 = 0;
        if (i < count) {
          const float4 r = rnext;
          const double Mx = r.x, My = r.y;
          double ww = h6 * Mx + h7 * My + 1.;
          ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
          const double xi = (h0 * Mx + h1 * My + h2) * ww;
          const double yi = (h3 * Mx + h4 * My + h5) * ww;
          const double rx = xi - r.z, ry = yi - r.w;
          double* t = MW_TERMBUF(n) + lane;                          // term j of this point at t[j * MW_TSTR]
          t[8 * MW_TSTR] = rx; t[9 * MW_TSTR] = ry;
          t[0] = Mx * ww; t[MW_TSTR] = My * ww; t[2 * MW_TSTR] = ww; t[3 * MW_TSTR] = 0.0;
          t[4 * MW_TSTR] = -Mx * ww * xi; t[5 * MW_TSTR] = -My * ww * xi; t[6 * MW_TSTR] = -Mx * ww * yi; t[7 * MW_TSTR] = -My * ww * yi;
          rmax = fmax(rmax, fabs(rx));
          rmax = fmax(rmax, fabs(ry));
          q0 = rx * rx; q1 = ry * ry;
        }
        const int cnt = min(NL, count - c0);
        const double n0 = __shfl_down(q0, 1), n1 = __shfl_down(q1, 1);
        if (!(lane & 1)) {
          if (lane + 1 < cnt) S.P2[lane >> 1] = ((q0 + q1) + n0) + n1;
          else if (lane < cnt) { S.P2[NL / 2] = q0; S.P2[NL / 2 + 1] = q1; }
        }
      }
    } else if (wave == 0) {
      if (k >= 0 && lane < MW_NS) {
        const double* r = MW_PRODBUF(k) + lane * MW_PSTR;               // this entry's products of the step's 16 points
        const int cnt = min(MW_SUB, count - k * MW_SUB);
        if (cnt == MW_SUB) {
          // all words requested before the first addition (left alone the compiler waits for every read in turn); two
          // points per ds_read_b128 (a quarter of the LDS cycles of the 8-byte reads: this loop was LDS-issue bound)
          mw_d2 v[MW_SUB / 2];
#pragma unroll
          for (int q = 0; q < MW_SUB / 2; q++) v[q] = *reinterpret_cast<const mw_d2*>(r + 2 * q);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < MW_SUB / 2; q++) { s += v[q].x; s += v[q].y; }
        } else {
          for (int q = 0; q < cnt; q++) s += r[q];
        }
      }
      // the squared norm of tile n (pair sums left by wave 1 one step ago), in order, by a lane with nothing else to do
      if (((k + 1) & 3) == 0 && ((k + 1) >> 2) * NL < count && lane == 44) {
        const int cnt = min(NL, count - ((k + 1) >> 2) * NL);
        nrm = add_in_order(nrm, S.P2, cnt >> 1);
        if (cnt & 1) { nrm += S.P2[NL / 2]; nrm += S.P2[NL / 2 + 1]; }
      }
    } else if (wave == 3) {
      if (k >= 0 && lane < MW_NP) {
        const double* rx = MW_PRODBUF(k) + (MW_NS + 2 * lane) * MW_PSTR;  // x products of the 16 points; y: the next row
        const double* ry = rx + MW_PSTR;
        const int cnt = min(MW_SUB, count - k * MW_SUB);
        if (cnt == MW_SUB) {
          mw_d2 vx[MW_SUB / 2], vy[MW_SUB / 2];
#pragma unroll
          for (int q = 0; q < MW_SUB / 2; q++) { vx[q] = *reinterpret_cast<const mw_d2*>(rx + 2 * q); vy[q] = *reinterpret_cast<const mw_d2*>(ry + 2 * q); }
          __builtin_amdgcn_sched_barrier(0);
          if (lane < 3) {
#pragma unroll
            for (int q = 0; q < MW_SUB / 2; q++) { s += vx[q].x; s += vy[q].x; s += vx[q].y; s += vy[q].y; }
          } else {
#pragma unroll
            for (int q = 0; q < MW_SUB / 2; q++) { s0 += vx[q].x; s1 += vy[q].x; s2 += vx[q].y; s3 += vy[q].y; }
          }
        } else {
          int q = 0;
          for (; q + 1 < cnt; q += 2) {
            const double a = rx[q], b = ry[q], c = rx[q + 1], d = ry[q + 1];
            if (lane < 3) { s += a; s += b; s += c; s += d; }
            else { s0 += a; s1 += b; s2 += c; s3 += d; }
          }
          if (q < cnt) {
            const double a = rx[q], b = ry[q];
            if (lane < 3) { s += a; s += b; }
            else { s0 += a; s0 += b; }
          }
        }
      }
    } else {
      const int sub = k + 1;
      if (sub >= 0 && sub < nsub && lane < MW_NPR) {
        const double* t = MW_TERMBUF(sub >> 2) + (sub & 3) * MW_SUB;     // term j of the step's point q at t[j * MW_TSTR + q]
        double* out = MW_PRODBUF(sub) + lane * MW_PSTR;
        const int cnt = min(MW_SUB, count - sub * MW_SUB);
        if (cnt == MW_SUB) {
          mw_d2 va[MW_SUB / 2], vb[MW_SUB / 2];
#pragma unroll
          for (int q = 0; q < MW_SUB / 2; q++) {
            va[q] = *reinterpret_cast<const mw_d2*>(t + ia * MW_TSTR + 2 * q);
            vb[q] = *reinterpret_cast<const mw_d2*>(t + ib * MW_TSTR + 2 * q);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < MW_SUB / 2; q++) {
            mw_d2 pr; pr.x = va[q].x * vb[q].x; pr.y = va[q].y * vb[q].y;
            *reinterpret_cast<mw_d2*>(out + 2 * q) = pr;
          }
        } else {
          for (int q = 0; q < cnt; q++) out[q] = t[ia * MW_TSTR + q] * t[ib * MW_TSTR + q];
        }
      }
    }
    const unsigned long long pm1 = pf_now_if(prof);
    __syncthreads();
    pf_busy += pm1 - pm0; pf_wait += pf_now_if(prof) - pm1;
  }
  pf_add_wave(prof, PF_MW_W0 + wave, pf_busy);
  if (wave == 0) { pf_add(prof, PF_MW_WAIT, pf_wait); pf_add(prof, PF_MW_STEPS, nsub + 2); }
  if (wave == 0 && a8i >= 0) { S.A8[a8i * 8 + a8j] = s; S.A8[a8j * 8 + a8i] = s; }      // (s = 0 for the structural zeros)
  if (wave == 0 && lane == 44) S.sc[0] = nrm;
  if (wave == 3 && lane < 3) { S.A8[a8i * 8 + a8j] = s; S.A8[a8j * 8 + a8i] = s; }
  if (wave == 3 && lane >= 3 && lane < MW_NP) S.v[lane - 3] = (s0 + s1 + s2 + s3) * 1.0;
  if (wave == 1) {
    for (int sft = 32; sft > 0; sft >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, sft));
    if (lane == 44) S.sc[1] = rmax;
  }
  __syncthreads();
#undef MW_PRODBUF
#undef MW_TERMBUF
}

// ---- the pass WITHOUT the Jacobian (the trial point of every LM iteration: squared norm and max |r| only) on two waves:
// wave 1 computes the residuals of the next 64-point tile and their pair sums (rows requested a tile ahead), wave 0's
// lane 44 adds the pair sums of the present tile in order; one workgroup barrier per tile, two pair-sum buffers.
template <int NW, bool LANES>
__device__ __forceinline__ void lm_eval_noj_mw(BlockLds<NW, LANES>& B, int wave, int lane, const float* rows, int count) {
  static_assert(NW == 4 && !LANES, "helper waves: the four-wave row form only");
  SolveLds& S = B.s;
  double* const Pa = S.P2;
  double* const Pb = reinterpret_cast<double*>(&B.m[0][1]);         // dead during the refinement (see lm_eval_mw)
#define MW_P2BUF(i) (((i) & 1) ? Pb : Pa)
  const int ntile = (count + NL - 1) / NL;
  double h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0;
  if (wave == 1) { h0 = S.xd[0]; h1 = S.xd[1]; h2 = S.xd[2]; h3 = S.xd[3]; h4 = S.xd[4]; h5 = S.xd[5]; h6 = S.xd[6]; h7 = S.xd[7]; }
  double nrm = 0, rmax = 0;
  float4 rnext = make_float4(0, 0, 0, 0);
  if (wave == 1 && lane < count) rnext = *reinterpret_cast<const float4*>(rows + 4 * lane);
  for (int n = -1; n < ntile; n++) {
    if (wave == 1 && n + 1 < ntile) {
      const int c0 = (n + 1) * NL, i = c0 + lane;
      const float4 r = rnext;
      if (i + NL < count) rnext = *reinterpret_cast<const float4*>(rows + 4 * (i + NL));
      double q0 = 0, q1 = 0;
      if (i < count) {
        const double Mx = r.x, My = r.y;
        double ww = h6 * Mx + h7 * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h0 * Mx + h1 * My + h2) * ww;
        const double yi = (h3 * Mx + h4 * My + h5) * ww;
        const double rx = xi - r.z, ry = yi - r.w;
        rmax = fmax(rmax, fabs(rx));
        rmax = fmax(rmax, fabs(ry));
        q0 = rx * rx; q1 = ry * ry;
      }
      const int cnt = min(NL, count - c0);
      const double n0 = __shfl_down(q0, 1), n1 = __shfl_down(q1, 1);
      double* P = MW_P2BUF(n + 1);
      if (!(lane & 1)) {
        if (lane + 1 < cnt) P[lane >> 1] = ((q0 + q1) + n0) + n1;
        else if (lane < cnt) { P[NL / 2] = q0; P[NL / 2 + 1] = q1; }
      }
    } else if (wave == 0 && n >= 0 && lane == 44) {
      const int cnt = min(NL, count - n * NL);
      const double* P = MW_P2BUF(n);
      nrm = add_in_order(nrm, P, cnt >> 1);
      if (cnt & 1) { nrm += P[NL / 2]; nrm += P[NL / 2 + 1]; }
    }
    __syncthreads();
  }
#undef MW_P2BUF
  if (wave == 0 && lane == 44) S.sc[5] = nrm;
  if (wave == 1) {
    for (int sft = 32; sft > 0; sft >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, sft));
    if (lane == 44) S.sc[6] = rmax;
  }
  __syncthreads();
}

// The helper waves (1..NW-1) of a workgroup while wave 0 refines: they sleep at the workgroup barrier until wave 0 posts a
// command in S.ib[4] (1: one lm_eval_mw pass over S.ib[5] rows of `crow`; 2: one lm_eval_noj_mw pass; 0: done).
template <int NW, bool LANES>
__device__ __forceinline__ void lm_helper_loop(BlockLds<NW, LANES>& B, int wave, int lane, const float* crow, unsigned long long* prof) {
  if constexpr (NW == 4 && !LANES) {
    for (;;) {
      __syncthreads();
      const int cmd = B.s.ib[4];
      if (cmd == 0) break;
      if (cmd == 1) lm_eval_mw<NW, LANES>(B, wave, lane, crow, B.s.ib[5], prof);
      else lm_eval_noj_mw<NW, LANES>(B, wave, lane, crow, B.s.ib[5]);
    }
  }
}

// Hypotheses of one RANSAC call that were evaluated ahead of it by other workgroups (k_scan_hyp: the fixed-iteration
// stream scan spreads the 2000 samples of a pair over ~140 workgroups; only the replay below is serial).
struct ScanPre {
  const int* hyp;                  // [count] valid << 31 | model << 30 | inlier count, in sample order
  const double* H;                 // [count][9] the models
  int count;                       // < 10000
  unsigned long long rng_after;    // generator state after `count` quadruples
};

// ---- cv2.findHomography(a, b, RANSAC, thr) on `n` rows; result in B.s.H (LDS), mask[n] in global.  All NW*64
// threads of the workgroup call this; the return value is uniform.  scratch: crow = float rows [n][4] for the
// compacted inliers.  Ends with a workgroup barrier.
template <int NW, bool LANES>
__device__ __forceinline__ bool find_homography_block(BlockLds<NW, LANES>& B, const float* rows, int n, double thr, int maxItersArg, double conf,
                                      int force_max, uint8_t* mask, float* crow, int* info, unsigned long long* prof,
                                      double* lane_v /* LANES: this workgroup's NW * 81 * 64 doubles of global scratch */,
                                      const ScanPre* pre = nullptr /* force_max only */) {
  const unsigned long long pf0 = pf_now_if(prof);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, gl = lane & 15;
  SolveLds& S = B.s;
  if (info && tid < 3) info[tid] = 0;
  for (int i = tid; i < n; i += NW * NL) mask[i] = 0;
  if (thr <= 0) thr = 3;
  if (n < 4) { __threadfence_block(); __syncthreads(); return false; }
  if (n == 4) {
    if (wave == 0) {
      const bool ok = dlt_rows(S, B.m[0][0], lane, rows, 4, S.H);
      if (lane == 0) S.ib[2] = ok ? 1 : 0;
      if (ok && lane < 4) mask[lane] = 1;
      if (ok && info && lane == 0) info[1] = 4;
    }
    __threadfence_block();
    __syncthreads();
    return S.ib[2] != 0;
  }
  const float t = (float)(thr * thr);
  constexpr int HC = LANES ? NW * NL : NW * NG;             // hypotheses per chunk
  const int myh = LANES ? wave * NL + lane : wave * NG + row;
  Rng rng;
  const FastMod fm((unsigned)n);
  int niters = max(maxItersArg, 1), maxGood = 0, iter = 0, run = 0, chunk = 0;
  bool stop = false;
  if (pre) {
    // Replay of the samples evaluated ahead, 64 at a time.  With the iteration count fixed the serial rule of the
    // chunk loop below reduces to: the samples that count are the valid ones among the first `niters` valid ones, the
    // winner is the FIRST of them with the largest inlier count (> 3); `run` = trailing rejected samples.
    int lastvalid = -1;
    unsigned long long bestkey = 0ull;
    for (int c0 = 0; c0 < pre->count; c0 += NL) {
      const int h = c0 + lane;
      const unsigned e = h < pre->count ? (unsigned)pre->hyp[h] : 0u;
      const bool v = (e >> 31) != 0u;
      const unsigned long long vb = __ballot(v);
      const int before = iter + __popcll(vb & ((1ull << lane) - 1ull));
      if (v && before < niters && ((e >> 30) & 1u) && (e & 0x3FFFFFFFu) > 3u) {
        const unsigned long long key = ((unsigned long long)(e & 0x3FFFFFFFu) << 32) | (0xFFFFFFFFu - (unsigned)h);
        bestkey = key > bestkey ? key : bestkey;
      }
      if (__ballot(v && before >= niters) != 0ull) stop = true;
      iter = min(iter + (int)__popcll(vb), niters);
      if (vb) lastvalid = c0 + 63 - __clzll((long long)vb);
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
      const unsigned long long o = __shfl_xor(bestkey, sft);
      bestkey = o > bestkey ? o : bestkey;
    }
    run = pre->count - 1 - lastvalid;
    if (bestkey) {
      maxGood = (int)(bestkey >> 32);
      const int bh = (int)(0xFFFFFFFFu - (unsigned)bestkey);
      if (tid < 9) S.bestH[tid] = pre->H[9 * (int64_t)bh + tid];
    }
    rng.state = pre->rng_after;
  }
  while (!stop && iter < niters) {
    // every lane advances the generator identically through HC quadruples; the lanes of hypothesis h keep quadruple #h
    const unsigned long long pr0 = pf_now_if(prof);
    int my[4] = {0, 1, 2, 3};
    for (int h = 0; h < HC; h++) {
      // four distinct indices, a duplicate is redrawn (getSubset's inner loop, straight-line)
      const int q0 = (int)fm.mod(rng.next());
      int q1, q2, q3;
      do q1 = (int)fm.mod(rng.next()); while (q1 == q0);
      do q2 = (int)fm.mod(rng.next()); while (q2 == q0 || q2 == q1);
      do q3 = (int)fm.mod(rng.next()); while (q3 == q0 || q3 == q1 || q3 == q2);
      if (h == myh) { my[0] = q0; my[1] = q1; my[2] = q2; my[3] = q3; }
    }
    float Mx[4], My[4], mx[4], my_[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float4 r = *reinterpret_cast<const float4*>(rows + 4 * my[i]);
      Mx[i] = r.x; My[i] = r.y; mx[i] = r.z; my_[i] = r.w;
    }
    const bool valid = check_subset4(Mx, My, mx, my_);
    double H[9];
    const unsigned long long ps0 = pf_now_if(prof);
    pf_add(prof, PF_RNG, ps0 - pr0);
    bool ok;
    if constexpr (LANES) ok = dlt4_lane(B.u.lmat + (wave * LM_ELEMS * NL + lane), lane_v + (wave * LV_ELEMS * NL + lane), valid, Mx, My, mx, my_, H);
    else ok = dlt4_rows(B.m[wave][row], lane, valid, Mx, My, mx, my_, H);
    const unsigned long long ps1 = pf_now_if(prof);
    pf_add(prof, PF_SETUP, ps1 - ps0);
    int good = 0;
    {
      float Hf[8];
#pragma unroll
      for (int i = 0; i < 8; i++) Hf[i] = ok ? (float)H[i] : 0.f;
      if constexpr (LANES) {   // every lane walks all points (same addresses across the wave: one transaction each)
        if (__ballot(ok) != 0ull) {
#pragma unroll 4
          for (int i = 0; i < n; i++) {
            const float4 r = *reinterpret_cast<const float4*>(rows + 4 * i);
            good += is_inlier(Hf, r.x, r.y, r.z, r.w, t) ? 1 : 0;
          }
        }
        if (!ok) good = 0;
      } else {
        if (ok) {   // the 16 lanes of the row split the points; integer count, order-free
#pragma unroll 4
          for (int i = gl; i < n; i += GL) {
            const float4 r = *reinterpret_cast<const float4*>(rows + 4 * i);
            good += is_inlier(Hf, r.x, r.y, r.z, r.w, t) ? 1 : 0;
          }
        }
        good = rsum16(good);
      }
    }
    int* hyp = B.hyp[chunk & 1];
    if (LANES || gl == 0) hyp[myh] = (valid ? 0x80000000u : 0u) | (ok ? 0x40000000u : 0u) | (unsigned)good;
    const unsigned long long ps2 = pf_now_if(prof);
    pf_add(prof, PF_COUNT, ps2 - ps1);
    __syncthreads();
    const unsigned long long ps3 = pf_now_if(prof);
    pf_add(prof, PF_BARRIER, ps3 - ps2);
    // sequential replay in sample order (every thread, identically); the chunk's entries are fetched once into
    // registers (lane j holds entries j, j + 64, ...) and walked with v_readlane instead of one LDS round trip each
    constexpr int NSEG = (HC + NL - 1) / NL;
    unsigned er[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; sg++) er[sg] = sg * NL + lane < HC ? (unsigned)hyp[sg * NL + lane] : 0u;
    int best_h = -1;
#pragma unroll
    for (int sg = 0; sg < NSEG; sg++)
    for (int hj = 0; hj < (HC - sg * NL < NL ? HC - sg * NL : NL) && !stop; hj++) {
      const int h = sg * NL + hj;
      const unsigned e = (unsigned)__builtin_amdgcn_readlane((int)er[sg], hj);
      if (!(e >> 31)) {  // rejected sample: counts towards the 10000-attempt bound of one draw
        if (++run >= 10000) { stop = true; break; }
        continue;
      }
      run = 0;
      if (iter >= niters) { stop = true; break; }
      iter++;
      if (!((e >> 30) & 1u)) continue;
      const int g = (int)(e & 0x3FFFFFFFu);
      if (g > max(maxGood, 3)) {
        maxGood = g;
        best_h = h;
        if (!force_max) niters = update_num_iters(conf, (double)(n - g) / n, 4, niters);
      }
    }
    if (best_h == myh && (LANES || gl == 0)) {
#pragma unroll
      for (int i = 0; i < 9; i++) S.bestH[i] = H[i];
    }
    chunk++;
    pf_add(prof, PF_REPLAY, pf_now_if(prof) - ps3);
  }
  __threadfence_block();
  __syncthreads();
  const unsigned long long pf1 = pf_now_if(prof);
  pf_add(prof, PF_CALLS, 1); pf_add(prof, PF_HYP, pf1 - pf0); pf_add(prof, PF_CHUNKS, chunk);
  if (info && tid == 0) { info[0] = iter; info[1] = maxGood; }
  if (maxGood <= 0) return false;
  if (wave == 0) {
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
      const unsigned long long mm = __ballot(in);
      if (in) *reinterpret_cast<float4*>(crow + 4 * (ni + __popcll(mm & ((1ull << lane) - 1ull)))) = r;
      ni += __popcll(mm);
    }
    __threadfence_block();
    WSYNC();
    if (lane < 9) S.H[lane] = S.bestH[lane];
    WSYNC();
    const unsigned long long pf2 = pf_now_if(prof);
    pf_add(prof, PF_COMPACT, pf2 - pf1);
    if (ni > 0) {
      dlt_rows(S, B.m[0][0], lane, crow, ni, S.H, prof);  // keeps the RANSAC model when the refit is degenerate
      const unsigned long long pf3 = pf_now_if(prof);
      pf_add(prof, PF_REFIT, pf3 - pf2);
      int it;
      if constexpr (NW == 4 && !LANES) {
        // the passes with the Jacobian run on all four waves: post the command, meet the helpers at the barrier
        // (from MW_MIN_ROWS rows on: below that the extra barriers cost what the helpers save)
        it = lm_refine(S, B.m[0][0], lane, crow, ni, prof, [&]() {
          if (S.fast) { lm_eval_fast(S, lane, crow, ni, S.x, true, 0, 1); return; }
          if (ni < MW_MIN_ROWS) { lm_eval(S, lane, crow, ni, S.x, true, 0, 1); return; }
          if (lane == 0) { S.ib[4] = 1; S.ib[5] = ni; }
          __threadfence_block();
          __syncthreads();
          lm_eval_mw<NW, LANES>(B, 0, lane, crow, ni, prof);
        }, [&]() {
          if (S.fast) { lm_eval_fast(S, lane, crow, ni, S.xd, false, 5, 6); return; }
          if (ni < MW_MIN_ROWS) { lm_eval(S, lane, crow, ni, S.xd, false, 5, 6); return; }
          if (lane == 0) { S.ib[4] = 2; S.ib[5] = ni; }
          __threadfence_block();
          __syncthreads();
          lm_eval_noj_mw<NW, LANES>(B, 0, lane, crow, ni);
        });
      } else {
        it = lm_refine(S, B.m[0][0], lane, crow, ni, prof,
                       [&]() { if (S.fast) lm_eval_fast(S, lane, crow, ni, S.x, true, 0, 1); else lm_eval(S, lane, crow, ni, S.x, true, 0, 1); },
                       [&]() { if (S.fast) lm_eval_fast(S, lane, crow, ni, S.xd, false, 5, 6); else lm_eval(S, lane, crow, ni, S.xd, false, 5, 6); });
      }
      pf_add(prof, PF_LM, pf_now_if(prof) - pf3); pf_add(prof, PF_LM_ITERS, it);
      if (info && lane == 0) info[2] = it;
    }
    if constexpr (NW == 4 && !LANES) {
      if (lane == 0) S.ib[4] = 0;                          // release the helpers
      __threadfence_block();
      __syncthreads();
    }
  } else {
    lm_helper_loop<NW, LANES>(B, wave, lane, crow, prof);
  }
  __threadfence_block();
  __syncthreads();
  pf_add(prof, PF_TOTAL, pf_now_if(prof) - pf0);
  return true;
}

// np.dot(H, (x, y, 1)) in the summation order pinned by the reference-glue fixtures: fma(h0, x, h1*y) + h2
__device__ __forceinline__ void hdot(const double* H, double x, double y, double* tx, double* ty, double* tw) {
  *tx = fma(H[0], x, H[1] * y) + H[2];
  *ty = fma(H[3], x, H[4] * y) + H[5];
  *tw = fma(H[6], x, H[7] * y) + H[8];
}

// find_point_displacement + get_largest_group_points; rbin = int scratch [n] (global); returns kept count (uniform).
// All threads of the workgroup; ends with a workgroup barrier.
template <int NW, bool LANES>
__device__ __forceinline__ int static_filter_block(BlockLds<NW, LANES>& B, const double* H /*LDS*/, const float* rows, int n, int* rbin,
                                   float* out) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NT = NW * NL;
  for (int i = tid; i < HB; i += NT) B.u.h.hist[i] = 0u;
  __syncthreads();
  int big = 0;
  for (int i = tid; i < n; i += NT) {
    double tx, ty, tw;
    hdot(H, (double)rows[4 * i], (double)rows[4 * i + 1], &tx, &ty, &tw);
    double dx = tx / tw - (double)rows[4 * i + 2], dy = ty / tw - (double)rows[4 * i + 3];
    double dist = sqrt(dx * dx + dy * dy);
    const int r = (int)__builtin_rint(dist);  // Python round(): half to even
    rbin[i] = r;
    if ((unsigned)r < (unsigned)HB) atomicAdd(&B.u.h.hist[r], 1u);
    else big = 1;
  }
  __threadfence_block();
  big = __syncthreads_or(big);
  if (n == 0) return 0;
  // most populated bin; ties -> the bin whose first member comes first.  key = population << 32 | (0x7FFFFFFF - first
  // member): the largest key over the POINTS (a point carries the population of its own bin) is the largest over the
  // bins -- no per-bin "first member" table is needed (it cost 8 KB of LDS in every solver workgroup)
  unsigned long long bestkey = 0;
  if (!big) {
    for (int i = tid; i < n; i += NT) {
      const unsigned long long key = ((unsigned long long)B.u.h.hist[rbin[i]] << 32) | (unsigned)(0x7FFFFFFF - i);
      bestkey = key > bestkey ? key : bestkey;
    }
  } else {
    for (int i = tid; i < n; i += NT) {
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
  }
  for (int s = 32; s > 0; s >>= 1) {
    unsigned long long o = __shfl_xor(bestkey, s);
    bestkey = o > bestkey ? o : bestkey;
  }
  if (lane == 0) B.red[wave] = bestkey;
  __syncthreads();
  for (int w = 0; w < NW; w++) { const unsigned long long o = B.red[w]; bestkey = o > bestkey ? o : bestkey; }
  const int ibest = 0x7FFFFFFF - (int)(bestkey & 0xFFFFFFFFull);
  const int rbest = rbin[ibest];
  int mcount = 0;
  if (wave == 0) {
    for (int c0 = 0; c0 < n; c0 += NL) {
      const int i = c0 + lane;
      const bool f = i < n && rbin[i] == rbest;
      const unsigned long long b = __ballot(f);
      if (f) *reinterpret_cast<float4*>(out + 4 * (mcount + __popcll(b & ((1ull << lane) - 1ull)))) =
          *reinterpret_cast<const float4*>(rows + 4 * i);
      mcount += __popcll(b);
    }
    if (lane == 0) B.s.ib[3] = mcount;
  }
  __threadfence_block();
  __syncthreads();
  return B.s.ib[3];
}

// compute_homography (utils.py:351-362): optional pre-transform by Hsup (f64 -> f32), RANSAC #2, 0.7 gate.
// All threads; uniform result; ends with a workgroup barrier.
template <int NW, bool LANES>
__device__ __forceinline__ int compute_homography_block(BlockLds<NW, LANES>& B, const float* rows, int n, const double* Hsup /*LDS|null*/,
                                        const EvhRansacArgs& A, uint8_t* mask, float* trow, float* crow, int* info,
                                        const ScanPre* pre = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* use = rows;
  if (Hsup) {
    for (int i = tid; i < n; i += NW * NL) {
      double tx, ty, tw;
      hdot(Hsup, (double)rows[4 * i], (double)rows[4 * i + 1], &tx, &ty, &tw);
      float ax = (float)(tx / tw), ay = (float)(ty / tw);
      hdot(Hsup, (double)rows[4 * i + 2], (double)rows[4 * i + 3], &tx, &ty, &tw);
      float bx = (float)(tx / tw), by = (float)(ty / tw);
      *reinterpret_cast<float4*>(trow + 4 * i) = make_float4(ax, ay, bx, by);
    }
    __threadfence_block();
    __syncthreads();
    use = trow;
  }
  const bool found = find_homography_block<NW, LANES>(B, use, n, A.thr, A.max_iters, A.conf, A.force_max, mask, crow, info, A.prof, A.lane_v ? A.lane_v + (int64_t)blockIdx.x * (NW * LV_ELEMS * NL) : nullptr, pre);
  if (wave == 0) {
    int s = 0;
    for (int i = lane; i < n; i += NL) s += mask[i];
    s = wave_sum(s);
    if (lane == 0) B.gate = (double)s < 0.7 * (double)n ? 1 : 0;
  }
  __syncthreads();
  if (B.gate) return EVH_PAIR_LOW_INLIER_RATIO;
  if (!found) return EVH_PAIR_NO_FINAL_H;
  return EVH_PAIR_OK;
}

template <int NW, bool LANES>
__device__ __forceinline__ BlockLds<NW, LANES>& block_lds() {
  __shared__ BlockLds<NW, LANES> g;
  return g;
}

// generic single-problem entry (evh_find_homography_ransac)
template <int NW, bool LANES>
__global__ __launch_bounds__(NW * NL) void k_find_homography(EvhRansacArgs A) {
  BlockLds<NW, LANES>& B = block_lds<NW, LANES>();
  if (threadIdx.x == 0) B.s.fast = A.fast_solver;   // read after the first barrier of the solve
  const int tid = threadIdx.x;
  const int n = A.n_fixed;
  const bool found = find_homography_block<NW, LANES>(B, A.pts, n, A.thr, A.max_iters, A.conf, A.force_max, A.mask, A.crow, A.info, A.prof, A.lane_v);
  if (tid < 9) A.H[tid] = found ? B.s.H[tid] : 0.0;
  if (tid == 0) A.found[0] = found ? 1 : 0;
}

// generic static filter entry
template <int NW, bool LANES>
__global__ __launch_bounds__(NW * NL) void k_static_filter(const double* H, const float* rows, int n, int* rbin,
                                                           float* out, int* count) {
  BlockLds<NW, LANES>& B = block_lds<NW, LANES>();
  const int tid = threadIdx.x;
  if (tid < 9) B.s.H[tid] = H[tid];
  __syncthreads();
  const int m = static_filter_block<NW, LANES>(B, B.s.H, rows, n, rbin, out);
  if (tid == 0) count[0] = m;
}

// phase 1 of a pair: RANSAC #1 on the matched rows, then the static-point filter (matching.py:152-163)
template <int NW, bool LANES>
__global__ __launch_bounds__(NW * NL) void k_ransac_static(EvhRansacArgs A) {
  BlockLds<NW, LANES>& B = block_lds<NW, LANES>();
  if (threadIdx.x == 0) B.s.fast = A.fast_solver;   // read after the first barrier of the solve
  const int p = blockIdx.x, tid = threadIdx.x;
  if (A.status[p] != EVH_PAIR_OK) { if (tid == 0) A.npts2[p] = 0; return; }
  const int n = A.npts[p];
  const float* rows = A.pts + (int64_t)p * A.row_stride * 4;
  float* out = A.pts2 + (int64_t)p * A.row_stride * 4;
  uint8_t* mask = A.mask + (int64_t)p * A.row_stride;
  float* crow = A.crow + (int64_t)p * A.row_stride * 4;
  int* rbin = reinterpret_cast<int*>(A.lm + (int64_t)p * A.row_stride * 4);
  int* info = A.info ? A.info + 8 * p : nullptr;
  const bool found = find_homography_block<NW, LANES>(B, rows, n, A.thr, A.max_iters, A.conf, A.force_max, mask, crow, info, A.prof, A.lane_v ? A.lane_v + (int64_t)blockIdx.x * (NW * LV_ELEMS * NL) : nullptr);
  if (!found) {
    if (tid == 0) { A.status[p] = EVH_PAIR_NO_PROVISIONAL_H; A.npts2[p] = 0; }
    return;
  }
  if (A.H1 && tid < 9) A.H1[9 * p + tid] = B.s.H[tid];
  const int m = static_filter_block<NW, LANES>(B, B.s.H, rows, n, rbin, out);
  if (tid == 0) A.npts2[p] = m;
}

// phase 2: compute_homography.  Independent pairs: one workgroup per pair, Hsup = None.
template <int NW, bool LANES>
__global__ __launch_bounds__(NW * NL) void k_ransac_final_pairs(EvhRansacArgs A) {
  BlockLds<NW, LANES>& B = block_lds<NW, LANES>();
  if (threadIdx.x == 0) B.s.fast = A.fast_solver;   // read after the first barrier of the solve
  const int p = blockIdx.x, tid = threadIdx.x;
  int st = A.status[p];
  if (st == EVH_PAIR_OK) {
    const int n = A.npts2[p];
    const float* rows = A.pts2 + (int64_t)p * A.row_stride * 4;
    st = compute_homography_block<NW, LANES>(B, rows, n, nullptr, A, A.mask + (int64_t)p * A.row_stride,
                                      A.pts + (int64_t)p * A.row_stride * 4 /* matched rows are dead: scratch */,
                                      A.crow + (int64_t)p * A.row_stride * 4, A.info ? A.info + 8 * p + 4 : nullptr);
  }
  if (tid < 9) A.H[9 * p + tid] = st == EVH_PAIR_OK ? B.s.H[tid] : 0.0;
  if (tid == 0) A.out_status[p] = st;
}

// the end of one step of the scan: status, H of the pair (a failed pair repeats the previous H, none_H_processing=True),
// running superposition.  All threads; returns true when the scan stops here (uniform); ends with a workgroup barrier.
template <int NW, bool LANES>
__device__ __forceinline__ bool scan_step_tail(BlockLds<NW, LANES>& B, int st, int p, int npairs, double* Hout, int* stout, bool first) {
  const int tid = threadIdx.x;
  if (tid == 0) stout[p] = st;
  if (st != EVH_PAIR_OK && !B.have_prev) {
    // the reference raises here (None.tolist()); mark the pair and stop the scan
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    for (int q = p + 1 + tid; q < npairs; q += NW * NL) stout[q] = st;
    for (int q = 9 * p + tid; q < 9 * npairs; q += NW * NL) Hout[q] = nan;
    return true;
  }
  if (tid < 9) B.Hcur[tid] = st == EVH_PAIR_OK ? B.s.H[tid] : B.Hprev[tid];
  __syncthreads();
  if (tid < NL) {                          // wave 0
    if (tid < 9) { Hout[9 * p + tid] = B.Hcur[tid]; B.Hprev[tid] = B.Hcur[tid]; }
    // matrix_superposition (utils.py:139-145); np.dot(3x3,3x3) = forward FMA chain (pinned by fixtures)
    double P = 0;
    if (!first && tid < 9) {
      const int r = tid / 3, c = tid - 3 * r;
      P = fma(B.Hcur[3 * r + 2], B.Hsup[6 + c], fma(B.Hcur[3 * r + 1], B.Hsup[3 + c], B.Hcur[3 * r] * B.Hsup[c]));
    }
    const double P8 = __shfl(P, 8);
    WSYNC();
    if (tid < 9) B.Hsup[tid] = first ? B.Hcur[tid] : P / P8;
    if (tid == 0) B.have_prev = 1;
  }
  __syncthreads();
  return false;
}

// phase 2, stream semantics (video_processing.py:83-105): sequential scan over the pairs of one stream with the
// running superposition; a failed pair repeats the previous H (none_H_processing=True).
template <int NW, bool LANES>
__global__ __launch_bounds__(NW * NL) void k_ransac_final_stream(EvhRansacArgs A, int npairs, int pitch) {
  // one workgroup per stream: block s scans the npairs pairs whose per-pair slots start at s * pitch (several streams
  // of one batch sit `pitch` pair slots apart); H / status are written compactly at s * npairs + p
  BlockLds<NW, LANES>& B = block_lds<NW, LANES>();
  if (threadIdx.x == 0) B.s.fast = A.fast_solver;   // read after the first barrier of the solve
  const int tid = threadIdx.x, s = blockIdx.x;
  const double* Hsup0 = A.Hsup0 ? A.Hsup0 + 18 * s : nullptr;
  const double* Hprev0 = A.Hprev0 ? A.Hprev0 + 18 * s : nullptr;
  const int64_t slot0 = (int64_t)s * pitch;                 // first pair slot of this stream; also its scratch slot
  double* Hout = A.H + (int64_t)9 * s * npairs;
  int* stout = A.out_status + (int64_t)s * npairs;
  if (tid == 0) B.have_prev = Hprev0 ? 1 : 0;
  if (tid < 9 && Hsup0) B.Hsup[tid] = Hsup0[tid];
  if (tid < 9 && Hprev0) B.Hprev[tid] = Hprev0[tid];
  __syncthreads();
  bool first = Hsup0 == nullptr;
  for (int p = 0; p < npairs; p++) {
    int st = A.status[slot0 + p];
    if (st == EVH_PAIR_OK) {
      const int n = A.npts2[slot0 + p];
      const float* rows = A.pts2 + (slot0 + p) * A.row_stride * 4;
      // the scan is sequential: one pair's worth of scratch (the stream's first slot) serves all its pairs
      st = compute_homography_block<NW, LANES>(B, rows, n, first ? nullptr : B.Hsup, A, A.mask + slot0 * A.row_stride,
                                        A.pts + slot0 * A.row_stride * 4, A.crow + slot0 * A.row_stride * 4,
                                        A.info ? A.info + 8 * (slot0 + p) + 4 : nullptr);
    }
    if (scan_step_tail<NW, LANES>(B, st, p, npairs, Hout, stout, first)) return;
    first = false;
  }
  if (A.state_out && tid < 9) { A.state_out[18 * s + tid] = B.Hsup[tid]; A.state_out[18 * s + 9 + tid] = B.Hprev[tid]; }
}

// ---- fixed-iteration stream scan (force_max): the samples of a pair are independent, only the replay and the
// refinement are serial, so every pair takes two launches: k_scan_hyp (one 16-hypothesis chunk per workgroup, ~140
// workgroups) and k_scan_finish (one workgroup: replay, inlier mask, refit, LM, superposition).  The state of the scan
// lives in global memory between the launches.
struct ScanState { double Hsup[9], Hprev[9]; int have_prev, first, aborted, pad; };
struct ScanWs {
  ScanState* state;                // [nstreams]
  unsigned long long* rng_after;   // [nstreams * npairs] generator state after the table of the pair
  ushort4* quads;                  // [nstreams * npairs][hmax] the sample table: depends on the row count of the pair alone
  int* hyp;                        // [nstreams][hmax]
  double* hypH;                    // [nstreams][hmax][9]
  int hmax;
};

__global__ __launch_bounds__(NL) void k_scan_init(EvhRansacArgs A, ScanWs W) {
  const int s = blockIdx.x, tid = threadIdx.x;
  ScanState& T = W.state[s];
  if (tid < 9) { T.Hsup[tid] = A.Hsup0 ? A.Hsup0[18 * s + tid] : 0.0; T.Hprev[tid] = A.Hprev0 ? A.Hprev0[18 * s + tid] : 0.0; }
  if (tid == 0) { T.have_prev = A.Hprev0 ? 1 : 0; T.first = A.Hsup0 ? 0 : 1; T.aborted = 0; T.pad = 0; }
}

// the sample quadruples of every pair of the batch, all pairs at once (getSubset: they depend on the row count only, so
// they can be drawn before the scan reaches the pair)
__global__ __launch_bounds__(NL) void k_scan_quads(EvhRansacArgs A, int npairs, int pitch, ScanWs W, int phase1) {
  // phase1: the tables of RANSAC #1 of independent pairs (row counts in A.npts, one "stream" of one pair per block)
  const int b = blockIdx.x, s = b / npairs, p = b - s * npairs, lane = threadIdx.x;
  const int64_t slot = (int64_t)s * pitch + p;
  if (A.status[slot] != EVH_PAIR_OK) return;
  const int n = phase1 ? A.npts[slot] : A.npts2[slot];
  if (n <= 4) return;
  Rng rng;
  const FastMod fm((unsigned)n);
  ushort4* out = W.quads + (int64_t)b * W.hmax;
  unsigned long long after = 0ull;
  for (int h0 = 0; h0 < W.hmax; h0 += NL) {
    int m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (int j = 0; j < NL; j++) {
      const int q0 = (int)fm.mod(rng.next());
      int q1, q2, q3;
      do q1 = (int)fm.mod(rng.next()); while (q1 == q0);
      do q2 = (int)fm.mod(rng.next()); while (q2 == q0 || q2 == q1);
      do q3 = (int)fm.mod(rng.next()); while (q3 == q0 || q3 == q1 || q3 == q2);
      if (j == lane) { m0 = q0; m1 = q1; m2 = q2; m3 = q3; }
      if (h0 + j + 1 == W.hmax) after = rng.state;
    }
    if (h0 + lane < W.hmax) out[h0 + lane] = make_ushort4((unsigned short)m0, (unsigned short)m1, (unsigned short)m2, (unsigned short)m3);
  }
  if (lane == 0) W.rng_after[b] = after;
}

// LDS of the hypothesis kernels: the 16 row matrices and nothing else (23 KB: seven workgroups per compute unit; the
// full BlockLds of the finishing kernels would allow three)
struct HypLds { RowMat m[4][NG]; double Hsup[9]; };

// one chunk of 16 hypotheses (a 16-lane row each) of pair p per workgroup; grid (hmax / 16, nstreams)
__global__ __launch_bounds__(4 * NL) void k_scan_hyp(EvhRansacArgs A, int p, int npairs, int pitch, ScanWs W) {
  __shared__ HypLds B;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, gl = lane & 15, s = blockIdx.y;
  const ScanState& T = W.state[s];
  if (T.aborted) return;
  const int64_t slot0 = (int64_t)s * pitch, slot = slot0 + p;
  if (A.status[slot] != EVH_PAIR_OK) return;
  const int n = A.npts2[slot];
  if (n <= 4) return;
  const float* use = A.pts2 + slot * A.row_stride * 4;
  if (!T.first) {
    // the rows in the fixed plane (compute_homography_block's transform).  Every workgroup of the pair writes the same
    // values to the stream's one scratch slot and reads back what it wrote itself -- identical bits from every writer.
    const float* rows = use;
    float* trow = A.pts + slot0 * A.row_stride * 4;
    if (tid < 9) B.Hsup[tid] = T.Hsup[tid];
    __syncthreads();
    for (int i = tid; i < n; i += 4 * NL) {
      double tx, ty, tw;
      hdot(B.Hsup, (double)rows[4 * i], (double)rows[4 * i + 1], &tx, &ty, &tw);
      float ax = (float)(tx / tw), ay = (float)(ty / tw);
      hdot(B.Hsup, (double)rows[4 * i + 2], (double)rows[4 * i + 3], &tx, &ty, &tw);
      float bx = (float)(tx / tw), by = (float)(ty / tw);
      *reinterpret_cast<float4*>(trow + 4 * i) = make_float4(ax, ay, bx, by);
    }
    __threadfence_block();
    __syncthreads();
    use = trow;
  }
  double thr = A.thr;
  if (thr <= 0) thr = 3;
  const float t = (float)(thr * thr);
  const int hg = blockIdx.x * (4 * NG) + wave * NG + row;
  const ushort4 q = W.quads[((int64_t)s * npairs + p) * W.hmax + hg];
  const int my[4] = {q.x, q.y, q.z, q.w};
  float Mx[4], My[4], mx[4], my_[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float4 r = *reinterpret_cast<const float4*>(use + 4 * my[i]);
    Mx[i] = r.x; My[i] = r.y; mx[i] = r.z; my_[i] = r.w;
  }
  const bool valid = check_subset4(Mx, My, mx, my_);
  double H[9];
  const bool ok = dlt4_rows(B.m[wave][row], lane, valid, Mx, My, mx, my_, H);
  int good = 0;
  if (ok) {
    float Hf[8];
#pragma unroll
    for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
#pragma unroll 4
    for (int i = gl; i < n; i += GL) {
      const float4 r = *reinterpret_cast<const float4*>(use + 4 * i);
      good += is_inlier(Hf, r.x, r.y, r.z, r.w, t) ? 1 : 0;
    }
  }
  good = rsum16(good);
  if (gl == 0) {
    W.hyp[(int64_t)s * W.hmax + hg] = (int)((valid ? 0x80000000u : 0u) | (ok ? 0x40000000u : 0u) | (unsigned)good);
    double* Ho = W.hypH + ((int64_t)s * W.hmax + hg) * 9;
#pragma unroll
    for (int i = 0; i < 9; i++) Ho[i] = ok ? H[i] : 0.0;
  }
}

// the serial part of pair p: replay of the hypotheses, refinement, step of the scan; grid = nstreams
__global__ __launch_bounds__(4 * NL) void k_scan_finish(EvhRansacArgs A, int p, int npairs, int pitch, ScanWs W) {
  BlockLds<4, false>& B = block_lds<4, false>();
  if (threadIdx.x == 0) B.s.fast = A.fast_solver;
  const int tid = threadIdx.x, s = blockIdx.x;
  ScanState& T = W.state[s];
  if (T.aborted) return;
  const int64_t slot0 = (int64_t)s * pitch;
  double* Hout = A.H + (int64_t)9 * s * npairs;
  int* stout = A.out_status + (int64_t)s * npairs;
  const bool first = T.first != 0;
  if (tid == 0) B.have_prev = T.have_prev;
  if (tid < 9) { B.Hsup[tid] = T.Hsup[tid]; B.Hprev[tid] = T.Hprev[tid]; }
  __syncthreads();
  int st = A.status[slot0 + p];
  if (st == EVH_PAIR_OK) {
    const int n = A.npts2[slot0 + p];
    const float* rows = A.pts2 + (slot0 + p) * A.row_stride * 4;
    const ScanPre pre{W.hyp + (int64_t)s * W.hmax, W.hypH + (int64_t)s * W.hmax * 9, W.hmax, n > 4 ? W.rng_after[(int64_t)s * npairs + p] : 0ull};
    st = compute_homography_block<4, false>(B, rows, n, first ? nullptr : B.Hsup, A, A.mask + slot0 * A.row_stride,
                                            A.pts + slot0 * A.row_stride * 4, A.crow + slot0 * A.row_stride * 4,
                                            A.info ? A.info + 8 * (slot0 + p) + 4 : nullptr, n > 4 ? &pre : nullptr);
  }
  if (scan_step_tail<4, false>(B, st, p, npairs, Hout, stout, first)) {
    if (tid == 0) T.aborted = 1;
    return;
  }
  if (tid < 9) { T.Hsup[tid] = B.Hsup[tid]; T.Hprev[tid] = B.Hprev[tid]; }
  if (tid == 0) { T.have_prev = 1; T.first = 0; }
  if (p == npairs - 1 && A.state_out && tid < 9) { A.state_out[18 * s + tid] = B.Hsup[tid]; A.state_out[18 * s + 9 + tid] = B.Hprev[tid]; }
}

// ---- fixed-iteration RANSAC #1 of a SMALL batch of pairs (a stream chunk): the same split -- k_static_hyp evaluates one
// 16-hypothesis chunk per workgroup (grid: chunks x pairs), k_static_finish replays, refines and runs the static filter.
// (One workgroup per pair with per-lane solvers is the throughput form for hundreds of pairs; alone it takes 4.8 ms.)
__global__ __launch_bounds__(4 * NL) void k_static_hyp(EvhRansacArgs A, ScanWs W) {
  __shared__ HypLds B;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, gl = lane & 15, p = blockIdx.y;
  if (A.status[p] != EVH_PAIR_OK) return;
  const int n = A.npts[p];
  if (n <= 4) return;
  const float* use = A.pts + (int64_t)p * A.row_stride * 4;
  double thr = A.thr;
  if (thr <= 0) thr = 3;
  const float t = (float)(thr * thr);
  const int hg = blockIdx.x * (4 * NG) + wave * NG + row;
  const ushort4 q = W.quads[(int64_t)p * W.hmax + hg];
  const int my[4] = {q.x, q.y, q.z, q.w};
  float Mx[4], My[4], mx[4], my_[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float4 r = *reinterpret_cast<const float4*>(use + 4 * my[i]);
    Mx[i] = r.x; My[i] = r.y; mx[i] = r.z; my_[i] = r.w;
  }
  const bool valid = check_subset4(Mx, My, mx, my_);
  double H[9];
  const bool ok = dlt4_rows(B.m[wave][row], lane, valid, Mx, My, mx, my_, H);
  int good = 0;
  if (ok) {
    float Hf[8];
#pragma unroll
    for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
#pragma unroll 4
    for (int i = gl; i < n; i += GL) {
      const float4 r = *reinterpret_cast<const float4*>(use + 4 * i);
      good += is_inlier(Hf, r.x, r.y, r.z, r.w, t) ? 1 : 0;
    }
  }
  good = rsum16(good);
  if (gl == 0) {
    W.hyp[(int64_t)p * W.hmax + hg] = (int)((valid ? 0x80000000u : 0u) | (ok ? 0x40000000u : 0u) | (unsigned)good);
    double* Ho = W.hypH + ((int64_t)p * W.hmax + hg) * 9;
#pragma unroll
    for (int i = 0; i < 9; i++) Ho[i] = ok ? H[i] : 0.0;
  }
}

__global__ __launch_bounds__(4 * NL) void k_static_finish(EvhRansacArgs A, ScanWs W) {
  BlockLds<4, false>& B = block_lds<4, false>();
  if (threadIdx.x == 0) B.s.fast = A.fast_solver;
  const int p = blockIdx.x, tid = threadIdx.x;
  if (A.status[p] != EVH_PAIR_OK) { if (tid == 0) A.npts2[p] = 0; return; }
  const int n = A.npts[p];
  const float* rows = A.pts + (int64_t)p * A.row_stride * 4;
  float* out = A.pts2 + (int64_t)p * A.row_stride * 4;
  uint8_t* mask = A.mask + (int64_t)p * A.row_stride;
  float* crow = A.crow + (int64_t)p * A.row_stride * 4;
  int* rbin = reinterpret_cast<int*>(A.lm + (int64_t)p * A.row_stride * 4);
  int* info = A.info ? A.info + 8 * p : nullptr;
  const ScanPre pre{W.hyp + (int64_t)p * W.hmax, W.hypH + (int64_t)p * W.hmax * 9, W.hmax, n > 4 ? W.rng_after[p] : 0ull};
  const bool found = find_homography_block<4, false>(B, rows, n, A.thr, A.max_iters, A.conf, A.force_max, mask, crow, info, A.prof,
                                                     nullptr, n > 4 ? &pre : nullptr);
  if (!found) {
    if (tid == 0) { A.status[p] = EVH_PAIR_NO_PROVISIONAL_H; A.npts2[p] = 0; }
    return;
  }
  if (A.H1 && tid < 9) A.H1[9 * p + tid] = B.s.H[tid];
  const int m = static_filter_block<4, false>(B, B.s.H, rows, n, rbin, out);
  if (tid == 0) A.npts2[p] = m;
}

// waves per workgroup: enough rows to cover the handful of hypotheses an adaptive RANSAC needs in one chunk when the
// launch is small (latency), one wave per pair when the launch fills the chip anyway (throughput); with the iteration
// count forced, four waves of lane-per-hypothesis solvers (256 hypotheses per chunk)
int waves_for(int nblocks, int force_max) {
  if (force_max) return 0;      // the LANES form
  return nblocks >= 512 ? 1 : 4;
}

// workspace of the fixed-iteration stream scan, grown on demand
int scan_ws(evh_ctx* c, int nstreams, int npairs, int hmax, ScanWs* W) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t slots = (size_t)nstreams * npairs;
  const size_t o_state = 0, o_rng = o_state + up(sizeof(ScanState) * nstreams), o_quads = o_rng + up(8 * slots),
               o_hyp = o_quads + up(sizeof(ushort4) * slots * hmax), o_H = o_hyp + up(sizeof(int) * (size_t)nstreams * hmax),
               total = o_H + up(sizeof(double) * 9 * (size_t)nstreams * hmax);
  if (c->scan_ws_bytes < total) {
    if (c->d_scan_ws) { EVH_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_scan_ws); c->d_scan_ws = nullptr; c->scan_ws_bytes = 0; }
    EVH_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_scan_ws), total));
    c->scan_ws_bytes = total;
  }
  char* b = c->d_scan_ws;
  W->state = reinterpret_cast<ScanState*>(b + o_state); W->rng_after = reinterpret_cast<unsigned long long*>(b + o_rng);
  W->quads = reinterpret_cast<ushort4*>(b + o_quads); W->hyp = reinterpret_cast<int*>(b + o_hyp);
  W->hypH = reinterpret_cast<double*>(b + o_H); W->hmax = hmax;
  return EVH_SUCCESS;
}

// enough 16-hypothesis chunks for max_iters counted samples plus 1/8 of rejected ones; a pair that needs more continues
// inside the finishing kernel with the ordinary chunk loop
int forced_chunks(const EvhRansacArgs& A) {
  const int iters = A.max_iters > 0 ? A.max_iters : 1;
  int chunks = (iters + iters / 8 + 4 * NG - 1) / (4 * NG) + 1;
  if (const char* e = getenv("EVH_SCAN_CHUNKS")) chunks = atoi(e) > 0 ? atoi(e) : chunks;   // tests: a short table, the rest in the finishing kernel
  return chunks > 608 ? 608 : chunks;                               // < 10000 samples (ScanPre)
}

int launch_forced_static(evh_ctx* c, const EvhRansacArgs& A, int npairs) {
  if (A.row_stride > 65536) return evh_fail(c, EVH_ERR_INVALID, "fixed-iteration RANSAC: more than 65536 rows per pair");
  const int chunks = forced_chunks(A);
  ScanWs W;
  int rc = scan_ws(c, npairs, 1, chunks * 4 * NG, &W);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan_quads, dim3(npairs), dim3(NL), 0, c->stream, A, 1, 1, W, 1);
  hipLaunchKernelGGL(k_static_hyp, dim3(chunks, npairs), dim3(4 * NL), 0, c->stream, A, W);
  hipLaunchKernelGGL(k_static_finish, dim3(npairs), dim3(4 * NL), 0, c->stream, A, W);
  return EVH_SUCCESS;
}

int launch_forced_scan(evh_ctx* c, const EvhRansacArgs& A, int npairs, int nstreams, int pitch) {
  if (A.row_stride > 65536) return evh_fail(c, EVH_ERR_INVALID, "fixed-iteration scan: more than 65536 rows per pair");
  const int chunks = forced_chunks(A);
  ScanWs W;
  int rc = scan_ws(c, nstreams, npairs, chunks * 4 * NG, &W);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan_init, dim3(nstreams), dim3(NL), 0, c->stream, A, W);
  hipLaunchKernelGGL(k_scan_quads, dim3(nstreams * npairs), dim3(NL), 0, c->stream, A, npairs, pitch, W, 0);
  for (int p = 0; p < npairs; p++) {
    hipLaunchKernelGGL(k_scan_hyp, dim3(chunks, nstreams), dim3(4 * NL), 0, c->stream, A, p, npairs, pitch, W);
    hipLaunchKernelGGL(k_scan_finish, dim3(nstreams), dim3(4 * NL), 0, c->stream, A, p, npairs, pitch, W);
  }
  return EVH_SUCCESS;
}

}  // namespace

#define EVH_LAUNCH_NW(nw, kernel, grid, stream, ...)                                                               \
  do {                                                                                                            \
    if ((nw) == 0) hipLaunchKernelGGL((kernel<4, true>), dim3(grid), dim3(4 * NL), 0, stream, __VA_ARGS__);        \
    else if ((nw) == 4) hipLaunchKernelGGL((kernel<4, false>), dim3(grid), dim3(4 * NL), 0, stream, __VA_ARGS__);  \
    else hipLaunchKernelGGL((kernel<1, false>), dim3(grid), dim3(NL), 0, stream, __VA_ARGS__);                     \
  } while (0)

int evh_launch_find_homography(evh_ctx* c, const EvhRansacArgs& A) {
  if (A.force_max && !A.lane_v) return evh_fail(c, EVH_ERR_HIP, "fixed-iteration RANSAC: the per-lane scratch could not be allocated");
  EVH_LAUNCH_NW(waves_for(1, A.force_max), k_find_homography, 1, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
int evh_launch_static_filter(evh_ctx* c, const double* d_H, const float* d_rows, int n, int* d_rbin, float* d_out,
                             int* d_count) {
  hipLaunchKernelGGL((k_static_filter<4, false>), dim3(1), dim3(4 * NL), 0, c->stream, d_H, d_rows, n, d_rbin, d_out, d_count);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
int evh_launch_ransac_static(evh_ctx* c, const EvhRansacArgs& A, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  if (A.force_max && !A.lane_v) return evh_fail(c, EVH_ERR_HIP, "fixed-iteration RANSAC: the per-lane scratch could not be allocated");
  if (A.force_max && npairs <= 256 && !getenv("EVH_SCAN_ONE_WG")) {        // a stream chunk: spread the samples of every pair
    const int fr = launch_forced_static(c, A, npairs);
    if (fr) return fr;
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  EVH_LAUNCH_NW(waves_for(npairs, A.force_max), k_ransac_static, npairs, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
int evh_launch_ransac_final(evh_ctx* c, const EvhRansacArgs& A_, int npairs, int nstreams, int pitch) {
  if (npairs <= 0) return EVH_SUCCESS;
  EvhRansacArgs A = A_;
  if (A.force_max && !A.lane_v) return evh_fail(c, EVH_ERR_HIP, "fixed-iteration RANSAC: the per-lane scratch could not be allocated");
  static const bool want_prof = getenv("EVH_RANSAC_PROF") != nullptr;   // debugging aid: cycle accounting to stderr
  unsigned long long* d_prof = nullptr;
  if (want_prof) {
    EVH_HIP(c, hipMalloc(&d_prof, sizeof(unsigned long long) * PF_NSLOTS));
    if (hipMemsetAsync(d_prof, 0, sizeof(unsigned long long) * PF_NSLOTS, c->stream) != hipSuccess) {
      (void)hipFree(d_prof);
      return evh_fail(c, EVH_ERR_HIP, "EVH_RANSAC_PROF: hipMemsetAsync failed");
    }
    A.prof = d_prof;
  }
  // nstreams == 0: independent pairs; otherwise nstreams sequential scans of npairs pairs each, `pitch` pair slots apart
  if (nstreams > 0 && A.force_max && !getenv("EVH_SCAN_ONE_WG")) {
    const int fr = launch_forced_scan(c, A, npairs, nstreams, pitch);
    if (fr) { if (d_prof) (void)hipFree(d_prof); return fr; }
  } else if (nstreams > 0) EVH_LAUNCH_NW(waves_for(nstreams, A.force_max), k_ransac_final_stream, nstreams, c->stream, A, npairs, pitch);
  else EVH_LAUNCH_NW(waves_for(npairs, A.force_max), k_ransac_final_pairs, npairs, c->stream, A);
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { if (d_prof) (void)hipFree(d_prof); return evh_fail(c, EVH_ERR_HIP, std::string("k_ransac_final: ") + hipGetErrorString(le)); }
  }
  if (d_prof) {
    unsigned long long h[PF_NSLOTS];
    hipError_t pe = hipStreamSynchronize(c->stream);
    if (pe == hipSuccess) pe = hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d_prof);                       // released on every path
    if (pe != hipSuccess) return evh_fail(c, EVH_ERR_HIP, std::string("EVH_RANSAC_PROF read-back: ") + hipGetErrorString(pe));
    const double n = h[PF_CALLS] ? (double)h[PF_CALLS] : 1.0;
    fprintf(stderr, "[evh ransac_final prof] calls %llu | per call (cycles): total %.0f hyp %.0f (chunks %.2f, dlt4+jacobi %.0f) "
            "compact %.0f refit %.0f lm %.0f (iters %.2f, solve8 %.0f, eval %.0f) | rotations: 9x9 %.1f 8x8 %.1f | chunk loop: rng %.0f "
            "count %.0f barrier %.0f replay %.0f | 4-wave eval: steps %.1f, busy per step wave0 (singles) %.0f wave1 (terms) %.0f wave2 (products) %.0f wave3 (pairs) %.0f, wave0 at the barrier %.0f\n",
            h[PF_CALLS], h[PF_TOTAL] / n, h[PF_HYP] / n, h[PF_CHUNKS] / n, h[PF_SETUP] / n, h[PF_COMPACT] / n,
            h[PF_REFIT] / n, h[PF_LM] / n, h[PF_LM_ITERS] / n, h[PF_SOLVE8] / n, h[PF_EVAL] / n, h[PF_ROT9] / n,
            h[PF_ROT8] / n, h[PF_RNG] / n, h[PF_COUNT] / n, h[PF_BARRIER] / n, h[PF_REPLAY] / n, h[PF_MW_STEPS] / n,
            h[PF_MW_W0] / (double)(h[PF_MW_STEPS] ? h[PF_MW_STEPS] : 1), h[PF_MW_W1] / (double)(h[PF_MW_STEPS] ? h[PF_MW_STEPS] : 1),
            h[PF_MW_W2] / (double)(h[PF_MW_STEPS] ? h[PF_MW_STEPS] : 1), h[PF_MW_W3] / (double)(h[PF_MW_STEPS] ? h[PF_MW_STEPS] : 1),
            h[PF_MW_WAIT] / (double)(h[PF_MW_STEPS] ? h[PF_MW_STEPS] : 1));
  }
  return EVH_SUCCESS;
}
