// evh_match.hip -- brute-force 2-NN over 32-byte descriptors + the reference's match filters, gfx950.
// Replaces cv2.DescriptorMatcher_create("BruteForce").knnMatch(q, t, 2) (matching.py:102-108) and the glue
// lowes_ratio_test / filter_corresponding_points (matching.py:166-239) / remove_double_matching (utils.py:41-68).
// Distances are exact integers: D = |q|^2 + |t|^2 - 2 q.t with v_dot4_u32_u8 on LDS-staged train tiles.
#include "evh_internal.h"
#include "evh_match.h"
#include <cfloat>
#include <climits>
#include <cstdlib>

namespace {

#define MT_TILE 512  // train descriptors per LDS tile (16 KB)

__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t acc) {   // v_dot4_u32_u8
  return __builtin_amdgcn_udot4(a, b, acc, false);
}

// a < b as the operator sees it: distances are compared after sqrt in float32 (batchDistance: dist = sqrt(D), insertion
// on strictly smaller dist).  sqrtf is strictly monotone on integers below 2^22 (spacing of sqrt >= 2^-12 > half an ulp
// there); above, two different D can round to the same float, and then they tie.  32-byte descriptors never get there
// (D <= 2 080 800); 128-byte ones can (D <= 8 323 200).
__device__ __forceinline__ bool dist_lt(uint32_t a, uint32_t b) {
  if (a >= b) return false;
  if (a < (1u << 22) || b == 0xFFFFFFFFu) return true;
  return sqrtf((float)a) < sqrtf((float)b);
}

// one workgroup (256 threads) per (pair, chunk of 256 queries): blockIdx.y = chunk, so that a few pairs with many
// descriptors (4K frames, N = 4000) still fill the chip; the train set is streamed through LDS by every chunk.
// NV = uint4 per descriptor: 2 (32 bytes, ORB) or 8 (128 bytes: SIFT's descriptor values, 0..255 each -- the operator
// holds them as float32 and its float accumulation of the squared differences is exact, every partial sum being an
// integer below 2^24).
template <int NV, int TILE>
__global__ __launch_bounds__(256) void k_knn2(EvhKnnArgs A) {
  __shared__ uint4 tdesc[TILE * NV];
  __shared__ uint32_t tnorm[TILE];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int qs = A.q_slot0 + p * A.q_slot_step, ts = A.t_slot0 + p * A.t_slot_step;
  const int nq = A.nq_arr ? A.nq_arr[qs] : A.nq_fixed;
  const int nt = A.nt_arr ? A.nt_arr[ts] : A.nt_fixed;
  const uint4* Q = reinterpret_cast<const uint4*>(A.q + (int64_t)qs * A.slot_bytes);
  const uint4* T = reinterpret_cast<const uint4*>(A.t + (int64_t)ts * A.slot_bytes);
  int32_t* oidx = A.idx + (int64_t)p * A.out_stride * 2;
  uint32_t* od2 = A.d2 + (int64_t)p * A.out_stride * 2;
  for (int q0 = blockIdx.y * 256; q0 < nq; q0 += 256 * gridDim.y) {   // workgroup-uniform bounds
    const int qi = q0 + tid;
    const bool act = qi < nq;
    uint4 qv[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) qv[v] = act ? Q[NV * qi + v] : make_uint4(0, 0, 0, 0);
    uint32_t qn = 0;
    if (!A.hamming) {
#pragma unroll
      for (int v = 0; v < NV; v++) {
        qn = dot4(qv[v].x, qv[v].x, qn); qn = dot4(qv[v].y, qv[v].y, qn); qn = dot4(qv[v].z, qv[v].z, qn); qn = dot4(qv[v].w, qv[v].w, qn);
      }
    }
    uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
    int i0 = -1, i1 = -1;
    for (int t0 = 0; t0 < nt; t0 += TILE) {
      const int tn = min(TILE, nt - t0);
      __syncthreads();
      for (int i = tid; i < tn * NV; i += 256) tdesc[i] = T[NV * t0 + i];
      __syncthreads();
      if (!A.hamming)
        for (int i = tid; i < tn; i += 256) {
          uint32_t n = 0;
#pragma unroll
          for (int v = 0; v < NV; v++) {
            const uint4 a = tdesc[NV * i + v];
            n = dot4(a.x, a.x, n); n = dot4(a.y, a.y, n); n = dot4(a.z, a.z, n); n = dot4(a.w, a.w, n);
          }
          tnorm[i] = n;
        }
      __syncthreads();
      if (act) {
        for (int j = 0; j < tn; j++) {
          uint32_t d;
          if (A.hamming) {
            d = 0;
#pragma unroll
            for (int v = 0; v < NV; v++) {
              const uint4 ta = tdesc[NV * j + v];
              d += __popc(qv[v].x ^ ta.x) + __popc(qv[v].y ^ ta.y) + __popc(qv[v].z ^ ta.z) + __popc(qv[v].w ^ ta.w);
            }
          } else {
            uint32_t s = 0;
#pragma unroll
            for (int v = 0; v < NV; v++) {
              const uint4 ta = tdesc[NV * j + v];
              s = dot4(qv[v].x, ta.x, s); s = dot4(qv[v].y, ta.y, s); s = dot4(qv[v].z, ta.z, s); s = dot4(qv[v].w, ta.w, s);
            }
            d = qn + tnorm[j] - 2u * s;
          }
          // ascending train order, strict '<': ties keep the lowest train index
          if (NV == 2) {
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = t0 + j; }
            else if (d < b1) { b1 = d; i1 = t0 + j; }
          } else {
            if (dist_lt(d, b0)) { b1 = b0; i1 = i0; b0 = d; i0 = t0 + j; }
            else if (dist_lt(d, b1)) { b1 = d; i1 = t0 + j; }
          }
        }
      }
    }
    if (act) {
      oidx[2 * qi] = i0; oidx[2 * qi + 1] = i1;
      od2[2 * qi] = b0; od2[2 * qi + 1] = b1;
    }
  }
}

// ---- the same 2-NN over 128-byte rows on the matrix cores (round 4) -------------------------------------------------------
// SIFT at 720p is 25 000 x 25 000 rows per pair: 85 G byte products, 1.06 ms per pair with v_dot4 above.  The Gram matrix is
// an integer GEMM, so v_mfma_i32_32x32x32_i8 computes it EXACTLY: with a = t - 128 (= t ^ 0x80) and b = 127 - q (= q ^ 0x7F)
// as signed bytes,  sum a.b = 127 St + 128 Sq - 2 080 768 - q.t,  hence
//     D = |q|^2 + |t|^2 - 2 q.t = [ |t|^2 - 254 St ] + 2 sum a.b + [ |q|^2 - 256 Sq + 4 161 536 ] = T(t) + 2 acc + C(q).
// Trains are the A operand (rows of the 32x32 result), queries the B operand (column = lane & 31), so a lane owns ONE query
// and sees 16 trains of a tile in its 16 accumulators; lane l and l + 32 split the 32 trains of a tile between them and are
// merged once at the end under (distance class, index) -- the serial strict insertion keeps exactly the first two under that
// order.  floor(T / 2) is the C-in of the first MFMA (read from LDS straight into the accumulators), so the epilogue of a
// tile is a 15-instruction minimum and ONE compare against ceil((best2 - C) / 2): 2 e' >= best2 - C  =>  D >= best2, reject;
// everything that passes takes the exact path (D rebuilt with T's parity bit, dist_lt as in k_knn2).  A workgroup stages 64
// train rows per barrier (XOR-swizzled 16-byte slots: ds_read_b128 without bank conflicts) for 4 waves x 128 queries.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define KM_ROWS 64
#define KM_QB 4
#define KM_WQ (32 * KM_QB)
#define KM_GQ (4 * KM_WQ)
struct KmStage {
  alignas(16) uint8_t rows[KM_ROWS * 128];
  alignas(16) int th[KM_ROWS];      // floor(T / 2)
  alignas(16) int tf[KM_ROWS];      // T (parity used by the exact path); rows past the end: 0x3FFFFFFE
};

__device__ __forceinline__ int min16(const v16i& a) {
  const int m0 = min(min(a[0], a[1]), min(a[2], a[3])), m1 = min(min(a[4], a[5]), min(a[6], a[7]));
  const int m2 = min(min(a[8], a[9]), min(a[10], a[11])), m3 = min(min(a[12], a[13]), min(a[14], a[15]));
  return min(min(m0, m1), min(m2, m3));
}
// (d, i) before (e, j) in the operator's order: smaller distance class first, then the lower train index
__device__ __forceinline__ bool knn_before(uint32_t d, int i, uint32_t e, int j) {
  return dist_lt(d, e) || (!dist_lt(e, d) && i < j);
}

__global__ __launch_bounds__(256) void k_knn2_mfma128(EvhKnnArgs A) {
  __shared__ KmStage S[2];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int qs = A.q_slot0 + p * A.q_slot_step, ts = A.t_slot0 + p * A.t_slot_step;
  const int nq = A.nq_arr ? A.nq_arr[qs] : A.nq_fixed;
  const int nt = A.nt_arr ? A.nt_arr[ts] : A.nt_fixed;
  const uint4* Q = reinterpret_cast<const uint4*>(A.q + (int64_t)qs * A.slot_bytes);
  const uint4* T = reinterpret_cast<const uint4*>(A.t + (int64_t)ts * A.slot_bytes);
  int32_t* oidx = A.idx + (int64_t)p * A.out_stride * 2;
  uint32_t* od2 = A.d2 + (int64_t)p * A.out_stride * 2;
  const int ntile = (nt + KM_ROWS - 1) / KM_ROWS;
  for (int q0 = blockIdx.y * KM_GQ; q0 < nq; q0 += KM_GQ * gridDim.y) {      // workgroup-uniform bounds
    v4i bq[KM_QB][4];
    int Cq[KM_QB], thr[KM_QB], i0[KM_QB], i1[KM_QB];
    uint32_t b0[KM_QB], b1[KM_QB];
#pragma unroll
    for (int b = 0; b < KM_QB; b++) {
      const int qi = q0 + wave * KM_WQ + 32 * b + r;
      uint32_t s1 = 0, s2 = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const uint4 x = qi < nq ? Q[8 * qi + 2 * s + h] : make_uint4(0, 0, 0, 0);
        s1 = dot4(x.x, 0x01010101u, s1); s1 = dot4(x.y, 0x01010101u, s1); s1 = dot4(x.z, 0x01010101u, s1); s1 = dot4(x.w, 0x01010101u, s1);
        s2 = dot4(x.x, x.x, s2); s2 = dot4(x.y, x.y, s2); s2 = dot4(x.z, x.z, s2); s2 = dot4(x.w, x.w, s2);
        bq[b][s] = v4i{(int)(x.x ^ 0x7F7F7F7Fu), (int)(x.y ^ 0x7F7F7F7Fu), (int)(x.z ^ 0x7F7F7F7Fu), (int)(x.w ^ 0x7F7F7F7Fu)};
      }
      s1 += (uint32_t)__shfl_xor((int)s1, 32); s2 += (uint32_t)__shfl_xor((int)s2, 32);     // the other half of the row
      Cq[b] = (int)s2 - 256 * (int)s1 + 4161536;
      thr[b] = INT_MAX; b0[b] = 0xFFFFFFFFu; b1[b] = 0xFFFFFFFFu; i0[b] = -1; i1[b] = -1;
    }
    uint4 g[2];
    auto gload = [&](int t0) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int i = tid + 256 * k, gr = t0 + (i >> 3);
        g[k] = gr < nt ? T[8 * gr + (i & 7)] : make_uint4(0, 0, 0, 0);
      }
    };
    auto lstore = [&](KmStage& st, int t0) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int i = tid + 256 * k, row = i >> 3, c = i & 7;
        const uint4 x = g[k];
        uint32_t s1 = 0, s2 = 0;
        s1 = dot4(x.x, 0x01010101u, s1); s1 = dot4(x.y, 0x01010101u, s1); s1 = dot4(x.z, 0x01010101u, s1); s1 = dot4(x.w, 0x01010101u, s1);
        s2 = dot4(x.x, x.x, s2); s2 = dot4(x.y, x.y, s2); s2 = dot4(x.z, x.z, s2); s2 = dot4(x.w, x.w, s2);
        int t = (int)s2 - 254 * (int)s1;
        t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4);
        *reinterpret_cast<uint4*>(&st.rows[row * 128 + ((c ^ ((row >> 1) & 7)) << 4)]) =
            make_uint4(x.x ^ 0x80808080u, x.y ^ 0x80808080u, x.z ^ 0x80808080u, x.w ^ 0x80808080u);
        if (c == 0) {
          const int tfull = t0 + row < nt ? t : 0x3FFFFFFE;
          st.tf[row] = tfull; st.th[row] = tfull >> 1;
        }
      }
    };
    __syncthreads();                                   // the previous query block is done with both stages
    if (ntile > 0) { gload(0); lstore(S[0], 0); }
    __syncthreads();
    for (int it = 0; it < ntile; it++) {
      const int t0 = it * KM_ROWS;
      KmStage& st = S[it & 1];
      if (it + 1 < ntile) gload(t0 + KM_ROWS);
      for (int sub = 0; sub < 2; sub++) {
        const int row = 32 * sub + r;
        v4i a[4];
        v16i tc;                                       // floor(T / 2) of this lane's 16 rows: C-in of the first MFMA
#pragma unroll
        for (int s = 0; s < 4; s++)
          a[s] = *reinterpret_cast<const v4i*>(&st.rows[row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4)]);
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
          const v4i t4 = *reinterpret_cast<const v4i*>(&st.th[32 * sub + 8 * gq + 4 * h]);
          tc[4 * gq] = t4[0]; tc[4 * gq + 1] = t4[1]; tc[4 * gq + 2] = t4[2]; tc[4 * gq + 3] = t4[3];
        }
#pragma unroll
        for (int b = 0; b < KM_QB; b++) {
          v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], bq[b][0], tc, 0, 0, 0);
#pragma unroll
          for (int s = 1; s < 4; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[b][s], acc, 0, 0, 0);
          const int m = min16(acc);
          if (__ballot(m < thr[b]) != 0ull) {          // some lane may have a new best two (every other tile early on, rare later)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
              const int mg = min(min(acc[4 * gq], acc[4 * gq + 1]), min(acc[4 * gq + 2], acc[4 * gq + 3]));
              if (__ballot(mg < thr[b]) == 0ull) continue;
              const v4i par4 = *reinterpret_cast<const v4i*>(&st.tf[32 * sub + 8 * gq + 4 * h]);
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const int e1 = acc[4 * gq + j];
                const int gi = t0 + 32 * sub + 8 * gq + 4 * h + j;
                const bool cand = e1 < thr[b] && gi < nt;
                const uint32_t d = (uint32_t)(2 * e1 + (par4[j] & 1) + Cq[b]);
                // below 2^22 the float32 square roots are strictly ordered like the integers: plain compares.  A wave
                // with a candidate at or above 2^22 (saturated rows only) takes the exact form for this element.
                if (__ballot(cand && d >= (1u << 22)) == 0ull) {
                  const bool lt0 = cand && d < b0[b], lt1 = cand && d < b1[b];
                  b1[b] = lt0 ? b0[b] : lt1 ? d : b1[b]; i1[b] = lt0 ? i0[b] : lt1 ? gi : i1[b];
                  b0[b] = lt0 ? d : b0[b]; i0[b] = lt0 ? gi : i0[b];
                } else if (cand) {
                  if (dist_lt(d, b0[b])) { b1[b] = b0[b]; i1[b] = i0[b]; b0[b] = d; i0[b] = gi; }
                  else if (dist_lt(d, b1[b])) { b1[b] = d; i1[b] = gi; }
                }
              }
              thr[b] = b1[b] == 0xFFFFFFFFu ? INT_MAX : ((int)b1[b] - Cq[b] + 1) >> 1;
            }
          }
        }
      }
      if (it + 1 < ntile) lstore(S[(it + 1) & 1], t0 + KM_ROWS);
      __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < KM_QB; b++) {
      const uint32_t ob0 = (uint32_t)__shfl_xor((int)b0[b], 32), ob1 = (uint32_t)__shfl_xor((int)b1[b], 32);
      const int oi0 = __shfl_xor(i0[b], 32), oi1 = __shfl_xor(i1[b], 32);
      const bool other_first = knn_before(ob0, oi0, b0[b], i0[b]);
      const uint32_t f_d = other_first ? ob0 : b0[b]; const int f_i = other_first ? oi0 : i0[b];
      const uint32_t c1d = other_first ? b0[b] : ob0; const int c1i = other_first ? i0[b] : oi0;
      const uint32_t c2d = other_first ? ob1 : b1[b]; const int c2i = other_first ? oi1 : i1[b];
      const bool two = knn_before(c2d, c2i, c1d, c1i);
      const int qi = q0 + wave * KM_WQ + 32 * b + r;
      if (h == 0 && qi < nq) {
        oidx[2 * qi] = f_i; oidx[2 * qi + 1] = two ? c2i : c1i;
        od2[2 * qi] = f_d; od2[2 * qi + 1] = two ? c2d : c1d;
      }
    }
  }
}

// BruteForce knnMatch(q, t, 2) on float32 descriptors (matching.py:102-108 with SIFT / SURF rows): hal::normL2Sqr_ --
// two 4-lane partial sums over steps of 8 elements, (d0 + d1) then its four lanes added left to right -- dist = sqrt,
// insertion on strictly smaller dist.  A workgroup owns 64 queries (thread = query, its row in registers); its four
// waves split the train rows into four consecutive ranges, each wave streams its range through its own double-buffered
// LDS tile (coalesced 16-byte loads one tile ahead, then every lane reads the same row: broadcast) and keeps the
// best two of its range; the ranges are merged in index order with the same strict insertion, which is the serial
// result (the best two under (distance, index)).  Round 3: the first form fed the train rows through scalar loads with
// one wave per SIMD and waited 2.4 us per train row (1.78 ms per 32 pairs of 749 x 749 rows).
#define KF_TILE 8                                    // train rows per staged tile
template <int DIM>
__global__ __launch_bounds__(256) void k_knn2_f32(EvhKnnF32Args A) {
  constexpr int R4 = DIM / 4;                        // float4 per row
  constexpr int PER_LANE = KF_TILE * R4 / 64;        // float4 a lane moves per tile
  __shared__ float4 s_t[4][2][KF_TILE * R4];
  __shared__ float s_b[3][2][64];
  __shared__ int s_i[3][2][64];
  const int p = blockIdx.y;
  int nq = A.nq, nt = A.nt;
  const float* Qb = A.q; const float* Tb = A.t;
  int32_t* oidx = A.idx; float* odist = A.dist;
  if (A.n_arr) {
    const int qs = A.q_slot0 + p * A.q_slot_step, ts = A.t_slot0 + p * A.t_slot_step;
    nq = A.n_arr[qs]; nt = A.n_arr[ts];
    Qb += (int64_t)qs * A.slot_floats; Tb += (int64_t)ts * A.slot_floats;
    oidx += (int64_t)p * A.out_stride * 2; odist += (int64_t)p * A.out_stride * 2;
  }
  if ((int)blockIdx.x * 64 >= nq) return;            // workgroup-uniform
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x * 64 + lane;
  const bool act = qi < nq;
  // two-element vectors: the compiler emits v_pk_add_f32 / v_pk_mul_f32 (two IEEE f32 operations per lane and issue slot,
  // no contraction); the pairs are neighbouring elements, so every one of the eight running sums still sees exactly
  // the operator's sequence of operations
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f q[DIM / 2];
  const float4* Q = reinterpret_cast<const float4*>(Qb + (int64_t)(act ? qi : 0) * DIM);
#pragma unroll
  for (int k = 0; k < R4; k++) { const float4 v = Q[k]; q[2 * k] = v2f{v.x, v.y}; q[2 * k + 1] = v2f{v.z, v.w}; }
  float b0 = FLT_MAX, b1 = FLT_MAX;
  int i0 = -1, i1 = -1;
  const int per = (nt + 3) >> 2, j0 = wave * per, j1 = min(nt, j0 + per);
  const int ntile = j1 > j0 ? (j1 - j0 + KF_TILE - 1) / KF_TILE : 0;
  const float4* T4 = reinterpret_cast<const float4*>(Tb);
  // the tile in flight: named registers (as an array -- through a lambda or a macro alike -- the compiler kept it in scratch)
  static_assert(PER_LANE == 2 || PER_LANE == 4, "64 or 128 floats per row");
  float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0, p3 = p0;
#define KF_LD(tile_, u) T4[(int64_t)min(j0 + (tile_) * KF_TILE + ((u) * 64 + lane) / R4, j1 - 1) * R4 + (((u) * 64 + lane) % R4)]
#define KF_LOAD_TILE(tile_)                                      /* rows past the range: its last row again (never used) */ \
  {                                                                                                                        \
    p0 = KF_LD(tile_, 0); p1 = KF_LD(tile_, 1);                                                                            \
    if constexpr (PER_LANE == 4) { p2 = KF_LD(tile_, 2); p3 = KF_LD(tile_, 3); }                                           \
  }
  if (ntile > 0) KF_LOAD_TILE(0)
  for (int tile = 0; tile < ntile; tile++) {
    float4* buf = s_t[wave][tile & 1];
    buf[lane] = p0; buf[64 + lane] = p1;
    if constexpr (PER_LANE == 4) { buf[128 + lane] = p2; buf[192 + lane] = p3; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tile + 1 < ntile) KF_LOAD_TILE(tile + 1)
    const int jb = j0 + tile * KF_TILE, cnt = min(KF_TILE, j1 - jb);
    for (int r = 0; r < cnt; r++) {
      const float4* row = buf + r * R4;
      v2f d0a = {0.f, 0.f}, d0b = {0.f, 0.f}, d1a = {0.f, 0.f}, d1b = {0.f, 0.f};   // d0[0..1], d0[2..3], d1[0..1], d1[2..3]
      // the row in batches of 16 words x 4: all reads of a batch requested before its arithmetic (the compiler alone waits
      // after every LDS read), a whole row at once would not fit the registers next to the query
      constexpr int HB = R4 < 16 ? R4 : 16;
#pragma unroll
      for (int h = 0; h < R4; h += HB) {
        float4 tv[HB];
#pragma unroll
        for (int k = 0; k < HB; k++) tv[k] = row[h + k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < HB; k += 2) {
          const v2f e0 = q[2 * (h + k)] - v2f{tv[k].x, tv[k].y}, e1 = q[2 * (h + k) + 1] - v2f{tv[k].z, tv[k].w};
          const v2f e2 = q[2 * (h + k) + 2] - v2f{tv[k + 1].x, tv[k + 1].y}, e3 = q[2 * (h + k) + 3] - v2f{tv[k + 1].z, tv[k + 1].w};
          d0a = d0a + e0 * e0;
          d0b = d0b + e1 * e1;
          d1a = d1a + e2 * e2;
          d1b = d1b + e3 * e3;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const float u0 = d0a.x + d1a.x, u1 = d0a.y + d1a.y, u2 = d0b.x + d1b.x, u3 = d0b.y + d1b.y;
      const float ds = sqrtf(u0 + u1 + u2 + u3);
      const int j = jb + r;
      if (ds < b1) {
        if (b0 > ds) { b1 = b0; i1 = i0; b0 = ds; i0 = j; }
        else { b1 = ds; i1 = j; }
      }
    }
  }
#undef KF_LOAD_TILE
#undef KF_LD
  // ranges in index order: wave 0 inserts the best two of waves 1, 2, 3 with the same rule
  if (wave > 0) { s_b[wave - 1][0][lane] = b0; s_i[wave - 1][0][lane] = i0; s_b[wave - 1][1][lane] = b1; s_i[wave - 1][1][lane] = i1; }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; w++)
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const float ds = s_b[w][e][lane];
        const int j = s_i[w][e][lane];
        if (j >= 0 && ds < b1) {
          if (b0 > ds) { b1 = b0; i1 = i0; b0 = ds; i0 = j; }
          else { b1 = ds; i1 = j; }
        }
      }
    if (act) {
      oidx[2 * qi] = i0; oidx[2 * qi + 1] = i1;
      odist[2 * qi] = b0; odist[2 * qi + 1] = b1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// ordered compaction helper: appends the flagged items of a 256-wide chunk in thread order.
__device__ __forceinline__ int block_ordered_slot(bool flag, int* wave_tot /*[4]*/, int& base) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long m = __ballot(flag);
  int in_wave = __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) wave_tot[wv] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
  for (int i = 0; i < 4; i++) { int c = wave_tot[i]; if (i < wv) off += c; tot += c; }
  int slot = base + off + in_wave;
  base += tot;
  return slot;
}

// ratio test + one-to-one filter + duplicate-coordinate filter; one workgroup per pair.
// GLOBAL: the five work arrays in a global scratch (A.work, 5 * kcap ints per pair) instead of LDS -- for key-point budgets
// above 7 680 per frame (SIFT on frames wider than ~500 px), where 5 * kcap ints no longer fit a compute unit's LDS.  A
// template, not a runtime pointer choice: generic pointers would turn the LDS form's accesses into FLAT instructions.
template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_filter(EvhFilterArgs A) {
  extern __shared__ uint32_t dyn[];
  // work arrays: claims[kcap] | mq[kcap] | mt[kcap] | keep[kcap] | lastj[kcap]
  int* claims;
  if constexpr (GLOBAL) claims = A.work + (int64_t)blockIdx.x * (5 * A.kcap + 2);      // + hdr[2]: survivors, second stage wanted
  else claims = reinterpret_cast<int*>(dyn);
  int* mq = claims + A.kcap;
  int* mt = mq + A.kcap;
  int* keep = mt + A.kcap;
  int* lastj = keep + A.kcap;
  __shared__ int wave_tot[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int qs = A.q_slot0 + p * A.q_slot_step, ts = A.t_slot0 + p * A.t_slot_step;
  const int nq = A.nq_arr ? A.nq_arr[qs] : A.nq_fixed;
  const int nt = A.nt_arr ? A.nt_arr[ts] : A.nt_fixed;
  const int32_t* idx = A.idx + (int64_t)p * A.knn_stride * 2;
  const uint32_t* d2 = A.d2 + (int64_t)p * A.knn_stride * 2;
  const float* xyq = A.xy_q + (int64_t)qs * A.xy_slot_floats;
  const float* xyt = A.xy_t + (int64_t)ts * A.xy_slot_floats;
  float* out = A.pts + (int64_t)p * A.pts_stride * 4;
  if constexpr (GLOBAL) { if (tid == 0) claims[5 * A.kcap + 1] = 0; }     // hdr[1]: no second stage unless the survivors get that far
  if ((A.flags_arr && (A.flags_arr[qs] | A.flags_arr[ts])) != 0) {
    if (tid == 0) { A.npts[p] = 0; A.status[p] = EVH_PAIR_CAPACITY; }
    return;
  }
  if (nq == 0 || nt == 0) {  // detectAndCompute returned descriptors None (matching.py:104-107)
    if (tid == 0) { A.npts[p] = 0; A.status[p] = EVH_PAIR_NO_DESCRIPTORS; }
    return;
  }
  for (int i = tid; i < nt; i += 256) claims[i] = 0;
  __syncthreads();
  // Lowe's ratio on sqrt distances: (double)sqrtf(D0) < (double)sqrtf(D1) * ratio   (matching.py:190)
  for (int i = tid; i < nq; i += 256) {
    bool pass = false;
    if (idx[2 * i] >= 0 && idx[2 * i + 1] >= 0) {
      double dist0 = A.d2_is_dist ? (double)__uint_as_float(d2[2 * i]) : (double)sqrtf((float)d2[2 * i]);
      double dist1 = A.d2_is_dist ? (double)__uint_as_float(d2[2 * i + 1]) : (double)sqrtf((float)d2[2 * i + 1]);
      pass = dist0 < dist1 * A.ratio;
    }
    keep[i] = pass ? 1 : 0;
    if (pass) atomicAdd(&claims[idx[2 * i]], 1);
  }
  __syncthreads();
  // survivors in ascending query order whose train index is claimed exactly once (matching.py:228-238)
  int m = 0;
  for (int c0 = 0; c0 < nq; c0 += 256) {
    int i = c0 + tid;
    bool f = i < nq && keep[i] && claims[idx[2 * i]] == 1;
    int slot = block_ordered_slot(f, wave_tot, m);
    if (f) { mq[slot] = i; mt[slot] = idx[2 * i]; }
  }
  __syncthreads();
  if (m < A.min_matches) {  // matching.py:113
    if (tid == 0) { A.npts[p] = 0; A.status[p] = EVH_PAIR_FEW_MATCHES; }
    return;
  }
  // remove_double_matching: key = exact (ax, ay); first occurrence keeps its place, last occurrence gives b.  The x
  // coordinates of the m survivors are staged in LDS (the claim counters are dead); a thread walks them eight at a time
  // and looks at y -- a gather from global memory -- only where x matches.  (Both coordinates gathered from global memory
  // in the inner loop cost 1.4 ms on 1 000 SIFT survivors.)
  if constexpr (GLOBAL) {
    // thousands of survivors: the quadratic duplicate search is spread over workgroups (k_filter_dup) and k_filter_out
    // finishes the pair; hdr[0..1] behind the five arrays carry the survivor count across the launches
    if (tid == 0) { claims[5 * A.kcap] = m; claims[5 * A.kcap + 1] = 1; }
    return;
  }
  float* sx = reinterpret_cast<float*>(claims);
  for (int i = tid; i < m; i += 256) sx[i] = xyq[2 * mq[i]];
  __syncthreads();
  for (int i = tid; i < m; i += 256) {
    const float ax = sx[i], ay = xyq[2 * mq[i] + 1];
    int first = 1, last = i;
    int j = 0;
    for (; j + 8 <= m; j += 8) {
      float bx[8];
#pragma unroll
      for (int u = 0; u < 8; u++) bx[u] = sx[j + u];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (bx[u] == ax && xyq[2 * mq[j + u] + 1] == ay) { if (j + u < i) first = 0; if (j + u > last) last = j + u; }
    }
    for (; j < m; j++)
      if (sx[j] == ax && xyq[2 * mq[j] + 1] == ay) { if (j < i) first = 0; if (j > last) last = j; }
    keep[i] = first; lastj[i] = last;
  }
  __syncthreads();
  int u = 0;
  for (int c0 = 0; c0 < m; c0 += 256) {
    int i = c0 + tid;
    bool f = i < m && keep[i];
    int slot = block_ordered_slot(f, wave_tot, u);
    if (f) {
      int tq = mq[i], tt = mt[lastj[i]];
      out[4 * slot] = xyq[2 * tq]; out[4 * slot + 1] = xyq[2 * tq + 1];
      out[4 * slot + 2] = xyt[2 * tt]; out[4 * slot + 3] = xyt[2 * tt + 1];
    }
  }
  if (tid == 0) { A.npts[p] = u; A.status[p] = EVH_PAIR_OK; }
}


// frame_processing.py:91-98: the static rows of one feature type are appended to the pair's concatenation; a type that
// raised (NoMatchesException) fails the pair with ITS status and nothing after it counts (the exception propagates)
__global__ __launch_bounds__(256) void k_accumulate(EvhAccArgs A) {
  const int p = blockIdx.x;
  __shared__ int s_base, s_go;
  if (threadIdx.x == 0) {
    int st = A.first ? 0 : A.accstatus[p];
    int base = A.first ? 0 : A.nacc[p];
    int go = 0;
    if (st == 0) {
      if (A.status[p] != 0) { st = A.status[p]; base = 0; }
      else go = 1;
    }
    s_base = base; s_go = go;
    A.accstatus[p] = st;
    A.nacc[p] = go ? base + A.nrows[p] : base;
  }
  __syncthreads();
  if (!s_go) return;
  const int n = A.nrows[p], base = s_base;
  const float4* src = reinterpret_cast<const float4*>(A.rows + (int64_t)p * A.row_stride * 4);
  float4* dst = reinterpret_cast<float4*>(A.acc + (int64_t)p * A.acc_stride * 4);
  for (int i = threadIdx.x; i < n; i += 256)
    if (base + i < A.acc_stride) dst[base + i] = src[i];
}

// second stage of k_filter<true>: remove_double_matching's duplicate search, 256 survivors per workgroup against all of
// them (x coordinates staged through an LDS tile, y fetched only where x matches); grid (chunks, pairs)
__global__ __launch_bounds__(256) void k_filter_dup(EvhFilterArgs A) {
  __shared__ float s_x[2048];
  const int p = blockIdx.y, tid = threadIdx.x;
  int* w = A.work + (int64_t)p * (5 * A.kcap + 2);
  const int* hdr = w + 5 * A.kcap;
  if (!hdr[1]) return;
  const int m = hdr[0];
  if ((int)blockIdx.x * 256 >= m) return;
  const int* mq = w + A.kcap;
  int* keep = w + 3 * A.kcap;
  int* lastj = w + 4 * A.kcap;
  const int qs = A.q_slot0 + p * A.q_slot_step;
  const float* xyq = A.xy_q + (int64_t)qs * A.xy_slot_floats;
  const int i = blockIdx.x * 256 + tid;
  const bool act = i < m;
  const float ax = act ? xyq[2 * mq[i]] : 0.f, ay = act ? xyq[2 * mq[i] + 1] : 0.f;
  int first = 1, last = i;
  for (int t0 = 0; t0 < m; t0 += 2048) {
    const int tn = min(2048, m - t0);
    __syncthreads();
    for (int k = tid; k < tn; k += 256) s_x[k] = xyq[2 * mq[t0 + k]];
    __syncthreads();
    if (act) {
      int j = 0;
      for (; j + 8 <= tn; j += 8) {
        float bx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) bx[u] = s_x[j + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (bx[u] == ax && xyq[2 * mq[t0 + j + u] + 1] == ay) { if (t0 + j + u < i) first = 0; if (t0 + j + u > last) last = t0 + j + u; }
      }
      for (; j < tn; j++)
        if (s_x[j] == ax && xyq[2 * mq[t0 + j] + 1] == ay) { if (t0 + j < i) first = 0; if (t0 + j > last) last = t0 + j; }
    }
  }
  if (act) { keep[i] = first; lastj[i] = last; }
}

// third stage: the kept survivors in order -> the matched rows of the pair
__global__ __launch_bounds__(256) void k_filter_out(EvhFilterArgs A) {
  __shared__ int wave_tot[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int* w = A.work + (int64_t)p * (5 * A.kcap + 2);
  const int* hdr = w + 5 * A.kcap;
  if (!hdr[1]) return;                               // the first stage has already written the pair's status
  const int m = hdr[0];
  const int* mq = w + A.kcap; const int* mt = w + 2 * A.kcap; const int* keep = w + 3 * A.kcap; const int* lastj = w + 4 * A.kcap;
  const int qs = A.q_slot0 + p * A.q_slot_step, ts = A.t_slot0 + p * A.t_slot_step;
  const float* xyq = A.xy_q + (int64_t)qs * A.xy_slot_floats;
  const float* xyt = A.xy_t + (int64_t)ts * A.xy_slot_floats;
  float* out = A.pts + (int64_t)p * A.pts_stride * 4;
  int u = 0;
  for (int c0 = 0; c0 < m; c0 += 256) {
    int i = c0 + tid;
    bool f = i < m && keep[i];
    int slot = block_ordered_slot(f, wave_tot, u);
    if (f) {
      int tq = mq[i], tt = mt[lastj[i]];
      out[4 * slot] = xyq[2 * tq]; out[4 * slot + 1] = xyq[2 * tq + 1];
      out[4 * slot + 2] = xyt[2 * tt]; out[4 * slot + 3] = xyt[2 * tt + 1];
    }
  }
  if (tid == 0) { A.npts[p] = u; A.status[p] = EVH_PAIR_OK; }
}

// frame_processing.py:102-104: remove_double_matching over the concatenated rows of all feature types (utils.py:60-68):
// key = exact (ax, ay), first occurrence keeps its place, the LAST occurrence gives b
// stage 1 (grid: chunks of 256 rows x pairs): for every row whether it is the first of its (ax, ay) key and the index of
// the key's last row -> A.work ([pair][acc_stride][2]); every workgroup walks ALL rows of its pair through an LDS tile.
// (One workgroup per pair for the whole quadratic search took 27 ms on the 15 000 rows of a 720p SIFT + SURF + ORB pair.)
__global__ __launch_bounds__(256) void k_merge_dup(EvhMergeArgs A) {
  __shared__ float2 s_xy[1024];
  const int p = blockIdx.y, tid = threadIdx.x;
  if (A.accstatus[p] != 0) return;
  const int m = min(A.nacc[p], (int)A.acc_stride);
  if ((int)blockIdx.x * 256 >= m) return;
  const float4* R = reinterpret_cast<const float4*>(A.acc + (int64_t)p * A.acc_stride * 4);
  int* W = A.work + (int64_t)p * A.acc_stride * 2;
  const int i = blockIdx.x * 256 + tid;
  const bool act = i < m;
  bool first = true; int last = i;
  const float ax = act ? R[i].x : 0.f, ay = act ? R[i].y : 0.f;
  for (int t0 = 0; t0 < m; t0 += 1024) {
    const int tn = min(1024, m - t0);
    __syncthreads();
    for (int k = tid; k < tn; k += 256) { const float4 o = R[t0 + k]; s_xy[k] = make_float2(o.x, o.y); }
    __syncthreads();
    if (act) {
      int j = 0;
      for (; j + 8 <= tn; j += 8) {
        float2 o[8];
#pragma unroll
        for (int q = 0; q < 8; q++) o[q] = s_xy[j + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (o[q].x == ax && o[q].y == ay) { if (t0 + j + q < i) first = false; if (t0 + j + q > last) last = t0 + j + q; }
      }
      for (; j < tn; j++) {
        const float2 o = s_xy[j];
        if (o.x == ax && o.y == ay) { if (t0 + j < i) first = false; if (t0 + j > last) last = t0 + j; }
      }
    }
  }
  if (act) { W[2 * i] = first ? 1 : 0; W[2 * i + 1] = last; }
}

// stage 2 (one workgroup per pair): the first rows in order, each with the b of its key's last row
__global__ __launch_bounds__(256) void k_merge(EvhMergeArgs A) {
  __shared__ int wave_tot[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int st = A.accstatus[p];
  if (st != 0) {
    if (tid == 0) { A.nout[p] = 0; A.status[p] = st; }
    return;
  }
  const int m = min(A.nacc[p], (int)A.acc_stride);
  const float4* R = reinterpret_cast<const float4*>(A.acc + (int64_t)p * A.acc_stride * 4);
  const int* W = A.work + (int64_t)p * A.acc_stride * 2;
  float4* out = reinterpret_cast<float4*>(A.out + (int64_t)p * A.out_stride * 4);
  int u = 0;
  for (int c0 = 0; c0 < m; c0 += 256) {          // workgroup-uniform
    const int i = c0 + tid;
    const bool first = i < m && W[2 * i] != 0;
    const int slot = block_ordered_slot(first, wave_tot, u);
    if (first && slot < A.out_stride) { const float4 a = R[i], b = R[W[2 * i + 1]]; out[slot] = make_float4(a.x, a.y, b.z, b.w); }
  }
  if (tid == 0) { A.nout[p] = u; A.status[p] = 0; }
}

}  // namespace

int evh_launch_knn2(evh_ctx* c, const EvhKnnArgs& A, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  // chunks of 256 queries in grid.y, bounded by the largest possible query count
  const int nq_max = A.nq_arr ? A.out_stride : A.nq_fixed;
  const int chunks = std::max(1, std::min((nq_max + 255) / 256, 64));
  static const bool dot4_form = getenv("EVH_KNN_DOT4") != nullptr;      // A/B switch: the v_dot4 kernel for 128-byte rows too
  if (A.desc_bytes == 128 && !A.hamming && !dot4_form)
    hipLaunchKernelGGL(k_knn2_mfma128, dim3(npairs, std::max(1, std::min((nq_max + KM_GQ - 1) / KM_GQ, 256))), dim3(256), 0, c->stream, A);
  else if (A.desc_bytes == 128) hipLaunchKernelGGL((k_knn2<8, 128>), dim3(npairs, chunks), dim3(256), 0, c->stream, A);
  else hipLaunchKernelGGL((k_knn2<2, MT_TILE>), dim3(npairs, chunks), dim3(256), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_knn2_f32(evh_ctx* c, const EvhKnnF32Args& A, int npairs) {
  const int nq_max = A.n_arr ? (int)A.out_stride : A.nq;
  if (nq_max <= 0 || npairs <= 0) return EVH_SUCCESS;
  const dim3 grid((nq_max + 63) / 64, npairs);
  if (A.dim == 128) hipLaunchKernelGGL(k_knn2_f32<128>, grid, dim3(256), 0, c->stream, A);
  else if (A.dim == 64) hipLaunchKernelGGL(k_knn2_f32<64>, grid, dim3(256), 0, c->stream, A);
  else return evh_fail(c, EVH_ERR_UNSUPPORTED, "float descriptors: 64 or 128 elements per row (SURF / SIFT)");
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_accumulate(evh_ctx* c, const EvhAccArgs& A, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  hipLaunchKernelGGL(k_accumulate, dim3(npairs), dim3(256), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_merge(evh_ctx* c, const EvhMergeArgs& A_, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  EvhMergeArgs A = A_;
  const size_t need = sizeof(int) * 2 * (size_t)A.acc_stride * (size_t)npairs;
  if (c->merge_ws_bytes < need) {
    if (c->d_merge_ws) { EVH_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_merge_ws); c->d_merge_ws = nullptr; c->merge_ws_bytes = 0; }
    EVH_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_merge_ws), need));
    c->merge_ws_bytes = need;
  }
  A.work = c->d_merge_ws;
  hipLaunchKernelGGL(k_merge_dup, dim3((unsigned)((A.acc_stride + 255) / 256), npairs), dim3(256), 0, c->stream, A);
  hipLaunchKernelGGL(k_merge, dim3(npairs), dim3(256), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_filter(evh_ctx* c, const EvhFilterArgs& A_, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  EvhFilterArgs A = A_;
  size_t lds = sizeof(int) * 5 * (size_t)A.kcap;
  if (lds > EVH_FILTER_LDS_MAX) {          // beyond a compute unit's LDS: the work arrays in a global scratch, grown on demand
    const size_t need = (lds + 2 * sizeof(int)) * (size_t)npairs;
    if (c->filter_ws_bytes < need) {
      if (c->d_filter_ws) { EVH_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_filter_ws); c->d_filter_ws = nullptr; c->filter_ws_bytes = 0; }
      EVH_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_filter_ws), need));
      c->filter_ws_bytes = need;
    }
    A.work = c->d_filter_ws;
    hipLaunchKernelGGL(k_filter<true>, dim3(npairs), dim3(256), 0, c->stream, A);
    hipLaunchKernelGGL(k_filter_dup, dim3((A.kcap + 255) / 256, npairs), dim3(256), 0, c->stream, A);
    hipLaunchKernelGGL(k_filter_out, dim3(npairs), dim3(256), 0, c->stream, A);
    EVH_HIP(c, hipGetLastError());
    return EVH_SUCCESS;
  }
  if (lds > 48 * 1024)   // large key-point budgets (N = 4000 -> ~105 KB): opt in to more dynamic LDS than the default
    EVH_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
  hipLaunchKernelGGL(k_filter<false>, dim3(npairs), dim3(256), lds, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
