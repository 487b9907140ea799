// evh_match.hip -- brute-force 2-NN over 32-byte descriptors + the reference's match filters, gfx950.
// Replaces cv2.DescriptorMatcher_create("BruteForce").knnMatch(q, t, 2) (matching.py:102-108) and the glue
// lowes_ratio_test / filter_corresponding_points (matching.py:166-239) / remove_double_matching (utils.py:41-68).
// Distances are exact integers: D = |q|^2 + |t|^2 - 2 q.t with v_dot4_u32_u8 on LDS-staged train tiles.
#include "evh_internal.h"
#include "evh_match.h"

namespace {

#define MT_TILE 512  // train descriptors per LDS tile (16 KB)

__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t acc) {
#if __has_builtin(__builtin_amdgcn_udot4)
  return __builtin_amdgcn_udot4(a, b, acc, false);
#else
  return acc + (a & 0xFF) * (b & 0xFF) + ((a >> 8) & 0xFF) * ((b >> 8) & 0xFF) + ((a >> 16) & 0xFF) * ((b >> 16) & 0xFF) +
         (a >> 24) * (b >> 24);
#endif
}

// one workgroup (256 threads) per (pair, chunk of 256 queries): blockIdx.y = chunk, so that a few pairs with many
// descriptors (4K frames, N = 4000) still fill the chip; the train set is streamed through LDS by every chunk
__global__ __launch_bounds__(256) void k_knn2(EvhKnnArgs A) {
  __shared__ uint4 tdesc[MT_TILE * 2];
  __shared__ uint32_t tnorm[MT_TILE];
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
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (act) { qa = Q[2 * qi]; qb = Q[2 * qi + 1]; }
    uint32_t qn = 0;
    if (!A.hamming) {
      qn = dot4(qa.x, qa.x, qn); qn = dot4(qa.y, qa.y, qn); qn = dot4(qa.z, qa.z, qn); qn = dot4(qa.w, qa.w, qn);
      qn = dot4(qb.x, qb.x, qn); qn = dot4(qb.y, qb.y, qn); qn = dot4(qb.z, qb.z, qn); qn = dot4(qb.w, qb.w, qn);
    }
    uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
    int i0 = -1, i1 = -1;
    for (int t0 = 0; t0 < nt; t0 += MT_TILE) {
      const int tn = min(MT_TILE, nt - t0);
      __syncthreads();
      for (int i = tid; i < tn * 2; i += 256) tdesc[i] = T[2 * t0 + i];
      __syncthreads();
      if (!A.hamming)
        for (int i = tid; i < tn; i += 256) {
          uint4 a = tdesc[2 * i], b = tdesc[2 * i + 1];
          uint32_t n = 0;
          n = dot4(a.x, a.x, n); n = dot4(a.y, a.y, n); n = dot4(a.z, a.z, n); n = dot4(a.w, a.w, n);
          n = dot4(b.x, b.x, n); n = dot4(b.y, b.y, n); n = dot4(b.z, b.z, n); n = dot4(b.w, b.w, n);
          tnorm[i] = n;
        }
      __syncthreads();
      if (act) {
        for (int j = 0; j < tn; j++) {
          const uint4 ta = tdesc[2 * j], tb = tdesc[2 * j + 1];
          uint32_t d;
          if (A.hamming) {
            d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
          } else {
            uint32_t s = 0;
            s = dot4(qa.x, ta.x, s); s = dot4(qa.y, ta.y, s); s = dot4(qa.z, ta.z, s); s = dot4(qa.w, ta.w, s);
            s = dot4(qb.x, tb.x, s); s = dot4(qb.y, tb.y, s); s = dot4(qb.z, tb.z, s); s = dot4(qb.w, tb.w, s);
            d = qn + tnorm[j] - 2u * s;
          }
          // ascending train order, strict '<': ties keep the lowest train index
          if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = t0 + j; }
          else if (d < b1) { b1 = d; i1 = t0 + j; }
        }
      }
    }
    if (act) {
      oidx[2 * qi] = i0; oidx[2 * qi + 1] = i1;
      od2[2 * qi] = b0; od2[2 * qi + 1] = b1;
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
__global__ __launch_bounds__(256) void k_filter(EvhFilterArgs A) {
  extern __shared__ uint32_t dyn[];
  // dynamic LDS layout: claims[kcap] | mq[kcap] | mt[kcap] | keep[kcap] | lastj[kcap]
  int* claims = reinterpret_cast<int*>(dyn);
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
      double dist0 = (double)sqrtf((float)d2[2 * i]);
      double dist1 = (double)sqrtf((float)d2[2 * i + 1]);
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
  // remove_double_matching: key = exact (ax, ay); first occurrence keeps its place, last occurrence gives b
  for (int i = tid; i < m; i += 256) {
    float ax = xyq[2 * mq[i]], ay = xyq[2 * mq[i] + 1];
    int first = 1, last = i;
    for (int j = 0; j < m; j++) {
      float bx = xyq[2 * mq[j]], by = xyq[2 * mq[j] + 1];
      if (bx == ax && by == ay) { if (j < i) first = 0; if (j > last) last = j; }
    }
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

}  // namespace

int evh_launch_knn2(evh_ctx* c, const EvhKnnArgs& A, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  // chunks of 256 queries in grid.y, bounded by the largest possible query count
  const int nq_max = A.nq_arr ? A.out_stride : A.nq_fixed;
  const int chunks = std::max(1, std::min((nq_max + 255) / 256, 64));
  hipLaunchKernelGGL(k_knn2, dim3(npairs, chunks), dim3(256), 0, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}

int evh_launch_filter(evh_ctx* c, const EvhFilterArgs& A, int npairs) {
  if (npairs <= 0) return EVH_SUCCESS;
  size_t lds = sizeof(int) * 5 * (size_t)A.kcap;
  if (lds > 48 * 1024)   // large key-point budgets (N = 4000 -> ~105 KB): opt in to more dynamic LDS than the default
    EVH_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
  hipLaunchKernelGGL(k_filter, dim3(npairs), dim3(256), lds, c->stream, A);
  EVH_HIP(c, hipGetLastError());
  return EVH_SUCCESS;
}
