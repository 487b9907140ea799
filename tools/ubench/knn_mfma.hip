// Measurement only (not part of libevhip.so): the 2-NN of float32[N,128] descriptors with the distance matrix in its Gram
// form |q|^2 + |t|^2 - 2 q.t, q.t on the f32-input matrix cores (v_mfma_f32_32x32x2_f32) -- what VERDICT r2 item 1 asked to
// be measured against the exact-order VALU form the library ships (k_knn2_f32).  Built by tools/knn_mfma_probe.py's
// instructions into tools/ubench/knn_mfma.so; entry: knn_mfma_run.
//
// One wave owns 32 queries (B operand, 64 registers per lane, fixed), the workgroup's four waves share a 32-row train
// tile in LDS (A operand), 64 MFMAs per tile and wave; C/D layout: column = lane & 31 = query, row = (reg & 3) +
// 8 * (reg >> 2) + 4 * (lane >> 5) = train row of the tile.  Untuned (one LDS read per MFMA, no software pipelining).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_norms(const float* __restrict__ X, int n, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < 128; k++) s += X[(int64_t)i * 128 + k] * X[(int64_t)i * 128 + k];
  out[i] = s;
}

__global__ __launch_bounds__(256) void k_knn2_mfma(const float* __restrict__ Q, const float* __restrict__ T,
                                                   const float* __restrict__ qn2, const float* __restrict__ tn2, int nq, int nt,
                                                   int pair_stride_rows, int32_t* __restrict__ idx, float* __restrict__ dist) {
  __shared__ float sT[32][129];
  const int p = blockIdx.y;
  Q += (int64_t)p * pair_stride_rows * 128; T += (int64_t)p * pair_stride_rows * 128;
  qn2 += (int64_t)p * pair_stride_rows; tn2 += (int64_t)p * pair_stride_rows;
  idx += (int64_t)p * pair_stride_rows * 2; dist += (int64_t)p * pair_stride_rows * 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int q = (blockIdx.x * 4 + wave) * 32 + col, qc = min(q, nq - 1);
  float bq[64];
#pragma unroll
  for (int kk = 0; kk < 64; kk++) bq[kk] = Q[(int64_t)qc * 128 + 2 * kk + half];
  const float myq = qn2[qc];
  float b0 = FLT_MAX, b1 = FLT_MAX;
  int i0 = -1, i1 = -1;
  for (int tb = 0; tb < nt; tb += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * 128; e += 256) {
      const int r = e >> 7, c = e & 127;
      sT[r][c] = T[(int64_t)min(tb + r, nt - 1) * 128 + c];
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 64; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sT[col][2 * kk + half], bq[kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int i = tb + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < nt) {
        const float d2 = (myq + tn2[i]) - 2.f * acc[r];
        if (d2 < b1 || (d2 == b1 && i < i1)) {
          if (d2 < b0 || (d2 == b0 && i < i0)) { b1 = b0; i1 = i0; b0 = d2; i0 = i; }
          else { b1 = d2; i1 = i; }
        }
      }
    }
  }
  // the other half of the wave holds the other rows of every tile for the same query
  const float ob0 = __shfl_xor(b0, 32), ob1 = __shfl_xor(b1, 32);
  const int oi0 = __shfl_xor(i0, 32), oi1 = __shfl_xor(i1, 32);
  auto ins = [&](float d2, int i) {
    if (i < 0) return;
    if (d2 < b1 || (d2 == b1 && i < i1)) {
      if (d2 < b0 || (d2 == b0 && i < i0)) { b1 = b0; i1 = i0; b0 = d2; i0 = i; }
      else { b1 = d2; i1 = i; }
    }
  };
  ins(ob0, oi0); ins(ob1, oi1);
  if (half == 0 && q < nq) {
    idx[2 * q] = i0; idx[2 * q + 1] = i1;
    dist[2 * q] = sqrtf(fmaxf(b0, 0.f)); dist[2 * q + 1] = sqrtf(fmaxf(b1, 0.f));
  }
}

// q, t: [npairs][stride_rows][128] floats on the device; idx / dist: [npairs][stride_rows][2]; norms: scratch [2][npairs * stride_rows]
extern "C" int knn_mfma_run(const float* q, const float* t, int nq, int nt, int npairs, int stride_rows, float* norms,
                            int32_t* idx, float* dist, int reps, float* ms_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
  const int total = npairs * stride_rows;
  float* qn = norms; float* tn = norms + total;
  for (int r = 0; r < reps + 1; r++) {
    if (r == 1) (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_norms, dim3((total + 255) / 256), dim3(256), 0, s, q, total, qn);
    hipLaunchKernelGGL(k_norms, dim3((total + 255) / 256), dim3(256), 0, s, t, total, tn);
    hipLaunchKernelGGL(k_knn2_mfma, dim3((nq + 127) / 128, npairs), dim3(256), 0, s, q, t, qn, tn, nq, nt, stride_rows, idx, dist);
  }
  (void)hipEventRecord(e1, s);
  if (hipEventSynchronize(e1) != hipSuccess) return -2;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
