#include <hip/hip_runtime.h>
#include <cstdio>
#define T() __builtin_readcyclecounter()
__global__ void k(double* out, unsigned long long* cyc, double seed) {
  __shared__ double L[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) L[i] = seed + i;
  __syncthreads();
  double s = seed;
  unsigned long long t[12];
  t[0] = T(); t[1] = T();
  double v[32];
  // A: 32 strided 8-byte loads, then wait
#pragma unroll
  for (int i = 0; i < 32; i++) v[i] = L[(threadIdx.x & 7) + 46 * i];
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  t[2] = T();
  // B: 32 dependent adds on registers
#pragma unroll
  for (int i = 0; i < 32; i++) { asm volatile("" : "+v"(v[i])); }
  __builtin_amdgcn_sched_barrier(0);
  t[3] = T();
#pragma unroll
  for (int i = 0; i < 32; i++) s += v[i];
  asm volatile("" : "+v"(s));
  t[4] = T();
  // C: only lanes 0..10 active (as wave 3 of the eval), same loads
  double w[32];
  if (threadIdx.x < 11) {
#pragma unroll
    for (int i = 0; i < 32; i++) w[i] = L[24 + 2 * threadIdx.x + 46 * i];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 32; i++) s += w[i];
  }
  asm volatile("" : "+v"(s));
  t[5] = T();
  // D: 46 lanes: two loads (term reads) + product + store, 16 points (wave 2 of the eval)
  if (threadIdx.x < 46) {
    double va[16], vb[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { va[q] = L[q * 10 + (threadIdx.x % 10)]; vb[q] = L[q * 10 + (threadIdx.x % 7)]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; q++) L[1024 + q * 46 + threadIdx.x] = va[q] * vb[q];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  t[6] = T();
  __syncthreads();
  t[7] = T();
  out[threadIdx.x] = s + L[1024 + threadIdx.x];
  if (threadIdx.x == 0) for (int i = 0; i < 7; i++) cyc[i] = t[i + 1] - t[i];
}
int main() {
  double* o; unsigned long long* c;
  (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 128);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0000001);
  unsigned long long h[8]; (void)hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  printf("timer %llu | 32 loads+wait %llu | (opaque) %llu | 32 dependent f64 adds %llu | 11 lanes: 32 loads + 32 adds %llu | 46 lanes: 32 loads, 16 mul, 16 stores %llu | barrier %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
  return 0;
}
