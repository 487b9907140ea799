// micro-benchmark: latency of dependent f64 adds / muls / fma, rcp, sqrt and of an LDS round trip, one wave (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned long long* cyc, double seed) {
  __shared__ double L[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) L[i] = seed + i;
  __syncthreads();
  double s = seed, a = seed * 0.5;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 256; i++) s += a;
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("" :: "v"(s));
#pragma unroll
  for (int i = 0; i < 256; i++) s = s * a;
  unsigned long long t2 = __builtin_readcyclecounter();
  asm volatile("" :: "v"(s));
#pragma unroll
  for (int i = 0; i < 256; i++) s = __builtin_fma(s, a, a);
  unsigned long long t3 = __builtin_readcyclecounter();
  asm volatile("" :: "v"(s));
  int idx = threadIdx.x & 7;
#pragma unroll
  for (int i = 0; i < 64; i++) { idx = (int)L[idx] & 1023; }          // dependent LDS reads
  unsigned long long t4 = __builtin_readcyclecounter();
  asm volatile("" :: "v"(idx));
  double v[32];
#pragma unroll
  for (int i = 0; i < 32; i++) v[i] = L[(threadIdx.x & 7) + 46 * i];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 32; i++) s += v[i];
  unsigned long long t5 = __builtin_readcyclecounter();
  asm volatile("" :: "v"(s));
  float f = (float)seed;
#pragma unroll
  for (int i = 0; i < 256; i++) f += 1.5f;
  unsigned long long t6 = __builtin_readcyclecounter();
  out[threadIdx.x] = s + idx + f;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; }
}
int main() {
  double* o; unsigned long long* c;
  hipMalloc(&o, 64 * 8); hipMalloc(&c, 64);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0000001);
  unsigned long long h[6]; hipMemcpy(h, c, 48, hipMemcpyDeviceToHost);
  printf("cycles (s_memtime units, 100 MHz? see ratio): add256 %llu mul256 %llu fma256 %llu ldsdep64 %llu load32+add32 %llu f32add256 %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  return 0;
}
