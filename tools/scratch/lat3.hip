#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long T(double dep) {
  int r = __builtin_amdgcn_readfirstlane((int)__double2loint(dep));
  asm volatile("s_nop 0" :: "s"(r));
  return __builtin_readcyclecounter();
}
__global__ void k(double* out, unsigned long long* cyc, double seed) {
  double s = seed, a = seed * 0.5, b = seed * 0.25;
  unsigned long long t[8];
  t[0] = T(s);
#pragma unroll
  for (int i = 0; i < 64; i++) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(a)); }
  t[1] = T(s);
#pragma unroll
  for (int i = 0; i < 64; i++) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(s) : "v"(a)); }
  t[2] = T(s);
  double s0 = s, s1 = a, s2 = b, s3 = seed;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    asm volatile("v_add_f64 %0, %0, %1" : "+v"(s0) : "v"(a)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(s1) : "v"(a));
    asm volatile("v_add_f64 %0, %0, %1" : "+v"(s2) : "v"(a)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(s3) : "v"(a));
  }
  s = s0 + s1 + s2 + s3;
  t[3] = T(s);
  float f = (float)seed, g = 1.5f;
#pragma unroll
  for (int i = 0; i < 64; i++) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(g)); }
  t[4] = T((double)f);
  out[threadIdx.x] = s + f;
  if (threadIdx.x == 0) for (int i = 0; i < 4; i++) cyc[i] = t[i + 1] - t[i];
}
int main() {
  double* o; unsigned long long* c;
  (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 128);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0000001);
  unsigned long long h[8]; (void)hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  printf("64 dependent v_add_f64: %llu | 64 dependent v_mul_f64: %llu | 64 v_add_f64 in 4 chains: %llu | 64 dependent v_add_f32: %llu\n", h[0], h[1], h[2], h[3]);
  return 0;
}
