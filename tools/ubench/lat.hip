// Dependent-chain latency microbenchmark for gfx950, ONE wavefront: cycles per dependent operation for the
// pieces the serial homography solver is made of (f64 fma / div / sqrt, LDS round trips, DPP, readlane, bpermute,
// barrier).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//                   lat.hip -o lat ; run: ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define ITERS 2048

__device__ __forceinline__ long long now() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ long long wall() { return wall_clock64(); }

struct Res { long long cyc, wall; double sink; };

#define BENCH_BEGIN(name)                                                                        \
  __global__ __launch_bounds__(64) void k_##name(Res* r, double seed, int iseed) {               \
    __shared__ double lds[1024];                                                                \
    const int lane = threadIdx.x;                                                               \
    double x = seed + lane * 1e-3, y = seed * 0.5 + 1.25;                                         \
    int ix = iseed + lane;                                                                      \
    lds[lane] = x; lds[lane + 64] = y;                                                          \
    __syncthreads();                                                                            \
    const long long w0 = wall(), t0 = now();                                                    \
    for (int it = 0; it < ITERS; it++) {
#define BENCH_END                                                                                \
    }                                                                                           \
    const long long t1 = now(), w1 = wall();                                                    \
    if (lane == 0) { r->cyc = t1 - t0; r->wall = w1 - w0; r->sink = x + y + ix + lds[(ix & 63)]; } \
  }

BENCH_BEGIN(empty) asm volatile("" : "+v"(x)); BENCH_END
BENCH_BEGIN(fma64) x = fma(x, y, 1e-9); BENCH_END
BENCH_BEGIN(mul64) x = x * y; BENCH_END
BENCH_BEGIN(add64) x = x + y; BENCH_END
BENCH_BEGIN(div64) x = y / x + 1.5; BENCH_END
BENCH_BEGIN(div64only) x = 3.0 / x; BENCH_END
BENCH_BEGIN(sqrt64) x = sqrt(x) + 1.5; BENCH_END
BENCH_BEGIN(sqrt64only) x = sqrt(x + 1.0); BENCH_END
BENCH_BEGIN(rcp64) x = __builtin_amdgcn_rcp(x) + 1.5; BENCH_END
BENCH_BEGIN(rsq64) x = __builtin_amdgcn_rsq(x) + 1.5; BENCH_END
BENCH_BEGIN(fma32) { float f = (float)x; f = fmaf(f, 1.0001f, 1e-3f); x = f; } BENCH_END
BENCH_BEGIN(hyp)
  { double a = fabs(x), b = fabs(y); double h;
    if (a > b) { b /= a; h = a * sqrt(1 + b * b); } else { a /= b; h = b * sqrt(1 + a * a); }
    x = h * 0.75 + 0.1; }
BENCH_END
// the whole c/s/t chain of one Jacobi rotation
BENCH_BEGIN(chain)
  { double p = x, yy = (y - x) * 0.5;
    double a = fabs(p), b = fabs(yy), h;
    if (a > b) { b /= a; h = a * sqrt(1 + b * b); } else if (b > 0) { a /= b; h = b * sqrt(1 + a * a); } else h = 0;
    double t = fabs(yy) + h;
    a = fabs(p); b = fabs(t); double s;
    if (a > b) { b /= a; s = a * sqrt(1 + b * b); } else if (b > 0) { a /= b; s = b * sqrt(1 + a * a); } else s = 0;
    double c = t / s; s = p / s; t = (p / t) * p;
    x = c * 0.5 + s * 0.25 + 0.3; y = t * 0.1 + 1.0; }
BENCH_END
BENCH_BEGIN(lds_rt)   // write then dependent read (value dependency) of an f64
  { lds[128 + lane] = x; x = lds[128 + ((lane + 1) & 63)] + 1e-9; }
BENCH_END
BENCH_BEGIN(lds_read_dep)  // pointer chase: address depends on the previous read
  { ix = ((int*)lds)[ix & 127] & 63; }
BENCH_END
BENCH_BEGIN(lds_rt_sync)
  { lds[128 + lane] = x; __syncthreads(); x = lds[128 + ((lane + 1) & 63)] + 1e-9; __syncthreads(); }
BENCH_END
BENCH_BEGIN(sync_only) __syncthreads(); asm volatile("" : "+v"(x)); BENCH_END
BENCH_BEGIN(dpp_mov)
  { ix = __builtin_amdgcn_update_dpp(ix, ix, 0xB1, 0xF, 0xF, false) + 1; }
BENCH_END
BENCH_BEGIN(dpp_f64_max)   // one argmax-free max step: two dpp movs + v_max_f64
  { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xF, 0xF, false);
    x = fmax(x, __hiloint2double(hi, lo)) + 1e-9; }
BENCH_END
BENCH_BEGIN(bpermute) { ix = __builtin_amdgcn_ds_bpermute(((ix + 1) & 63) << 2, ix) + 1; } BENCH_END
BENCH_BEGIN(readlane_dyn)  // readfirstlane -> readlane with a dynamic (uniform) lane -> back to vector
  { int s = __builtin_amdgcn_readfirstlane(ix) & 63; ix = __builtin_amdgcn_readlane(ix, s) + lane; }
BENCH_END
BENCH_BEGIN(readlane_f64)
  { int s = __builtin_amdgcn_readfirstlane(ix) & 63;
    int lo = __builtin_amdgcn_readlane(__double2loint(x), s), hi = __builtin_amdgcn_readlane(__double2hiint(x), s);
    x = __hiloint2double(hi, lo) * 1.0000001 + lane * 1e-12; ix = ix + 1; }
BENCH_END
BENCH_BEGIN(ballot_ff1)
  { unsigned long long m = __ballot(ix & 1); ix = ix + (int)__builtin_ctzll(m | 0x8000000000000000ull) + 1; }
BENCH_END
BENCH_BEGIN(cmp_sel64) { x = (x > y) ? x * 0.999 : y; } BENCH_END
BENCH_BEGIN(writelane)
  { int s = __builtin_amdgcn_readfirstlane(ix) & 63; int v = s + 3; asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(ix) : "s"(v), "s"(s) : "m0"); ix = ix + 1; }
BENCH_END
BENCH_BEGIN(imod)  { ix = (int)((unsigned)(ix * 4164903690u + 12345u) % (unsigned)(iseed + 137)) + lane; } BENCH_END
BENCH_BEGIN(global_rt)  // dependent global load (L2/L1 hit)
  { ix = ((volatile int*)r)[8 + (ix & 15)] + lane; }
BENCH_END

#define RUN(name, ops)                                                                            \
  do {                                                                                          \
    hipMemset(d, 0, 4096);                                                                      \
    hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, d, 1.7, 5);                            \
    hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, d, 1.7, 5);                            \
    hipDeviceSynchronize();                                                                     \
    Res h; hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);                                  \
    double per = (double)h.cyc / ITERS, wper = (double)h.wall / ITERS * 10.0;                    \
    printf("%-14s %9.1f cyc/iter  %9.1f ns/iter  (%d dependent op(s))\n", #name, per - base, wper - wbase, ops); \
    if (!strcmp(#name, "empty")) { base = per; wbase = wper; printf("  (loop overhead %.1f cyc, %.1f ns subtracted below)\n", per, wper); } \
  } while (0)

#include <cstring>
int main() {
  Res* d; hipMalloc(&d, 4096);
  double base = 0, wbase = 0;
  RUN(empty, 0);
  RUN(fma64, 1); RUN(mul64, 1); RUN(add64, 1); RUN(div64, 2); RUN(div64only, 1); RUN(sqrt64, 2); RUN(sqrt64only, 2);
  RUN(rcp64, 2); RUN(rsq64, 2); RUN(fma32, 3); RUN(hyp, 1); RUN(chain, 1);
  RUN(lds_rt, 1); RUN(lds_read_dep, 1); RUN(lds_rt_sync, 1); RUN(sync_only, 1);
  RUN(dpp_mov, 2); RUN(dpp_f64_max, 4); RUN(bpermute, 2); RUN(readlane_dyn, 3); RUN(readlane_f64, 4);
  RUN(ballot_ff1, 3); RUN(cmp_sel64, 2); RUN(writelane, 3); RUN(imod, 1); RUN(global_rt, 1);
  return 0;
}
