// VALU issue-rate microbenchmark for gfx950: wave-instructions per clock per CU for the integer ops the FAST
// kernel is built from.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OPS_PER_ITER 64
#define ITERS 2048
#define DEF(name, ASM)                                                                          \
  __global__ __launch_bounds__(256) void k_##name(int* out, int seed) {                          \
    int a[8];                                                                                   \
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * (i + 1);                             \
    int b = seed * 3 + 1, c = seed * 7 + 5;                                                     \
    for (int it = 0; it < ITERS; it++) {                                                        \
      _Pragma("unroll") for (int u = 0; u < OPS_PER_ITER / 8; u++) {                             \
        _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
      }                                                                                         \
    }                                                                                           \
    int s = 0;                                                                                  \
    for (int i = 0; i < 8; i++) s ^= a[i];                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                              \
  }
DEF(min3_i32, "v_min3_i32 %0, %0, %1, %2")
DEF(max3_i32, "v_max3_i32 %0, %0, %1, %2")
DEF(min_i32, "v_min_i32 %0, %0, %1")
DEF(add_u32, "v_add_u32 %0, %0, %1")
DEF(sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF(pk_min_i16, "v_pk_min_i16 %0, %0, %1")
DEF(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
DEF(fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF(and_b32, "v_and_b32 %0, %0, %1")
DEF(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF(perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEF(alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEF(lshl_or, "v_lshl_or_b32 %0, %0, %1, %2")
DEF(sad_u8, "v_sad_u8 %0, %0, %1, %2")
DEF(med3_i32, "v_med3_i32 %0, %0, %1, %2")
DEF(min_u16, "v_min_u16 %0, %0, %1")
DEF(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF(dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF(bfe_u32, "v_bfe_u32 %0, %0, %1, %2")
DEF(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF(max_i16, "v_max_i16 %0, %0, %1")
DEF(sub_u32, "v_sub_u32 %0, %0, %1")
DEF(or_b32, "v_or_b32 %0, %0, %1")
DEF(xor_b32, "v_xor_b32 %0, %0, %1")
DEF(lshlrev, "v_lshlrev_b32 %0, 3, %0")
DEF(lshrrev, "v_lshrrev_b32 %0, 3, %0")
DEF(bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF(or3_b32, "v_or3_b32 %0, %0, %1, %2")
DEF(add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEF(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
DEF(mul_u24, "v_mul_u32_u24 %0, %0, %1")
DEF(mul_f32, "v_mul_f32 %0, %0, %1")
DEF(min_f32, "v_min_f32 %0, %0, %1")
DEF(cvt_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
DEF(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
DEF(alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
DEF(dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
DEF(max_u16, "v_max_u16 %0, %0, %1")
DEF(sub_u16, "v_sub_u16 %0, %0, %1")
DEF(mov_b32, "v_mov_b32 %0, %1")
DEF(mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %1, %0")

typedef void (*kern_t)(int*, int);
struct K { const char* n; kern_t f; };
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount; double mhz = p.clockRate / 1000.0;
  printf("device %s CUs %d clock %.0f MHz\n", p.name, cus, mhz);
  int blocks = cus * 8; int* out; hipMalloc(&out, blocks * 256 * sizeof(int));
  K ks[] = {{"min3_i32", k_min3_i32}, {"max3_i32", k_max3_i32}, {"min_i32", k_min_i32}, {"add_u32", k_add_u32},
            {"sub_sdwa", k_sub_sdwa}, {"pk_min_i16", k_pk_min_i16}, {"pk_max_i16", k_pk_max_i16}, {"pk_sub_i16", k_pk_sub_i16},
            {"fma_f32", k_fma_f32}, {"and_b32", k_and_b32}, {"and_or", k_and_or}, {"perm_b32", k_perm_b32},
            {"alignbit", k_alignbit}, {"lshl_or", k_lshl_or}, {"sad_u8", k_sad_u8}, {"med3_i32", k_med3_i32},
            {"min_u16", k_min_u16}, {"mad_u32_u24", k_mad_u32_u24}, {"dot4_u8", k_dot4_u8}, {"bfe_u32", k_bfe_u32},
            {"cndmask", k_cndmask}, {"max_i16", k_max_i16}, {"sub_u32", k_sub_u32}, {"or_b32", k_or_b32},
            {"xor_b32", k_xor_b32}, {"lshlrev", k_lshlrev}, {"lshrrev", k_lshrrev}, {"bitop3", k_bitop3},
            {"or3_b32", k_or3_b32}, {"add3_u32", k_add3_u32}, {"lshl_add", k_lshl_add}, {"mul_u24", k_mul_u24},
            {"mul_f32", k_mul_f32}, {"min_f32", k_min_f32}, {"cvt_ubyte0", k_cvt_ubyte0}, {"cvt_u32_f32", k_cvt_u32_f32},
            {"alignbyte", k_alignbyte}, {"dot2_u16", k_dot2_u16}, {"max_u16", k_max_u16}, {"sub_u16", k_sub_u16},
            {"mov_b32", k_mov_b32}, {"mbcnt_lo", k_mbcnt_lo}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double waveinstr = (double)blocks * 4 * ITERS * OPS_PER_ITER;
    double per_clk_cu = waveinstr / (ms * 1e-3) / (mhz * 1e6) / cus;
    printf("%-12s %.3f ms  %.3f wave-instr/clk/CU  => %.1f lane-ops/clk/CU (at %0.f MHz nominal)\n", k.n, ms, per_clk_cu, per_clk_cu * 64, mhz);
  }
  return 0;
}
