// Achievable HBM bandwidth on one MI355X, for the roofline's denominator: read-only, write-only, copy and a 3:1
// read:write mix (the shape of k_gray_pyr1: 3 bytes in, 1.44 out) with 16-byte accesses, grid-stride, 2 GiB buffers.
// Build: hipcc --offload-arch=gfx950 -O3 bw.hip -o bw ; run: ./bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ a, size_t n, uint4* __restrict__ sink) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ b, size_t n) {
  const uint4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 3 reads : 1 write
__global__ __launch_bounds__(256) void k_mix31(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 x = a[3 * i], y = a[3 * i + 1], z = a[3 * i + 2];
    uint4 o = {x.x ^ y.x ^ z.x, x.y ^ y.y ^ z.y, x.z ^ y.z ^ z.z, x.w ^ y.w ^ z.w};
    b[i] = o;
  }
}
template <class F> double timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps * 1e-3;
}
int main() {
  const size_t bytes = (size_t)3 << 30, n = bytes / 16;
  uint4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  for (int grid : {2048, 8192, 65536}) {
    double t;
    t = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, b); }, 5);
    printf("grid %6d  read   %6.2f TB/s\n", grid, bytes / t * 1e-12);
    t = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); }, 5);
    printf("grid %6d  write  %6.2f TB/s\n", grid, bytes / t * 1e-12);
    t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 5);
    printf("grid %6d  copy   %6.2f TB/s (read + write bytes)\n", grid, 2.0 * bytes / t * 1e-12);
    t = timeit([&] { hipLaunchKernelGGL(k_mix31, dim3(grid), dim3(256), 0, 0, a, b, n / 3); }, 5);
    printf("grid %6d  mix3:1 %6.2f TB/s (read + write bytes)\n", grid, (bytes + bytes / 3.0) / t * 1e-12);
  }
  double t = timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
  printf("hipMemcpy D2D      %6.2f TB/s (read + write bytes)\n", 2.0 * bytes / t * 1e-12);
  return 0;
}
