// How much of the streaming bandwidth (bw.hip) survives a TILED walk of 2048 frames of 1280x720 BGR: every workgroup
// reads R rows x S bytes of one frame (row pitch 3840) and writes a third of that in the same tiling to a gray plane.
// Build: hipcc --offload-arch=gfx950 -O3 bw_tile.hip -o bw_tile ; run: ./bw_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

// S = tile row bytes in BGR (multiple of 48 so that the gray row is a multiple of 16), R = tile rows
__global__ __launch_bounds__(256) void k_tile(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int S, int R,
                                              int tiles_x, int tiles_y, int xcd_swizzle) {
  int b = blockIdx.x;
  if (xcd_swizzle) {   // consecutive tiles on the SAME XCD: workgroup b runs on XCD b % 8
    const int per = (gridDim.x + 7) / 8;
    b = (b % 8) * per + b / 8;
    if (b >= (int)gridDim.x) return;
  }
  const int f = blockIdx.y;
  const int ty = b / tiles_x, tx = b - ty * tiles_x;
  const uint8_t* s = src + (size_t)f * 3840 * 720 + (size_t)(ty * R) * 3840 + (size_t)tx * S;
  uint8_t* d = dst + (size_t)f * 1280 * 720 + (size_t)(ty * R) * 1280 + (size_t)tx * (S / 3);
  const int c16 = S / 16, n = c16 * R;          // 16-byte cells of the tile
  const int g16 = S / 48;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / c16, c = i - r * c16;
    if (ty * R + r >= 720) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(s + (size_t)r * 3840 + c * 16);
    // every third cell writes one (so that bytes out = bytes in / 3), value depends on the load
    if (c < g16) *reinterpret_cast<uint4*>(d + (size_t)r * 1280 + c * 16) = v;
    else if ((v.x ^ v.y) == 0x12345679u) d[0] = 1;
  }
}
int main() {
  const int F = 2048;
  const size_t sb = (size_t)F * 3840 * 720, db = (size_t)F * 1280 * 720;
  uint8_t *s, *d; CHECK(hipMalloc(&s, sb)); CHECK(hipMalloc(&d, db)); CHECK(hipMemset(s, 1, sb)); CHECK(hipMemset(d, 0, db));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int shapes[][2] = {{480, 40}, {960, 20}, {1920, 10}, {3840, 5}, {3840, 2}, {240, 80}, {3840, 40}};
  for (auto& sh : shapes) for (int sw = 0; sw < 2; sw++) {
    const int S = sh[0], R = sh[1], tiles_x = 3840 / S, tiles_y = (720 + R - 1) / R;
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_tile, dim3(tiles_x * tiles_y, F), dim3(256), 0, 0, s, d, S, R, tiles_x, tiles_y, sw);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    printf("tile %4d B x %2d rows  xcd_swizzle %d  %6.3f ms  %5.2f TB/s (read + write)\n", S, R, sw, best,
           (sb + db) / (best * 1e-3) * 1e-12);
  }
  return 0;
}
