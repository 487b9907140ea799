// FETCH_SIZE calibration on k_fast_main's own access pattern (/opt/skills/guides/MI355X_MICROARCH.md,
// HBM section: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Every workgroup stages a tile footprint of R rows x (C16 x 16) bytes with one 16-byte load per lane, exactly like
// fast_stage() (evh_detect.hip): item i = (row i / C16, column i % C16).  The image set is 2048 planes of 1280 x 720
// bytes (1.9 GB, far beyond L2 + Infinity Cache), each plane is walked once per launch.  Patterns (one kernel name
// each, so that rocprofv3 --pmc FETCH_SIZE reports them separately):
//   0  FAST-like      36 rows x 144 B, first byte 16 B into a 128-B line, tile pitch 128 x 28 (halo overlap, as k_fast_main)
//   1  no overlap     36 rows x 144 B, first byte 16 B into a line, tile pitch 256 x 36: touches 2 lines / 3 sectors per row
//   2  line aligned   36 rows x 128 B at a line start, tile pitch 128 x 36: exactly one 128-B line per row, no overlap
//   3  aligned 144    36 rows x 144 B at a line start, tile pitch 256 x 36
//   4  wide stream    whole 1280-B rows, 16 B per lane, fully coalesced (the guide's calibrated case: counter = 1/2)
// For every pattern the program prints the bytes REQUESTED and the unique bytes TOUCHED at 64-B and 128-B granularity;
// FETCH_SIZE (x 1024) / those = the correction factor for that pattern.
// Build: hipcc --offload-arch=gfx950 -O3 fetch_cal.hip -o fetch_cal
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o cal --output-format csv -- ./fetch_cal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

constexpr int W = 1280, H = 720, F = 2048;

struct Pat { int rows, c16, x_first, x_pitch, y_pitch, tiles_x, tiles_y; };

template <int ID>
__global__ __launch_bounds__(256) void k_cal(const uint8_t* __restrict__ src, unsigned* __restrict__ sink, Pat P) {
  const int f = blockIdx.y;
  const int ty = blockIdx.x / P.tiles_x, tx = blockIdx.x - ty * P.tiles_x;
  const uint8_t* s = src + (size_t)f * W * H + (size_t)(ty * P.y_pitch) * W + P.x_first + tx * P.x_pitch;
  const int n = P.rows * P.c16;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / P.c16, c = i - r * P.c16;
    if (ty * P.y_pitch + r >= H) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(s + (size_t)r * W + c * 16);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345679u) atomicAdd(sink, 1u);
}

template <int ID>
int run(const uint8_t* d, unsigned* sink, const Pat& P, const char* name) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_cal<ID>, dim3(P.tiles_x * P.tiles_y, F), dim3(256), 0, 0, d, sink, P);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  // ground truth on the host: requested bytes and unique sectors / lines of one plane
  std::vector<uint8_t> s64((size_t)W * H / 64 + 2, 0), s128((size_t)W * H / 128 + 2, 0);
  double req = 0;
  for (int ty = 0; ty < P.tiles_y; ty++)
    for (int tx = 0; tx < P.tiles_x; tx++)
      for (int r = 0; r < P.rows; r++) {
        const int y = ty * P.y_pitch + r;
        if (y >= H) continue;
        for (int c = 0; c < P.c16; c++) {
          const size_t a = (size_t)y * W + P.x_first + tx * P.x_pitch + c * 16;
          req += 16; s64[a / 64] = 1; s128[a / 128] = 1;
        }
      }
  double u64 = 0, u128 = 0;
  for (auto v : s64) u64 += v * 64.0;
  for (auto v : s128) u128 += v * 128.0;
  printf("pattern %d %-13s k_cal<%d>: per launch requested %.0f B, unique 64-B sectors %.0f B, unique 128-B lines %.0f B, %.3f ms\n",
         ID, name, ID, req * F, u64 * F, u128 * F, best);
  return 0;
}

int main() {
  uint8_t* d; unsigned* sink;
  CHECK(hipMalloc(&d, (size_t)W * H * F + 4096)); CHECK(hipMemset(d, 1, (size_t)W * H * F + 4096));
  CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(sink, 0, 4));
  // FAST tile grid of level 0 at 1280x720: x0 = 24 + 128 tx, staged from x0 - 8 = 16 + 128 tx; y0 = 31 + 28 ty, staged from y0 - 4
  if (run<0>(d, sink, Pat{36, 9, 16, 128, 28, 10, 24}, "fast-like")) return 1;
  if (run<1>(d, sink, Pat{36, 9, 16, 256, 36, 5, 20}, "no-overlap")) return 1;
  if (run<2>(d, sink, Pat{36, 8, 0, 128, 36, 10, 20}, "line-aligned")) return 1;
  if (run<3>(d, sink, Pat{36, 9, 0, 256, 36, 5, 20}, "aligned-144")) return 1;
  if (run<4>(d, sink, Pat{36, 80, 0, 1280, 36, 1, 20}, "wide-stream")) return 1;
  return 0;
}
