// evh_internal.h -- shared declarations of libevhip.so (host context + kernel launch prototypes).
// Product code: nothing here (or in any file of this directory) includes or links oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/evhip.h"

#define EVH_NLEVELS 8
#define EVH_EDGE 31          // ORB edgeThreshold
#define EVH_FAST_THR 20      // ORB fastThreshold
#define EVH_FAST_OX 24       // origin of the FAST tile grid (multiple of 8: the staging loads are 8-byte aligned)
#define EVH_FAST_OY 31
#define EVH_K1CAP 4096       // stage-1 (FAST-score) survivors per level held in LDS
#define EVH_K2CAP 1280       // stage-2 (Harris) survivors per level held in LDS

struct EvhLevel {
  int w, h, stride;     // stride in bytes (64-byte aligned)
  int quota;            // nfeaturesPerLevel
  float scale;          // layerScale
  int64_t off;          // byte offset inside one frame's pyramid
  int64_t cand_off;     // entry offset inside one frame's candidate buffer
  int cand_cap;
  int tile_start;       // first FAST tile index of this level
  int tiles_x, tiles_y;
  int tab_off;          // offset (ints) of the resize tables of this level inside d_tabs
  int kp_base, kp_cap;  // this level's segment inside a frame's keypoint staging arrays
};

struct EvhGeom {
  int w, h, nfeatures;
  EvhLevel lv[EVH_NLEVELS];
  int64_t pyr_frame_bytes;
  int64_t cand_frame_entries;
  int total_tiles;
};

// N4: SIFT scale-space geometry of one frame size (octave 0 = the frame doubled, 6 Gaussian layers per octave)
#define EVH_SIFT_MAXOCT 13
struct EvhSiftGeom {
  int w, h, noct;
  int ow[EVH_SIFT_MAXOCT], oh[EVH_SIFT_MAXOCT], os[EVH_SIFT_MAXOCT];   // octave width / height / row stride (floats)
  int64_t ooff[EVH_SIFT_MAXOCT];                                       // float offset of an octave's layer 0 inside a frame
  int64_t frame_floats, tmp_floats;
};

// per-pair working buffers of the matching / RANSAC stages (row stride `cap` rows per pair)
struct EvhPairBufs {
  int cap = 0;
  int32_t* knn_idx = nullptr; uint32_t* knn_d2 = nullptr;
  float* pts = nullptr; float* pts2 = nullptr; float* crow = nullptr;
  int* npts = nullptr; int* npts2 = nullptr; int* pstatus = nullptr;
  double* H1 = nullptr; uint8_t* mask = nullptr; double* lm = nullptr; int* info = nullptr;
};

struct evh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // asynchronous solve: the RANSAC kernels of a batch run on a second stream so that they overlap the next
  // batch's detect kernels (they occupy one wave per SIMD and are latency-bound)
  hipStream_t solve_stream = nullptr;
  hipEvent_t ev_match_done = nullptr, ev_solve_done = nullptr;
  bool async_solve = false, solve_pending = false;
  int max_w = 0, max_h = 0, max_features = 0, max_frames = 0;
  int kcap = 0;  // keypoint rows per frame slot
  EvhGeom g{};
  bool geom_valid = false;
  bool level1_fused = false;   // set by evh_launch_gray_level0: level 1 was produced together with level 0
  int nframes_resident = 0;
  // device buffers
  uint8_t* d_pyr = nullptr;       // [max_frames][pyr_frame_bytes]
  uint32_t* d_cand = nullptr;     // [max_frames][cand_frame_entries]
  int* d_cand_count = nullptr;    // [max_frames][8]
  int* d_tabs = nullptr;          // resize tables for levels 1..7
  int* h_tabs = nullptr;          // pinned staging of the same (configure() uploads stream-ordered, no host sync)
  hipEvent_t ev_tabs = nullptr;   // the last upload out of h_tabs
  float* d_kp_xy = nullptr;       // [max_frames][kcap][2]
  uint32_t* d_kp_meta = nullptr;  // [max_frames][kcap]  level<<24 | y<<12 | x
  float* d_kp_resp = nullptr;     // [max_frames][kcap]
  float* d_kp_angle = nullptr;    // [max_frames][kcap]
  uint8_t* d_desc = nullptr;      // [max_frames][kcap][32]
  int* d_kp_count = nullptr;      // [max_frames]
  int* d_frame_flags = nullptr;   // [max_frames] bit0: capacity overflow
  uint32_t* d_tmp_meta = nullptr; // [max_frames][8][kcap] per-level segments before packing
  float* d_tmp_resp = nullptr;    // [max_frames][8][kcap]
  int* d_lvl_count = nullptr;     // [max_frames][8]
  int* d_fast_thr = nullptr;      // [max_frames][8] lifted FAST threshold
  unsigned* d_fast_hist = nullptr;// [max_frames][8][256] sampled score histogram
  int* d_fast_hint = nullptr;     // [2][8] per-level threshold hint (ping-pong between detect calls) + [8][256] votes + [8] zeros
  int fast_hint_idx = 0;
  double* d_lane_v = nullptr;     // fixed-iteration RANSAC: per-lane eigenvector matrices [max_frames][4][81][64] (lazy)
  int* d_merge_ws = nullptr; size_t merge_ws_bytes = 0;     // k_merge_dup -> k_merge
  int* d_filter_ws = nullptr; size_t filter_ws_bytes = 0;   // k_filter<true>: work arrays of key-point budgets beyond the LDS form
  char* d_scan_ws = nullptr;      // fixed-iteration stream scan: state, sample table and hypothesis results (evh_ransac.hip, lazy)
  size_t scan_ws_bytes = 0;
  int* d_area_tab = nullptr;      // INTER_AREA tables of the last ingest geometry (evh_launch_ingest_level0)
  int64_t area_key = -1; int area_nx = 0, area_ny = 0;
  int* d_fast_redo = nullptr;     // [1 + max_frames*8] redo work list (count first)
  // key-point order of the reference (EVH_ORDER_OPENCV): work arrays of k_select_cv
  int order_mode = 1;             // EVH_ORDER_OPENCV
  int solver_mode = 0;            // EVH_SOLVER_EXACT
  unsigned long long* d_cv_seq = nullptr;   // [max_frames][cand_frame_entries] key << 32 | candidate, row-major then permuted
  uint32_t* d_cv_seq32 = nullptr; // [max_frames][cand_frame_entries] the first retainBest works on the candidates themselves
  uint32_t* d_cv_lpos = nullptr;  // [max_frames][cand_frame_entries] stopper positions of the partition passes
  uint32_t* d_cv_rpos = nullptr;
  uint32_t* d_cv_mask = nullptr;  // [max_frames][2][cv_mask_frame_words] prefix table of the tiles' row counts
  uint32_t* d_cv_tdesc = nullptr; // [max_frames][total_tiles][8] FAST tile burst descriptors
  int64_t cv_mask_frame_words = 0;
  bool fast_lift = true;
  bool fast_share = true;         // evh_set_fast_share
  bool fast_hint = true;          // evh_set_fast_hint
  int fast_share_group = 0;       // frames per group of consecutive frames for this detect call (0: unrelated frames)
  // pair buffers (max_pairs = max_frames)
  int32_t* d_knn_idx = nullptr;   // [max_pairs][kcap][2]
  uint32_t* d_knn_d2 = nullptr;   // [max_pairs][kcap][2]
  float* d_pts = nullptr;         // [max_pairs][kcap][4] matched rows
  float* d_pts2 = nullptr;        // [max_pairs][kcap][4] static rows
  int* d_npts = nullptr;          // [max_pairs]
  int* d_npts2 = nullptr;         // [max_pairs]
  int* d_pstatus = nullptr;       // [max_pairs]
  double* d_H1 = nullptr;         // [max_pairs][9]
  uint8_t* d_mask = nullptr;      // [max_pairs][kcap]
  double* d_lm = nullptr;         // [max_pairs][kcap][4] LM per-point temporaries
  float* d_crow = nullptr;        // [max_pairs][kcap][4] compacted inlier rows
  int* d_info = nullptr;          // [max_pairs][8]
  double* d_small = nullptr;      // small staging area for single-problem entries (H, counts)
  char* d_scratch = nullptr;      // growable scratch of the host-pointer entries (N1 / N3): no hipMalloc per call
  size_t scratch_bytes = 0;
  // ---- N4: SIFT (allocated by evh_sift_enable) ----
  int sift_cap = 0, sift_cand_cap = 0, sift_group = 0, sift_frames_resident = 0;
  EvhSiftGeom sg{}; bool sift_geom_valid = false;
  int64_t sift_pyr_frame_floats = 0, sift_tmp_frame_floats = 0;
  float* d_sift_pyr = nullptr;    // [group][frame_floats] Gaussian scale space
  float* d_sift_tmp = nullptr;    // [group][octave-0 layer] row-pass temporary
  uint32_t* d_sift_cand = nullptr; int* d_sift_ncand = nullptr;   // extrema: octave<<28 | layer<<26 | r<<13 | c
  float* d_sift_raw = nullptr; int* d_sift_nraw = nullptr;        // [F][cap][8] key points before the sort
  float* d_sift_srt = nullptr;    // [F][cap][8] sorted
  float* d_sift_kp = nullptr;     // [F][cap][8] final records: x, y, size, angle, response, octave bits
  float* d_sift_xy = nullptr;     // [F][cap][2]
  uint8_t* d_sift_desc = nullptr; // [F][cap][128] descriptor VALUES (0..255; the operator returns them as float32)
  int* d_sift_count = nullptr; int* d_sift_flags = nullptr;
  // ---- N4: SURF (allocated by evh_surf_enable) ----
  int surf_cap = 0, surf_group = 0, surf_frames_resident = 0, surf_tab_w = 0, surf_tab_h = 0;
  int64_t surf_sum_frame_ints = 0, surf_det_frame_floats = 0;
  char* d_surf_tabs = nullptr;    // SurfTabs: layer boxes, orientation / descriptor weights
  int* d_surf_sum = nullptr;      // [group] integral images, (h+1) x (w+1)
  float* d_surf_det = nullptr; float* d_surf_trace = nullptr;   // [group] the 20 Hessian layers
  float* d_surf_raw = nullptr; int* d_surf_nraw = nullptr; float* d_surf_srt = nullptr;
  float* d_surf_kp = nullptr;     // [F][cap][8] x, y, size, angle, response, octave bits, laplacian bits
  float* d_surf_xy = nullptr; float* d_surf_desc = nullptr;     // [F][cap][2], [F][cap][128] float
  int* d_surf_count = nullptr; int* d_surf_flags = nullptr;
  // multi-type pairs (frame_processing.py:91-104): per-type match / static rows, their concatenation, the merged rows
  EvhPairBufs mt;                 // stride mt.cap = kcap + sift_cap + surf_cap
  float* d_acc = nullptr; int* d_nacc = nullptr; int* d_accstatus = nullptr;
  size_t bytes_allocated = 0;
  std::string err;
  // per-stage timing (evh_profile_*)
  bool profiling = false;
  struct ProfSpan { int stage; hipEvent_t a, b; };
  std::vector<ProfSpan> prof_spans;       // recorded since the last read
  std::vector<hipEvent_t> prof_pool;      // recycled events
};

enum { EVH_ST_GRAY = 0, EVH_ST_PYRAMID, EVH_ST_FAST, EVH_ST_SELECT, EVH_ST_DESCRIBE, EVH_ST_KNN, EVH_ST_FILTER,
       EVH_ST_RANSAC_STATIC, EVH_ST_RANSAC_FINAL };
// RAII bracket: records an event pair around the launches issued while it is alive (no-op unless profiling)
struct EvhProfScope {
  evh_ctx* c; int idx; hipStream_t st;
  EvhProfScope(evh_ctx* ctx, int stage, hipStream_t on = nullptr);
  ~EvhProfScope();
};

int evh_fail(evh_ctx* ctx, int code, const std::string& msg);
#define EVH_HIP(ctx, call)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return evh_fail(ctx, EVH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));              \
  } while (0)

// ---- kernel launchers (each enqueues on ctx->stream) ----
int evh_launch_gray_level0(evh_ctx* c, const uint8_t* d_frames, int nframes, int channels, int64_t row_stride,
                           int64_t frame_stride);
int evh_launch_ingest_level0(evh_ctx* c, const uint8_t* d_src, int nimg, int sw, int sh, int cn, int64_t src_stride,
                             int64_t src_img_stride, int dw, int dh);
int evh_launch_pyramid(evh_ctx* c, int nframes);
int evh_launch_fast(evh_ctx* c, int nframes, int share_group);
int evh_launch_select(evh_ctx* c, int nframes);
int evh_launch_describe(evh_ctx* c, int nframes);
int evh_launch_superposition_scan(evh_ctx* c, const double* d_H, int n, double* d_out);
int evh_launch_transform_points(evh_ctx* c, const double* d_M, const int* d_idx, const double* d_pts, int n, double kx,
                                double ky, int decimals, double* d_out);
int evh_launch_fixed_plane(evh_ctx* c, const double* d_H, int n, int w, int h, double* d_field, unsigned long long* d_max);
// N4: SIFT (evh_sift.hip)
int evh_sift_allocate(evh_ctx* c, int max_sift_features);
void evh_sift_free(evh_ctx* c);
int evh_launch_sift(evh_ctx* c, int nframes, int w, int h);
// N4: SURF (evh_surf.hip)
int evh_surf_allocate(evh_ctx* c, int max_surf_features);
void evh_surf_free(evh_ctx* c);
int evh_launch_surf(evh_ctx* c, int nframes, int w, int h, float hessian_threshold);
int evh_launch_resize_area(evh_ctx* c, const uint8_t* d_src, int nimg, int sw, int sh, int cn, int64_t src_stride,
                           int64_t src_img_stride, uint8_t* d_dst, int dw, int dh, int64_t dst_stride,
                           int64_t dst_img_stride);
