// evh_ransac.h -- argument block of the homography kernels (one workgroup of 1 or 4 wavefronts per frame pair / stream).
#pragma once
#include <stdint.h>

struct EvhRansacArgs {
  // inputs
  float* pts;            // matched rows (ax, ay, bx, by), [pair][row_stride][4]; also scratch in the final phase
  float* pts2;           // static rows, [pair][row_stride][4]
  int64_t row_stride;    // rows per pair in every per-pair buffer
  const int* npts;       // rows in pts per pair
  int* npts2;            // rows in pts2 per pair
  int* status;           // per pair status carried between phases
  int n_fixed;           // single-problem entry: number of rows
  double thr; int max_iters; double conf; int force_max;
  int fast_solver;       // EVH_SOLVER_FAST: the 8x8 systems of the LM refinement by LDL^T instead of cv::solve(DECOMP_EIG)'s Jacobi sweeps
  const double* Hsup0;   // stream mode: superposition entering the batch (NULL = first pair of the stream)
  const double* Hprev0;  // stream mode: previous H entering the batch (NULL = none)
  double* state_out;     // stream mode: {H_sup, H_prev} after the last pair, f64[18] (may be NULL)
  // scratch
  uint8_t* mask;         // [pair][row_stride]
  float* crow;           // [pair][row_stride][4] compacted inlier rows
  double* lm;            // [pair][row_stride][4]
  // outputs
  double* H1;            // provisional H per pair (may be NULL)
  double* H;             // final H per pair
  int* out_status;       // final status per pair
  int* found;            // single-problem entry
  double* lane_v;        // fixed-iteration mode: global scratch for the per-lane eigenvector matrices, [workgroup][4][81][64] f64
  unsigned long long* prof;  // optional cycle accounting (debug, EVH_RANSAC_PROF); NULL normally
  int* info;             // [pair][8] (may be NULL): ransac iters, best inliers, LM iters for RANSAC #1 (+0) and #2 (+4)
};

#define EVH_LANE_V_DOUBLES (4 * 81 * 64)   // per workgroup
struct evh_ctx;
int evh_launch_find_homography(evh_ctx* c, const EvhRansacArgs& A);
int evh_launch_static_filter(evh_ctx* c, const double* d_H, const float* d_rows, int n, int* d_rbin, float* d_out,
                             int* d_count);
int evh_launch_ransac_static(evh_ctx* c, const EvhRansacArgs& A, int npairs);
int evh_launch_ransac_final(evh_ctx* c, const EvhRansacArgs& A, int npairs, int nstreams, int pitch);
