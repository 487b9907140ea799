// evh_match.h -- argument blocks of the matching kernels.  A "slot" is a frame's region inside a batched
// feature buffer; pair p uses query slot q_slot0 + p*q_slot_step and train slot t_slot0 + p*t_slot_step
// (independent pairs: 1,2 / 0,2; stream: 1,1 / 0,1; explicit buffers: all zero).
#pragma once
#include <stdint.h>

struct EvhKnnArgs {
  const uint8_t* q; const uint8_t* t;
  int64_t slot_bytes;             // bytes per descriptor slot
  const int* nq_arr; const int* nt_arr;  // per-slot counts, or NULL to use the fixed counts
  int nq_fixed, nt_fixed;
  int q_slot0, q_slot_step, t_slot0, t_slot_step;
  int32_t* idx; uint32_t* d2;
  int64_t out_stride;             // rows per pair in idx / d2
  int hamming;
  int desc_bytes;                 // 32 (ORB) or 128 (SIFT descriptor values as bytes); 0 = 32
};

struct EvhFilterArgs {
  const int32_t* idx; const uint32_t* d2;
  int64_t knn_stride;             // rows per pair in idx / d2
  const float* xy_q; const float* xy_t;
  int64_t xy_slot_floats;         // floats per coordinate slot
  const int* nq_arr; const int* nt_arr; const int* flags_arr;
  int nq_fixed, nt_fixed;
  int q_slot0, q_slot_step, t_slot0, t_slot_step;
  double ratio; int min_matches;
  float* pts; int64_t pts_stride; // rows per pair
  int* npts; int* status;
  int kcap;                       // LDS sizing: >= max(nq, nt)
  int d2_is_dist;                 // d2 holds the float32 bits of the (square-rooted) distance: float descriptors
  int* work;                      // k_filter<true>: global work arrays, 5 * kcap ints per pair (set by evh_launch_filter)
};
#define EVH_FILTER_LDS_MAX (150 * 1024)   // 5 * kcap ints of LDS: up to 7 680 key points per frame; beyond that a global scratch

// BruteForce 2-NN on float32 descriptors (SIFT / SURF rows of `dim` floats, dim = 64 or 128)
struct EvhKnnF32Args {
  const float* q; const float* t; int nq, nt, dim; int32_t* idx; float* dist;
  // batched form (pairs of frame slots, like EvhKnnArgs): counts per slot, slot stride in floats, rows per pair in idx / dist
  const int* n_arr; int64_t slot_floats; int q_slot0, q_slot_step, t_slot0, t_slot_step; int64_t out_stride;
};

// multi-type pairs (frame_processing.py:91-104): append one feature type's static rows, then remove_double_matching
struct EvhAccArgs {
  const float* rows; const int* nrows; const int* status; int64_t row_stride;   // per-type static rows [pair][row_stride][4]
  float* acc; int* nacc; int* accstatus; int64_t acc_stride;                    // concatenation [pair][acc_stride][4]
  int first;                                                                    // first type of the list: resets the concatenation
};
struct EvhMergeArgs {
  const float* acc; const int* nacc; const int* accstatus; int64_t acc_stride;
  float* out; int* nout; int* status; int64_t out_stride;
  int* work;                      // [pair][acc_stride][2]: first-of-key flag and the key's last row (set by evh_launch_merge)
};

struct evh_ctx;
int evh_launch_knn2_f32(evh_ctx* c, const EvhKnnF32Args& A, int npairs = 1);
int evh_launch_accumulate(evh_ctx* c, const EvhAccArgs& A, int npairs);
int evh_launch_merge(evh_ctx* c, const EvhMergeArgs& A, int npairs);
int evh_launch_knn2(evh_ctx* c, const EvhKnnArgs& A, int npairs);
int evh_launch_filter(evh_ctx* c, const EvhFilterArgs& A, int npairs);
