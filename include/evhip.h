/*
 * include/evhip.h -- C ABI of libevhip.so: the MI355X (gfx950) implementation of EvenVizion's
 * frame-to-frame homography hot path.
 *
 * The reference (gridl/EvenVizion) has no FFI of its own: its operator boundary is four third-party calls
 * made from Python.  Each entry point below names the reference call site it replaces:
 *
 *   evh_resize_area_u8            imutils.resize(frame, width=)          video_processing.py:62,73
 *   evh_orb_detect_batch          cv2.ORB_create().detectAndCompute      frame_processing.py:59-61
 *   evh_match_knn2_l2u8           DescriptorMatcher("BruteForce").knnMatch(q,t,2)   matching.py:102-108
 *   evh_ratio_unique_filter       lowes_ratio_test + filter_corresponding_points +
 *                                 remove_double_matching                 matching.py:112-119,166-239; utils.py:41-68
 *   evh_superposition_scan        utils.superposition_dict / matrix_superposition   utils.py:118-145,184-211
 *   evh_transform_points          from_original_to_fix / from_fix_to_original       fixed_coordinate_system.py:19-122
 *   evh_find_homography_ransac    cv2.findHomography(a,b,cv2.RANSAC,3.0) matching.py:156-157; utils.py:356-358
 *   evh_static_filter             find_point_displacement + get_largest_group_points   utils.py:258-325
 *   evh_sift_detect_batch         cv2.xfeatures2d.SIFT_create().detectAndCompute   frame_processing.py:62-64
 *   evh_surf_detect_batch         cv2.xfeatures2d.SURF_create(extended=1, hessianThreshold=400).detectAndCompute   frame_processing.py:65-67
 *   evh_match_knn2_l2f32          knnMatch on float32[N,128] descriptors            matching.py:102-108
 *   evh_*_homography_batch_types  concatenate_all_features_types over a type list   frame_processing.py:91-104
 *   evh_pair_homography_batch     the per-pair body of get_homography_dict video_processing.py:67-105
 *                                 (FrameProcessing.concatenate_all_features_types frame_processing.py:73-108
 *                                  + compute_homography utils.py:328-363 + matrix_superposition utils.py:118-145)
 *
 * Conventions: extern "C", plain pointers and sizes, int return (0 = EVH_SUCCESS, <0 = error; the message is
 * available from evh_last_error_string), never throws.  Pointers named d_* are DEVICE pointers (hipMalloc /
 * torch.Tensor.data_ptr()); pointers named h_* are host pointers.  One context is used from one host thread
 * at a time; multi-GPU = one context (one process) per device.  All work is enqueued on the context's HIP
 * stream; entry points that fill h_* outputs synchronise that stream before returning, the others do not.
 */
#ifndef EVHIP_H
#define EVHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct evh_ctx evh_ctx;

enum {
  EVH_SUCCESS = 0,
  EVH_ERR_INVALID = -1,   /* bad argument                          */
  EVH_ERR_HIP = -2,       /* a HIP runtime call failed             */
  EVH_ERR_CAPACITY = -3,  /* exceeds the sizes given to evh_create */
  EVH_ERR_UNSUPPORTED = -4
};

/* per-pair status (out_status); mirrors the reference's failure list:
 *  1 matching.py:104-107 descriptors None   2 matching.py:113 fewer than 4 matches
 *  3 matching.py:158 provisional H None     4 utils.py:359 inlier ratio < 0.7     5 utils.py:361 H None
 *  6 an internal fixed-capacity list overflowed for this frame (never silently truncated)             */
enum {
  EVH_PAIR_OK = 0,
  EVH_PAIR_NO_DESCRIPTORS = 1,
  EVH_PAIR_FEW_MATCHES = 2,
  EVH_PAIR_NO_PROVISIONAL_H = 3,
  EVH_PAIR_LOW_INLIER_RATIO = 4,
  EVH_PAIR_NO_FINAL_H = 5,
  EVH_PAIR_CAPACITY = 6
};

enum { EVH_MODE_INDEPENDENT_PAIRS = 0, EVH_MODE_STREAM = 1 };

/* ---- context ------------------------------------------------------------------------------------------------ */
/* stream: a hipStream_t to enqueue on, or NULL to let the context create its own non-blocking stream.          */
/* 64 <= max_w, max_h < 4096; 2 <= max_frames <= 65528 (frames are a grid dimension); else EVH_ERR_INVALID.    */
int evh_create(int device, int max_w, int max_h, int max_features, int max_frames, void* stream, evh_ctx** out);
void evh_destroy(evh_ctx* ctx);
const char* evh_last_error_string(const evh_ctx* ctx); /* ctx may be NULL: last error of evh_create */
void* evh_stream(const evh_ctx* ctx);                  /* the hipStream_t all kernels are launched on */
int evh_synchronize(evh_ctx* ctx);                     /* waits for the main AND the solve stream */
int evh_version(void);
/* Asynchronous solve (default off): the RANSAC kernels of evh_pair_homography_batch / evh_stream_homography_batch
 * are enqueued on a second, context-owned stream behind an event, so that they overlap the NEXT batch's detect
 * kernels (they are latency-bound and occupy one wave per SIMD).  d_H / d_status are then complete only after
 * evh_synchronize(), or on a stream that has called evh_solve_wait (stream = NULL: the context's main stream). */
int evh_set_async_solve(evh_ctx* ctx, int on);
int evh_solve_wait(evh_ctx* ctx, void* stream);

/* ---- per-stage device timing (measurement aid, used by bench.py) ------------------------------------------- */
/* When enabled, every kernel group launched by the entry points below is bracketed by a pair of hipEvents on
 * the context's stream.  evh_profile_read synchronises the stream, returns for each stage the number of
 * bracketed launches-groups and their summed device time in milliseconds since the last read, and resets.
 * Stage order: gray, pyramid, fast, select, describe, knn2, filter, ransac_static, ransac_final.           */
#define EVH_NSTAGES 9
int evh_profile_enable(evh_ctx* ctx, int on);
int evh_profile_read(evh_ctx* ctx, float* h_total_ms /*[EVH_NSTAGES]*/, int* h_counts /*[EVH_NSTAGES]*/);
const char* evh_profile_stage_name(int stage);

/* ---- K0: imutils.resize -> cv2.resize(INTER_AREA): area sums when shrinking, the operator's bilinear emulation
 * when enlarging (resize_width > frame width), identity = copy ------------------------------------------------- */
/* nimg images of sh x sw x cn uint8 (row stride src_stride bytes, image stride src_img_stride bytes).          */
int evh_resize_area_u8(evh_ctx* ctx, const uint8_t* d_src, int nimg, int sw, int sh, int cn, int64_t src_stride,
                       int64_t src_img_stride, uint8_t* d_dst, int dw, int dh, int64_t dst_stride,
                       int64_t dst_img_stride);

/* BGR convenience form of the same call (one image): cn = 3, tight rows.                                        */
int evh_resize_area_u8c3(evh_ctx* ctx, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh);

/* ---- N3: fixed-plane coordinate field of the heat-map (processing_visualization.py:407-408,419) ------- */
/* For each of n superposed matrices h_Hsup f64[n,9]: (u,v) = H.(x,y,1) for every pixel of a w x h grid (what
 * np.apply_along_axis(homography_transformation, 2, template, H) computes) and max over the grid of max(u,v) --
 * the per-frame value whose maximum over the video is written to metrics_file.txt.  d_field (device,
 * f64[n,h,w,2], may be NULL) receives the field; h_max f64[n] the maxima.  Synchronises.                     */
int evh_fixed_plane_field(evh_ctx* ctx, const double* h_Hsup, int n, int w, int h, double* d_field, double* h_max);

/* ---- N1: the consumers of dict_with_homography_matrix.json ------------------------------------------------ */
/* utils.superposition_dict (utils.py:184-211): h_H f64[n,9] = the per-frame H in frame order -> h_out f64[n,9] the running
 * superposition: out[0] = H[0], out[i] = np.dot(H[i], out[i-1]) / [2][2] (matrix_superposition, utils.py:139-145; with
 * n = 2 it is one matrix_superposition(H[1], H[0], False)).  Sequential scan on the device.  Synchronises.            */
int evh_superposition_scan(evh_ctx* ctx, const double* h_H, int n, double* h_out);
/* fixed_coordinate_system.from_original_to_fix / from_fix_to_original (fixed_coordinate_system.py:19-69, 72-122) and
 * utils.homography_transformation (utils.py:89-92), batched: point i = (x, y) of h_pts f64[n,2] ->
 * np.around(np.dot(M[h_idx[i]], (kx*x, ky*y, 1))[:2] / [2], decimals) into h_out f64[n,2]; h_M f64[nmat,9] holds the
 * superposed H per frame (or, for the inverse direction, their inverses); decimals < 0: no rounding.  Synchronises. */
int evh_transform_points(evh_ctx* ctx, const double* h_M, int nmat, const int32_t* h_idx, const double* h_pts, int n,
                         double kx, double ky, int decimals, double* h_out);

/* ---- K1..K6: ORB detectAndCompute on a batch of frames ------------------------------------------------------- */
/* channels: 1 (gray) or 3 (BGR, converted like cvtColor(BGR2GRAY)).  Results stay resident in the context
 * (frame slots 0..nframes-1) until the next call.                                                              */
int evh_orb_detect_batch(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int w, int h, int channels,
                         int64_t row_stride, int64_t frame_stride, int nfeatures);
/* FAST threshold lifting (default on): the exact corner score is only evaluated for pixels that can reach the
 * per-(frame, level) score that ORB's retainBest(2*quota) will cut at; keypoints and descriptors are identical
 * either way (a level that comes up short is redone at threshold 20).  With lifting off the candidate lists
 * returned by evh_orb_download_candidates hold every FAST corner at threshold 20.                          */
int evh_set_fast_lift(evh_ctx* ctx, int on);
/* Order (and, with ties at the cut, the set) in which ORB's KeyPointsFilter::retainBest leaves a level's key points --
 * frame_processing.py:59-61 cv2.ORB_create().detectAndCompute; OpenCV 3.4.2 features2d/src/keypoint.cpp:
 *     std::nth_element(begin, begin + n, end, greater-response); amb = kp[n - 1].response;
 *     std::partition(begin + n, end, response >= amb)
 * EVH_ORDER_OPENCV (default): the permutation libstdc++'s introselect / partition leave on the row-major list of a level's
 *   FAST corners -- what the reference's run produces; the order of the matches, and through it the minimal samples RANSAC
 *   draws, follow from it (pinned by the reference's dict_with_homography_matrix.json, tests/test_capture_golden.py).
 *   Needs every corner at threshold 20, so FAST threshold lifting does not apply.
 * EVH_ORDER_CANONICAL: every tie at the cut kept, key points in (level, y, x) order (rounds 1-3); faster (lifting applies),
 *   H agrees with the reference only where RANSAC's consensus does not depend on the draw. */
enum { EVH_ORDER_CANONICAL = 0, EVH_ORDER_OPENCV = 1 };
/* The 8x8 linear systems of findHomography's Levenberg-Marquardt refinement (matching.py:156-157, utils.py:356-358 ->
 * cv::findHomography -> LMSolver -> cv::solve(DECOMP_EIG)).
 * EVH_SOLVER_EXACT (default): the operator's own Jacobi eigen-solve, rotation by rotation in its order -- H bit-identical to the
 *   CPU restatement.
 * EVH_SOLVER_EXACT (default) also means: the operator's bit-exact result on the reference's own video (all 120 recorded matrices).
 * EVH_SOLVER_FAST ("tolerance mode"): after RANSAC -- whose random draw, hypotheses and inlier masks stay exact, so statuses are
 *   identical -- (1) LM's 8x8 systems by LDL^T (a non-positive pivot falls back to the exact path), (2) the sums of the refit and
 *   of the LM evaluations by per-lane partial sums and a tree instead of the operator's point order, (3) the refit itself by the
 *   inhomogeneous least-squares solution with h33 = 1 in the normalised frame (one 8x8 LDL^T; eigen-solve as the fall-back) --
 *   it only seeds LM.  A stream pair costs 2.5-3x less (bench.py --config 3: 2.0 k -> 5.8 k pairs/s; the reference's default
 *   detector list at 400x224: 0.85 k -> 1.8 k), but H is no longer OpenCV's to the digit: the systems are graded over 14 orders
 *   of magnitude (raw pixel coordinates), the loop is cut after 10 iterations, and where the data do not determine H the end
 *   points differ -- measured over 140 pairs: frame corners up to 5.3e-4 px, SURVEY 8d's floored-relative h_err up to 2.5e-3
 *   (tests/test_gpu_parity.py::test_fast_solver_mode; bars 5e-3 px / 2e-2).  For callers who need the geometry, not the
 *   operator's digits. */
enum { EVH_SOLVER_EXACT = 0, EVH_SOLVER_FAST = 1 };
int evh_set_solver_mode(evh_ctx* ctx, int mode);
int evh_get_solver_mode(const evh_ctx* ctx);
/* largest max_features evh_create accepts (the matching filter keeps five lists of a frame slot's rows in LDS) */
#define EVH_MAX_FEATURES 5984
int evh_set_keypoint_order(evh_ctx* ctx, int mode);
int evh_get_keypoint_order(const evh_ctx* ctx);
/* The pair / stream entries below additionally let the second frame of a pair -- every other frame of a stream --
 * borrow the sampled score histogram of the frame before it (default on): consecutive video frames look alike, and
 * a threshold that proves too high is redone at 20 like any other, so results never change.  Turn it off for
 * batches of UNRELATED frame pairs, where the borrowed threshold is often wrong and the redo costs more than the
 * sampling it saves.  evh_orb_detect_batch itself always samples every frame.                                */
int evh_set_fast_share(evh_ctx* ctx, int on);
/* A context also carries, per pyramid level, the lower quartile of the lifted thresholds of its previous detect call as
 * a hint for the next call's sampling pass (lifted instead of dense scoring of the sampled tiles; default on, results
 * never change).  Off = every call samples densely, as a context's first call does.                            */
int evh_set_fast_hint(evh_ctx* ctx, int on);
/* number of keypoints of a frame slot, or <0 */
int evh_orb_count(evh_ctx* ctx, int frame);
/* capacity (rows) a caller must provide to evh_orb_download */
/* N2, fused ingest: the same detect on frames of src_w x src_h that the reference would first shrink with
 * imutils.resize(frame, width=w) (video_processing.py:62,73): level 0 is produced straight from the full-size frame
 * (INTER_AREA per channel, rounded to uint8 as the resized image would be, then the gray weights) -- the resized BGR
 * image is never materialised.  (w, h) is the working size: h = int(src_h * (w / float(src_w))); it may
 * also be LARGER than the source (resize_width > frame width: INTER_AREA's bilinear emulation).                       */
int evh_orb_detect_batch_resized(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                                 int64_t row_stride, int64_t frame_stride, int w, int h, int nfeatures);
int evh_orb_capacity(const evh_ctx* ctx);
/* copies one frame's features to HOST arrays (any pointer may be NULL): xy f32[n,2] (level-0 pixels, what
 * the reference keeps from kp.pt), desc u8[n,32], octave i32[n], lxy i32[n,2] (level coordinates),
 * response f32[n], angle f32[n] (degrees).  Returns n. Canonical order: (octave, y, x).                      */
int evh_orb_download(evh_ctx* ctx, int frame, float* h_xy, uint8_t* h_desc, int32_t* h_octave, int32_t* h_lxy,
                     float* h_response, float* h_angle);
/* single-frame convenience form of detectAndCompute (frame_processing.py:60-61): one device frame in (tight
 * rows: row_stride = w * channels), features to HOST arrays sized evh_orb_capacity() rows (any may be NULL),
 * number of key points in *h_count.  Equivalent to evh_orb_detect_batch(nframes = 1) + evh_orb_download(0).   */
int evh_orb_detect_compute(evh_ctx* ctx, const uint8_t* d_frame, int w, int h, int channels, int nfeatures,
                           float* h_xy, uint8_t* h_desc, int32_t* h_octave, int* h_count);
/* test/inspection hooks: pyramid level geometry + contents, FAST candidates (packed score<<24|y<<12|x) */
int evh_orb_level_info(const evh_ctx* ctx, int level, int* w, int* h, int* quota, float* scale);
int evh_orb_download_level(evh_ctx* ctx, int frame, int level, uint8_t* h_pixels /* h*w tight */);
int evh_orb_download_candidates(evh_ctx* ctx, int frame, int level, uint32_t* h_packed, int cap);

/* ---- K7: brute-force 2-NN, L2 over the 32 descriptor bytes (squared distances, exact integers) --------------- */
/* d_idx i32[nq,2] (-1 = missing neighbour), d_d2 u32[nq,2].  Ties -> lowest train index.                      */
int evh_match_knn2_l2u8(evh_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx,
                        uint32_t* d_d2);
/* the same on 128-byte rows: SIFT's descriptor values (0..255 each, held by the operator as float32; its float sums of
 * squared differences are exact integers).  Squared distances up to 8 323 200: neighbours are compared AFTER sqrt in
 * float32 like the operator does (two different D above 2^22 can round to the same distance and then tie).          */
int evh_match_knn2_l2u8x128(evh_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx,
                            uint32_t* d_d2);
int evh_match_knn2_hamming(evh_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx,
                           uint32_t* d_d2);
/* ratio test (d0 < d1*ratio on f32 sqrt distances, evaluated in f64), one-to-one filter, duplicate-coordinate
 * filter.  d_pts f32[nq,4] receives (ax, ay, bx, by) rows; h_count the number of rows; h_status 0 or
 * EVH_PAIR_FEW_MATCHES.  a = query (current frame), b = train (previous frame).                              */
int evh_ratio_unique_filter(evh_ctx* ctx, const int32_t* d_idx, const uint32_t* d_d2, int nq, int nt,
                            const float* d_xy_q, const float* d_xy_t, double ratio, int min_matches, float* d_pts,
                            int* h_count, int* h_status);

/* ---- K8/K9: findHomography(RANSAC) ---------------------------------------------------------------------------- */
/* d_pts f32[n,4] rows (ax, ay, bx, by): H maps a -> b.  h_H f64[9], h_mask u8[n] (may be NULL),
 * h_found 0/1, h_info i32[3] = {ransac iterations, best inlier count, LM iterations} (may be NULL).           */
int evh_find_homography_ransac(evh_ctx* ctx, const float* d_pts, int n, double thr, int max_iters, double conf,
                               double* h_H, uint8_t* h_mask, int* h_found, int* h_info);
/* The same with the iteration bound held fixed: exactly max(max_iters, 1) accepted samples are evaluated (the
 * adaptive bound RANSACUpdateNumIters is not applied).  This is the force_max_iters mode of the fused entries below
 * -- the fixed-iteration workload of BASELINE.json configs[2] -- exposed for one point set.                       */
int evh_find_homography_ransac_fixed(evh_ctx* ctx, const float* d_pts, int n, double thr, int max_iters, double conf,
                                     double* h_H, uint8_t* h_mask, int* h_found, int* h_info);
/* find_point_displacement + get_largest_group_points: rows of the most populated rounded-displacement bin */
int evh_static_filter(evh_ctx* ctx, const double* h_H, const float* d_pts, int n, float* d_out_pts, int* h_count);

/* ---- fused batch entry: frames -> H -------------------------------------------------------------------------------- */
/* mode EVH_MODE_INDEPENDENT_PAIRS: 2*npairs frames laid out (prev0, cur0, prev1, cur1, ...), H_sup = None.
 * mode EVH_MODE_STREAM: npairs+1 consecutive frames, reference stream semantics (running superposition,
 * none_H_processing=True: a failed pair repeats the previous H; a failed FIRST pair gets NaNs + its status).
 * d_H f64[npairs,9] and d_status i32[npairs] are device buffers. Does not synchronise.                        */
int evh_pair_homography_batch(evh_ctx* ctx, const uint8_t* d_frames, int npairs, int mode, int w, int h,
                              int channels, int64_t row_stride, int64_t frame_stride, int nfeatures,
                              double ransac_thr, int ransac_max_iters, double ransac_conf, int force_max_iters,
                              double* d_H, int32_t* d_status);
/* Stream form with explicit carry-over so that a long video can be processed in chunks: nframes consecutive
 * frames -> nframes-1 pairs.  d_state_in: f64[18] = {H_sup(9), H_prev(9)} leaving the previous chunk, or NULL for
 * the first chunk of a stream; d_state_out: f64[18] receives the state after the last pair (may be NULL).
 * Consecutive chunks overlap by one frame.  Does not synchronise.                                              */
int evh_stream_homography_batch(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int w, int h, int channels,
                                int64_t row_stride, int64_t frame_stride, int nfeatures, double ransac_thr,
                                int ransac_max_iters, double ransac_conf, int force_max_iters,
                                const double* d_state_in, double* d_state_out, double* d_H, int32_t* d_status);
/* The stream form on full-size frames with the reference's resize_width fused in (see evh_orb_detect_batch_resized):
 * what get_homography_dict(capture, resize_width=w) runs per chunk.                                               */
int evh_stream_homography_batch_resized(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int src_w, int src_h,
                                        int channels, int64_t row_stride, int64_t frame_stride, int w, int h,
                                        int nfeatures, double ransac_thr, int ransac_max_iters, double ransac_conf,
                                        int force_max_iters, const double* d_state_in, double* d_state_out,
                                        double* d_H, int32_t* d_status);
/* Several independent streams in one batch: d_frames holds nstreams x frames_per_stream frames, stream-major
 * (all frames of stream 0, then stream 1, ...).  Everything up to the static filter runs over all frames / pairs
 * at once; the sequential scans of the streams then run concurrently, one wavefront per stream, each with its own
 * {H_sup, H_prev}: d_state_in / d_state_out f64[nstreams][18] (d_state_in NULL = every stream starts here).
 * d_H f64[nstreams][frames_per_stream-1][9], d_status i32[nstreams][frames_per_stream-1].  Per stream the results
 * are bit-identical to evh_stream_homography_batch on that stream alone.  Does not synchronise.                 */
int evh_multi_stream_homography_batch(evh_ctx* ctx, const uint8_t* d_frames, int nstreams, int frames_per_stream,
                                      int w, int h, int channels, int64_t row_stride, int64_t frame_stride,
                                      int nfeatures, double ransac_thr, int ransac_max_iters, double ransac_conf,
                                      int force_max_iters, const double* d_state_in, double* d_state_out,
                                      double* d_H, int32_t* d_status);
/* Two-phase form of the stream path, for sharding ONE stream over several GPUs with the reference's semantics
 * (video_processing.py:67-105).  Phase 1 is independent per pair and runs wherever the frames are:
 * detect, match, ratio/unique filter, RANSAC #1, static filter (frame_processing.py:91-98, matching.py:131-163) on
 * nframes consecutive frames -> nframes-1 pairs; it leaves per pair the static rows f32[row_cap,4] (ax,ay,bx,by),
 * their count and the phase-1 status (EVH_PAIR_*) in CALLER device buffers; row_cap must equal evh_orb_capacity().
 * Phase 2 is the sequential part (compute_homography with the running superposition, utils.py:328-363, and
 * matrix_superposition, utils.py:118-145): it scans npairs pairs (any number -- rows gathered from all shards) in
 * stream order from d_state_in (f64[18] {H_sup, H_prev} or NULL at the start of the stream) and writes H f64[9] and
 * the final status per pair, plus the state after the last pair.  Running phase 1 on blocks that overlap by one
 * frame and phase 2 once over the concatenated rows gives bit-identical results to evh_stream_homography_batch on
 * the whole stream.  Neither call synchronises.                                                                  */
int evh_stream_static_batch(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int w, int h, int channels,
                            int64_t row_stride, int64_t frame_stride, int nfeatures, double ransac_thr,
                            int ransac_max_iters, double ransac_conf, int force_max_iters, float* d_rows, int row_cap,
                            int32_t* d_counts, int32_t* d_status1);
int evh_stream_scan(evh_ctx* ctx, const float* d_rows, int row_cap, const int32_t* d_counts, const int32_t* d_status1,
                    int npairs, double ransac_thr, int ransac_max_iters, double ransac_conf, int force_max_iters,
                    const double* d_state_in, double* d_state_out, double* d_H, int32_t* d_status);
/* the same per-pair body starting from features already resident in the context (frame slots): used by the
 * Python FrameProcessing/KeyPoints mirror.  cur/prev are frame slots of the last evh_orb_detect_batch.
 * h_Hsup: f64[9] or NULL.  h_H f64[9]; returns the pair status in *h_status.                                 */
int evh_pair_from_slots(evh_ctx* ctx, int cur_slot, int prev_slot, const double* h_Hsup, double* h_H,
                        int* h_status);
/* KeyPoints.match_static_kps on resident slots: static point rows to host, f32[n,4].                          */
int evh_match_static_from_slots(evh_ctx* ctx, int cur_slot, int prev_slot, float* h_pts, int cap, int* h_count,
                                int* h_status);
/* compute_homography (utils.py:328-363) on host point rows f32[n,4] */
int evh_compute_homography(evh_ctx* ctx, const float* h_pts, int n, const double* h_Hsup, double* h_H,
                           int* h_status);

/* ---- N4: SIFT and the reference's multi-type pairs --------------------------------------------------------- */
/* feature types of frame_processing.py:59-67; a list is processed in its own order (reference default: SURF, SIFT, ORB) */
enum { EVH_FEATURE_ORB = 0, EVH_FEATURE_SIFT = 1, EVH_FEATURE_SURF = 2 };
/* Reserves the SIFT buffers of a context: max_sift_features key points per frame slot (SIFT_create() keeps every key
 * point -- 2 500 on a textured 400x224 frame), the float scale space of a group of frames.  Call once, before the entries
 * below.  A frame that delivers more is flagged: its pairs get EVH_PAIR_CAPACITY, evh_sift_count / _download fail.
 * Up to 65 535 per frame in the *_types entries (beyond 7 680 their matching filter works in global memory).         */
int evh_sift_enable(evh_ctx* ctx, int max_sift_features);
int evh_sift_capacity(const evh_ctx* ctx);
/* cv2.xfeatures2d.SIFT_create().detectAndCompute(frame, None) (frame_processing.py:62-64) on a batch of frames of
 * src_w x src_h, working size (w, h) as in evh_orb_detect_batch_resized (equal sizes = no resize).  Results stay
 * resident (frame slots 0..nframes-1).  OpenCV 3.4.2 defaults: 3 layers per octave, contrast 0.04, edge 10, sigma 1.6. */
int evh_sift_detect_batch(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                          int64_t row_stride, int64_t frame_stride, int w, int h);
int evh_sift_count(evh_ctx* ctx, int frame);
/* one frame's key points to HOST arrays (any may be NULL), in the operator's own order (removeDuplicatedSorted: by x,
 * then y, ...): xy f32[n,2], desc f32[n,128] (integer values 0..255 as the operator returns them), octave i32[n] (packed
 * like KeyPoint::octave: octave & 255 | layer << 8 | ...), size, angle (degrees), response.  Returns n.             */
int evh_sift_download(evh_ctx* ctx, int frame, float* h_xy, float* h_desc, int32_t* h_octave, float* h_size,
                      float* h_angle, float* h_response);
/* test / inspection hooks: octave geometry (returns 1 past the last octave) and one Gaussian layer (0..5) of the
 * scale space of a frame of the last detect call (octave 0 = the frame doubled)                                   */
int evh_sift_octave_info(const evh_ctx* ctx, int octave, int* w, int* h);
int evh_sift_download_gauss(evh_ctx* ctx, int frame, int octave, int layer, float* h_pixels /* h*w tight */);
/* SURF: cv2.xfeatures2d.SURF_create(extended=1, hessianThreshold=400).detectAndCompute(frame, None)
 * (frame_processing.py:65-67; OpenCV 3.4.2: 4 octaves x 3 layers, 128-float descriptors, rotation-aware).  Same call
 * shapes as the SIFT entries; hessian_threshold = 400 for the reference.  Key points come in the operator's own order
 * (std::sort by KeypointGreater: response descending); h_laplacian = KeyPoint::class_id (sign of the trace).        */
int evh_surf_enable(evh_ctx* ctx, int max_surf_features);
int evh_surf_capacity(const evh_ctx* ctx);
int evh_surf_detect_batch(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                          int64_t row_stride, int64_t frame_stride, int w, int h, double hessian_threshold);
int evh_surf_count(evh_ctx* ctx, int frame);
int evh_surf_download(evh_ctx* ctx, int frame, float* h_xy, float* h_desc /* f32[n,128] */, float* h_size, float* h_angle,
                      float* h_response, int32_t* h_octave, int32_t* h_laplacian);
/* test hook: the integral image (h+1) x (w+1) of a frame of the last SURF call                                      */
int evh_surf_download_integral(evh_ctx* ctx, int frame, int32_t* h_sum);
/* DescriptorMatcher("BruteForce").knnMatch(q, t, 2) on FLOAT descriptors (matching.py:102-108 with the float32[N,128]
 * rows of SIFT / SURF; dim = 64 or 128): d_idx i32[nq,2] (-1 = missing neighbour), d_dist f32[nq,2] (L2 distances,
 * summed in the operator's order).  Ties -> lowest train index.                                                      */
int evh_match_knn2_l2f32(evh_ctx* ctx, const float* d_q, int nq, const float* d_t, int nt, int dim, int32_t* d_idx,
                         float* d_dist);
/* evh_ratio_unique_filter on the float distances of evh_match_knn2_l2f32                                             */
int evh_ratio_unique_filter_f32(evh_ctx* ctx, const int32_t* d_idx, const float* d_dist, int nq, int nt,
                                const float* d_xy_q, const float* d_xy_t, double ratio, int min_matches, float* d_pts,
                                int* h_count, int* h_status);
/* The fused entries with a LIST of feature types (h_types: EVH_FEATURE_*, ntypes entries, processed in list order), i.e.
 * FrameProcessing(frame, features_type_list).concatenate_all_features_types (frame_processing.py:91-104): per type
 * detect + match + RANSAC #1 + static filter, the static rows of all types concatenated, remove_double_matching again,
 * then compute_homography.  A type that fails (NoMatchesException) fails the pair with its status.  (src_w, src_h) /
 * (w, h) as in evh_stream_homography_batch_resized.  Needs evh_sift_enable / evh_surf_enable (BEFORE the first such call)
 * when the list holds SIFT / SURF.  Do not synchronise.                                                                */
int evh_pair_homography_batch_types(evh_ctx* ctx, const uint8_t* d_frames, int npairs, int mode, int src_w, int src_h,
                                    int channels, int64_t row_stride, int64_t frame_stride, int w, int h, int nfeatures,
                                    const int32_t* h_types, int ntypes, double ransac_thr, int ransac_max_iters,
                                    double ransac_conf, int force_max_iters, double* d_H, int32_t* d_status);
int evh_stream_homography_batch_types(evh_ctx* ctx, const uint8_t* d_frames, int nframes, int src_w, int src_h,
                                      int channels, int64_t row_stride, int64_t frame_stride, int w, int h, int nfeatures,
                                      const int32_t* h_types, int ntypes, double ransac_thr, int ransac_max_iters,
                                      double ransac_conf, int force_max_iters, const double* d_state_in,
                                      double* d_state_out, double* d_H, int32_t* d_status);

#ifdef __cplusplus
}
#endif
#endif
