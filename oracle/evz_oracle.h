/*
 * oracle/evz_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain C++ restatement of the frame-to-frame homography hot path of gridl/EvenVizion
 * (reference: evenvizion/processing/{video_processing,frame_processing,matching,utils}.py) including the
 * OpenCV 3.4.2 operators the reference calls through cv2 (ORB detectAndCompute, BruteForce knnMatch,
 * findHomography(RANSAC), INTER_AREA resize).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * path (evenvizion_amd/ + libevhip.so) never includes, links or calls anything in oracle/.
 *
 * PARITY STATUS: the Python-glue semantics are pinned by fixtures captured from the reference itself
 * (tests/golden/glue_*.json) and by the reference's one known-answer artefact (metrics_file.txt).  The
 * OpenCV operator arithmetic is restated from the published OpenCV 3.4 algorithm (opencv-contrib-python
 * ==3.4.2.17 is a third-party dependency absent from /root/reference and from this image) and the
 * reference holds no test vectors at that boundary.  What it does hold is one end-to-end artefact of the real wheel: the H
 * dictionary its authors committed for evenvizion/examples/test_video/test_video.mp4.  Since round 4 that video is decoded by
 * this repository's own capture source and run through this oracle (tests/test_capture_golden.py, profiles/r04_golden_pinning.txt):
 * a free run reproduces all 120 recorded matrices to the last digit the JSON holds, which pins ALL the operators below jointly
 * and bit for bit -- including the order KeyPointsFilter::retainBest leaves ORB's key points in (evo_set_orb_order) and which
 * multiply-adds of SIFT's float Gaussian filter the wheel's AVX2/FMA3 build fuses (evo_set_sift_blur_mode).  What one video
 * cannot reach stays unpinned and is named where it lives: code paths its 121 frames never enter (e.g. introselect's heap-select
 * fall-back, RANSAC's degenerate-sample exits) and the float stand-ins whose last bit did not matter on it (cosf / sinf / powf
 * evaluated in double, exp32f's scalar remainder).
 */
#ifndef EVZ_ORACLE_H
#define EVZ_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* per-pair status; mirrors the failure list of the reference */
enum {
  EVO_OK = 0,
  EVO_NO_DESCRIPTORS = 1,     /* matching.py:104-107  descriptors is None                     */
  EVO_FEW_MATCHES = 2,        /* matching.py:113      len(matches) < min_matching_pts         */
  EVO_NO_PROVISIONAL_H = 3,   /* matching.py:158      findHomography #1 returned None         */
  EVO_LOW_INLIER_RATIO = 4,   /* utils.py:359         sum(mask) < 0.7*len(mask)               */
  EVO_NO_FINAL_H = 5          /* utils.py:361         findHomography #2 returned None         */
};

/* ---- K1 / K0: colour + resize (frame_processing.py:61 -> cvtColor; video_processing.py:62,73) ---- */
void evo_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray);
/* imutils.resize(width=) -> cv2.resize(INTER_AREA): area sums (shrink) or the bilinear emulation (enlarge); returns 0 */
int evo_resize_area(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh);
void evo_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);

/* ---- K2..K6: ORB (frame_processing.py:59-61 cv2.ORB_create().detectAndCompute) ---- */
/* fills lw/lh/lscale/lquota (8 entries each) */
void evo_orb_layout(int w, int h, int nfeatures, int* lw, int* lh, float* lscale, int* lquota);
/* writes the 8 levels tightly packed one after another; returns total bytes */
int64_t evo_orb_pyramid(const uint8_t* gray, int w, int h, uint8_t* out);
/* FAST-9/16 + 3x3 NMS on one image; row-major emission; returns count (may exceed cap; only cap written) */
int evo_fast_nms(const uint8_t* img, int w, int h, int threshold, int* xs, int* ys, int* scores, int cap);
void evo_fast_score_map(const uint8_t* img, int w, int h, int threshold, uint8_t* out);
/* candidates after border filter + retainBest(2*quota) for one level, row-major; returns count */
/* order (and, with ties at the cut, the set) KeyPointsFilter::retainBest leaves the key points in: 1 = OpenCV 3.4.2's call on
 * libstdc++'s nth_element/partition (default, pinned by the reference's golden), 0 = all ties kept in row-major order, 2..4 = the
 * other combinations of OpenCV's nth index and libstdc++'s pivot rule (evz_orb.cpp) */
void evo_set_orb_order(int mode);
int evo_get_orb_order(void);
long evo_orb_depth_limit_hits(void);   /* nth_element calls so far that fell back to heap select (introselect depth limit) */
/* which multiply-adds of SIFT's float Gaussian filter are fused: 2 = the vector bodies only (default, pinned by the reference's
 * golden: all 120 recorded pairs reproduced exactly), 0 = none, 1 = every column (evz_sift.cpp gaussian_blur) */
void evo_set_sift_blur_mode(int mode);
int evo_get_sift_blur_mode(void);
int evo_orb_level_candidates(const uint8_t* img, int w, int h, int quota, int* xs, int* ys, int* scores, int cap);
/* 7x7/sigma=2 8-bit Gaussian blur with reflect-101 borders (K6 first half) */
void evo_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst);
/* full detectAndCompute on a gray frame. Outputs (cap entries each): xy f32[cap,2], desc u8[cap,32],
 * octave i32, lx/ly i32 (level coords), response f32, angle f32 (degrees).  Order: per level, as retainBest leaves it (evo_set_orb_order).
 * Returns count. */
int evo_orb_detect(const uint8_t* gray, int w, int h, int nfeatures, float* xy, uint8_t* desc, int* octave,
                   int* lx, int* ly, float* response, float* angle, int cap);
/* deterministic sin/cos used for the steered pattern (checked against libm in tests) */
void evo_sincos(double x, double* s, double* c);
float evo_fast_atan2(float y, float x);

/* ---- N4: SIFT (frame_processing.py:62-64 cv2.xfeatures2d.SIFT_create().detectAndCompute), evz_sift.cpp ---- */
/* restated from recall; pinned jointly, since round 4, by the reference's own video and recorded result (tests/test_capture_golden.py: all 120 matrices of dict_with_homography_matrix.json reproduced to the last digit) (see the header of evz_sift.cpp) */
int evo_sift_layout(int w, int h, int* ow, int* oh, int cap);
int64_t evo_sift_gauss_pyramid(const uint8_t* gray, int w, int h, float* out, int64_t cap);
int evo_sift_detect(const uint8_t* gray, int w, int h, float* xy, uint8_t* desc, int* octave, float* size, float* angle,
                    float* response, int cap);
float evo_sift_exp32f(float x);
float evo_sift_exp2(float x);
/* BruteForce 2-NN on float32 descriptors (matching.py:102-108 for SIFT / SURF): idx i32[nq,2], dist f32[nq,2] */
void evo_knn2_l2f32(const float* q, int nq, const float* t, int nt, int dim, int32_t* idx, float* dist);
int evo_ratio_unique_f32(const int32_t* idx, const float* dist, int nq, double ratio, int32_t* out_q, int32_t* out_t);
int evo_match_static_f32(const float* xy_a, const float* desc_a, int na, const float* xy_b, const float* desc_b, int nb,
                         int dim, float* oa, float* ob, int* out_n);
/* ---- N4: SURF (frame_processing.py:65-67 SURF_create(extended=1, hessianThreshold=400).detectAndCompute), evz_surf.cpp ---- */
/* restated from recall; pinned jointly, since round 4, by the reference's own video and recorded result (tests/test_capture_golden.py: all 120 matrices of dict_with_homography_matrix.json reproduced to the last digit) (see the header of evz_surf.cpp) */
void evo_integral(const uint8_t* gray, int w, int h, int32_t* sum);
int evo_surf_detect(const uint8_t* gray, int w, int h, float* xy, float* desc, float* size, float* angle, float* response,
                    int* octave, int* laplacian, int cap);
/* one stream with a LIST of feature types (0 = ORB, 1 = SIFT, 2 = SURF), frame_processing.py:91-104 + video_processing.py:67-105 */
int evo_stream_gray_types(const uint8_t* frames, int nframes, int w, int h, int nfeatures, const int* types, int ntypes,
                          double* H, int* status);
/* force_max: all RANSACs run 2000 iterations; Hsup_forced (NULL or [F-1][9]): the plane pair k is solved in; npts (NULL or [F-1]) */
int evo_stream_gray_types_ex(const uint8_t* frames, int nframes, int w, int h, int nfeatures, const int* types, int ntypes,
                             int force_max, const double* Hsup_forced, double* H, int* status, int* npts);
int evo_match_static_f32_ex(const float* xy_a, const float* desc_a, int na, const float* xy_b, const float* desc_b, int nb,
                            int dim, int force_max, float* oa, float* ob, int* out_n);

/* ---- K7 + glue: matching (matching.py:75-129, 166-239; utils.py:41-68) ---- */
void evo_knn2_l2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, uint32_t* d2);
void evo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, uint32_t* d2);
/* lowes_ratio_test + filter_corresponding_points: returns M, writes (query,train) index pairs */
int evo_ratio_unique(const int32_t* idx, const uint32_t* d2, int nq, double ratio, int32_t* out_q, int32_t* out_t);
/* remove_double_matching: returns new count */
int evo_remove_double(const float* a, const float* b, int n, float* oa, float* ob);

/* ---- K8/K9: cv2.findHomography(a, b, RANSAC, thr) (matching.py:156; utils.py:356) ---- */
/* returns 1 when H found. info[0]=ransac iterations executed, info[1]=best inlier count, info[2]=LM iters */
int evo_find_homography(const float* a, const float* b, int n, double thr, int max_iters, double conf,
                        double* H, uint8_t* mask, int* info);
/* the same with force_max != 0: the iteration bound is never lowered -- all max(max_iters, 1) accepted samples are
 * evaluated (the fixed-iteration RANSAC of BASELINE configs[2]; not a reference mode, a stress variant of it) */
int evo_find_homography_ex(const float* a, const float* b, int n, double thr, int max_iters, double conf,
                           int force_max, double* H, uint8_t* mask, int* info);
/* building blocks exposed for tests */
int evo_dlt(const float* src, const float* dst, int n, double* H);
void evo_jacobi(double* A, int n, double* W, double* V);

/* ---- glue around the RANSACs (utils.py:258-363, matching.py:131-163) ---- */
/* find_point_displacement + get_largest_group_points: returns count of kept points */
int evo_static_filter(const double* H, const float* a, const float* b, int n, float* oa, float* ob);
/* compute_homography incl. optional pre-transform by Hsup (NULL = None). returns status (EVO_*) */
int evo_compute_homography(const float* a, const float* b, int n, const double* Hsup, double* H);
int evo_compute_homography_ex(const float* a, const float* b, int n, const double* Hsup, int force_max, double* H);
/* matrix_superposition (utils.py:118-145) */
void evo_matrix_superposition(const double* H, const double* Hsup, int first, double* out);

/* ---- whole path ---- */
/* KeyPoints.match_static_kps from two keypoint sets; returns status, static points in oa/ob (cap = nq) */
int evo_match_static(const float* xy_a, const uint8_t* desc_a, int na, const float* xy_b, const uint8_t* desc_b,
                     int nb, float* oa, float* ob, int* out_n);
int evo_match_static_ex(const float* xy_a, const uint8_t* desc_a, int na, const float* xy_b, const uint8_t* desc_b,
                        int nb, int force_max, float* oa, float* ob, int* out_n);
/* one frame pair from gray frames: cur = current frame (a), prev = previous frame (b) */
int evo_pair_gray(const uint8_t* cur, const uint8_t* prev, int w, int h, int nfeatures, const double* Hsup,
                  double* H);
/* B independent pairs, frames laid out [2B][h][w] as (prev0, cur0, prev1, cur1, ...); threads>=1 */
void evo_pairs_gray_batch(const uint8_t* frames, int npairs, int w, int h, int nfeatures, int threads,
                          double* H, int* status);
/* stream of F gray frames with the reference's running-superposition semantics (video_processing.py:67-105,
 * none_H_processing=True). H: [F-1][9]; status: [F-1]; a failed first pair yields status and stops (returns
 * the index of the failing pair, or -1 when all pairs were processed). */
int evo_stream_gray(const uint8_t* frames, int nframes, int w, int h, int nfeatures, double* H, int* status);
int evo_stream_gray_ex(const uint8_t* frames, int nframes, int w, int h, int nfeatures, int force_max, double* H,
                       int* status);

#ifdef __cplusplus
}
#endif
#endif
