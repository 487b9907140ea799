// evh_api.hip -- host side of libevhip.so: context, geometry, and the extern "C" entry points of include/evhip.h.
#include "evh_internal.h"
#include "evh_match.h"
#include "evh_ransac.h"
#include <cmath>
#include <cstring>
#include <vector>

static std::string g_create_error;

int evh_fail(evh_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}

static hipEvent_t prof_event(evh_ctx* c) {
  hipEvent_t e = nullptr;
  if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); return e; }
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
EvhProfScope::EvhProfScope(evh_ctx* ctx, int stage, hipStream_t on) : c(ctx), idx(-1), st(on) {
  if (!c || !c->profiling) return;
  if (!st) st = c->stream;
  evh_ctx::ProfSpan s{stage, prof_event(c), prof_event(c)};
  if (!s.a || !s.b) return;
  (void)hipEventRecord(s.a, st);
  c->prof_spans.push_back(s);
  idx = (int)c->prof_spans.size() - 1;
}
EvhProfScope::~EvhProfScope() {
  if (idx >= 0) (void)hipEventRecord(c->prof_spans[idx].b, st);
}

namespace {

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
inline int64_t align_up64(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int round_f(float v) { return (int)lrintf(v); }

// ORB_create() defaults: 8 levels, scale factor 1.2f (held as double), per-level size = cvRound(size / scale),
// per-level quota from the geometric series (orb.cpp); see SURVEY Appendix A.1.
void compute_geometry(int w, int h, int nfeatures, EvhGeom& g) {
  g.w = w; g.h = h; g.nfeatures = nfeatures;
  const double scaleFactor = (double)1.2f;
  int64_t off = 0, coff = 0;
  int tiles = 0, tab = 0;
  for (int l = 0; l < EVH_NLEVELS; l++) {
    EvhLevel& L = g.lv[l];
    L.scale = (float)std::pow(scaleFactor, (double)l);
    L.w = round_f((float)w / L.scale);
    L.h = round_f((float)h / L.scale);
    L.stride = align_up(L.w, 64);
    L.off = off;
    off += align_up64((int64_t)L.stride * L.h, 256);
    L.cand_cap = (L.w / 2 + 1) * (L.h / 2 + 1) + 64;  // NMS admits at most one corner per 2x2 block
    L.cand_off = coff;
    coff += L.cand_cap;
    // k_fast tiles (128 x 28) cover only what can be emitted: ORB drops corners within 31 px of the border
    // (runByImageBorder), so the tile grid starts at (EVH_FAST_OX, EVH_FAST_OY) = (24, 31) and ends at w-32 / h-32;
    // the one ring of neighbours NMS needs comes from the tiles' halo
    L.tiles_x = std::max(1, (L.w - 31 - EVH_FAST_OX + 127) / 128); L.tiles_y = std::max(1, (L.h - 31 - EVH_FAST_OY + 27) / 28);
    L.tile_start = tiles;
    tiles += L.tiles_x * L.tiles_y;
    L.tab_off = tab;
    if (l > 0) tab += 2 * L.w + 2 * L.h;
  }
  g.pyr_frame_bytes = off;
  g.cand_frame_entries = coff;
  g.total_tiles = tiles;
  const float factor = (float)(1.0 / scaleFactor);
  float ndes = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)EVH_NLEVELS));
  int sum = 0;
  for (int l = 0; l < EVH_NLEVELS - 1; l++) {
    g.lv[l].quota = round_f(ndes);
    sum += g.lv[l].quota;
    ndes *= factor;
  }
  g.lv[EVH_NLEVELS - 1].quota = std::max(nfeatures - sum, 0);
  int kb = 0;
  for (int l = 0; l < EVH_NLEVELS; l++) {   // room for ties at the Harris cut: quota + 25 % + 16 per level
    g.lv[l].kp_base = kb;
    g.lv[l].kp_cap = g.lv[l].quota + g.lv[l].quota / 4 + 16;
    kb += g.lv[l].kp_cap;
  }
}

// INTER_LINEAR_EXACT coefficient tables for one axis: offset of the left/top tap and the 8.8 weight of the
// right/bottom tap; samples that fall off either end are encoded as a full weight on the edge sample.
void linear_exact_tab(int ssize, int dsize, int* ofs, int* c1) {
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1.0 / inv_scale;
  for (int v = 0; v < dsize; v++) {
    const double fval = scale * ((double)v + 0.5) - 0.5;
    int ival = (int)std::floor(fval);
    if (ival >= 0 && ssize > 1) {
      if (ival < ssize - 1) {
        ofs[v] = ival;
        c1[v] = (int)std::lrint((fval - (double)ival) * 256.0);
      } else { ofs[v] = ssize - 2; c1[v] = 256; }
    } else { ofs[v] = 0; c1[v] = 0; }
  }
}

int kcap_for(int nfeatures) { return align_up(nfeatures + nfeatures / 4 + 8 * 17 + 64, 64); }

int configure(evh_ctx* c, int w, int h, int nfeatures) {
  if (c->geom_valid && c->g.w == w && c->g.h == h && c->g.nfeatures == nfeatures) return EVH_SUCCESS;
  if (w > c->max_w || h > c->max_h || w * (int64_t)h > (int64_t)c->max_w * c->max_h)
    return evh_fail(c, EVH_ERR_CAPACITY, "frame larger than the size given to evh_create");
  if (nfeatures > c->max_features || nfeatures < 1)
    return evh_fail(c, EVH_ERR_CAPACITY, "nfeatures outside [1, max_features]");
  if (w >= 4096 || h >= 4096) return evh_fail(c, EVH_ERR_UNSUPPORTED, "frames must be smaller than 4096 in each dimension");
  EvhGeom g;
  compute_geometry(w, h, nfeatures, g);
  EvhGeom gmax;
  compute_geometry(c->max_w, c->max_h, c->max_features, gmax);
  if (g.pyr_frame_bytes > gmax.pyr_frame_bytes || g.cand_frame_entries > gmax.cand_frame_entries)
    return evh_fail(c, EVH_ERR_CAPACITY, "geometry exceeds the buffers sized by evh_create");
  std::vector<int> tabs;
  for (int l = 1; l < EVH_NLEVELS; l++) {
    const EvhLevel& S = g.lv[l - 1];
    const EvhLevel& D = g.lv[l];
    size_t base = tabs.size();
    tabs.resize(base + 2 * D.w + 2 * D.h);
    linear_exact_tab(S.w, D.w, &tabs[base], &tabs[base + D.w]);
    linear_exact_tab(S.h, D.h, &tabs[base + 2 * D.w], &tabs[base + 2 * D.w + D.h]);
  }
  // stream-ordered upload through pinned staging: kernels already enqueued keep the old tables, later ones see the new
  EVH_HIP(c, hipEventSynchronize(c->ev_tabs));                       // the previous upload has left h_tabs
  std::memcpy(c->h_tabs, tabs.data(), tabs.size() * sizeof(int));
  EVH_HIP(c, hipMemcpyAsync(c->d_tabs, c->h_tabs, tabs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  EVH_HIP(c, hipEventRecord(c->ev_tabs, c->stream));
  c->g = g;
  c->geom_valid = true;
  return EVH_SUCCESS;
}

template <class T>
int dalloc(evh_ctx* c, T** p, size_t n) {
  EVH_HIP(c, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  c->bytes_allocated += n * sizeof(T);
  return EVH_SUCCESS;
}

// fixed-iteration mode keeps the per-lane eigenvector matrices of its hypotheses in a global scratch (one block per
// workgroup = per pair slot); allocated on first use
int ensure_lane_scratch(evh_ctx* c) {
  if (c->d_lane_v) return EVH_SUCCESS;
  return dalloc(c, &c->d_lane_v, (size_t)c->max_frames * EVH_LANE_V_DOUBLES);
}

EvhRansacArgs pair_ransac_args(evh_ctx* c, double thr, int max_iters, double conf, int force_max) {
  EvhRansacArgs R{};
  R.fast_solver = c->solver_mode;
  if (force_max && ensure_lane_scratch(c) == EVH_SUCCESS) R.lane_v = c->d_lane_v;
  R.pts = c->d_pts; R.pts2 = c->d_pts2; R.row_stride = c->kcap; R.npts = c->d_npts; R.npts2 = c->d_npts2;
  R.status = c->d_pstatus; R.thr = thr; R.max_iters = max_iters; R.conf = conf; R.force_max = force_max;
  R.mask = c->d_mask; R.crow = c->d_crow; R.lm = c->d_lm; R.H1 = c->d_H1; R.info = c->d_info;
  return R;
}

// K7 + glue on resident slots for `npairs` pairs
int match_pairs(evh_ctx* c, int npairs, int q0, int qstep, int t0, int tstep) {
  EvhKnnArgs K{};
  K.q = c->d_desc; K.t = c->d_desc; K.slot_bytes = (int64_t)c->kcap * 32;
  K.nq_arr = c->d_kp_count; K.nt_arr = c->d_kp_count;
  K.q_slot0 = q0; K.q_slot_step = qstep; K.t_slot0 = t0; K.t_slot_step = tstep;
  K.idx = c->d_knn_idx; K.d2 = c->d_knn_d2; K.out_stride = c->kcap; K.hamming = 0;
  int rc;
  { EvhProfScope ps(c, EVH_ST_KNN); rc = evh_launch_knn2(c, K, npairs); }
  if (rc) return rc;
  EvhFilterArgs F{};
  F.idx = c->d_knn_idx; F.d2 = c->d_knn_d2; F.knn_stride = c->kcap;
  F.xy_q = c->d_kp_xy; F.xy_t = c->d_kp_xy; F.xy_slot_floats = (int64_t)c->kcap * 2;
  F.nq_arr = c->d_kp_count; F.nt_arr = c->d_kp_count; F.flags_arr = c->d_frame_flags;
  F.q_slot0 = q0; F.q_slot_step = qstep; F.t_slot0 = t0; F.t_slot_step = tstep;
  F.ratio = 0.5; F.min_matches = 4;  // constants.py:25,28 (LOWES_RATIO, MINIMUM_MATCHING_POINTS)
  F.pts = c->d_pts; F.pts_stride = c->kcap; F.npts = c->d_npts; F.status = c->d_pstatus; F.kcap = c->kcap;
  // the filter overwrites the matched-row buffers the previous batch's (asynchronous) solve may still be reading
  if (c->solve_pending) EVH_HIP(c, hipStreamWaitEvent(c->stream, c->ev_solve_done, 0));
  EvhProfScope ps(c, EVH_ST_FILTER);
  return evh_launch_filter(c, F, npairs);
}

// context-owned scratch of the host-pointer entries (evh_transform_points, evh_superposition_scan,
// evh_fixed_plane_field): grown on demand, reused across calls (those entries synchronise before returning)
int ensure_scratch(evh_ctx* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return EVH_SUCCESS;
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  if (c->d_scratch) { (void)hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratch_bytes = 0; }
  const size_t want = std::max(bytes, (size_t)1 << 16);
  EVH_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_scratch), want));
  c->scratch_bytes = want;
  return EVH_SUCCESS;
}

// entry points that reuse the pair buffers on the main stream first order themselves behind a pending async solve
int join_solve(evh_ctx* c) {
  if (c->solve_pending) EVH_HIP(c, hipStreamWaitEvent(c->stream, c->ev_solve_done, 0));
  return EVH_SUCCESS;
}

// RANSAC #1 + static filter, then compute_homography, on the solve stream when asynchronous solve is enabled
int solve_pairs(evh_ctx* c, EvhRansacArgs R, int npairs, int nstreams = 0, int pairs_per_stream = 0, int pitch = 0) {
  hipStream_t main = c->stream;
  const bool async = c->async_solve && c->solve_stream;
  if (async) {
    EVH_HIP(c, hipEventRecord(c->ev_match_done, main));
    EVH_HIP(c, hipStreamWaitEvent(c->solve_stream, c->ev_match_done, 0));
    c->stream = c->solve_stream;          // the launchers enqueue on c->stream
  }
  int rc;
  { EvhProfScope ps(c, EVH_ST_RANSAC_STATIC, c->stream); rc = evh_launch_ransac_static(c, R, npairs); }
  if (!rc) {
    EvhProfScope ps(c, EVH_ST_RANSAC_FINAL, c->stream);
    rc = evh_launch_ransac_final(c, R, nstreams > 0 ? pairs_per_stream : npairs, nstreams, pitch);
  }
  if (async) {
    hipError_t e = hipEventRecord(c->ev_solve_done, c->solve_stream);
    c->stream = main;
    c->solve_pending = true;
    if (e != hipSuccess) return evh_fail(c, EVH_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
  }
  return rc;
}


// level 0 (gray) of every frame: (sw, sh) = size of the frames handed over, (w, h) = working size.  Different sizes =
// fused ingest (N2).  Shared by every feature type of a call.
int ingest_level0(evh_ctx* c, const char* who, const uint8_t* d_frames, int nframes, int sw, int sh, int w, int h,
                         int channels, int64_t row_stride, int64_t frame_stride, int nfeatures) {
  if (!c || !d_frames) return evh_fail(c, EVH_ERR_INVALID, std::string(who) + ": NULL argument");
  if (nframes < 1 || nframes > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "nframes exceeds max_frames");
  if (channels != 1 && channels != 3) return evh_fail(c, EVH_ERR_INVALID, "channels must be 1 or 3");
  if (row_stride < (int64_t)sw * channels) return evh_fail(c, EVH_ERR_INVALID, "row_stride smaller than a row");
  if (sw < 1 || sh < 1) return evh_fail(c, EVH_ERR_INVALID, "empty source frame");
  if (nframes > 65535 || h > 65535) return evh_fail(c, EVH_ERR_CAPACITY, "too many frames / rows for one launch");
  int rc = configure(c, w, h, nfeatures);
  if (rc) return rc;
  EvhProfScope ps(c, EVH_ST_GRAY);
  if (sw == w && sh == h) return evh_launch_gray_level0(c, d_frames, nframes, channels, row_stride, frame_stride);
  return evh_launch_ingest_level0(c, d_frames, nframes, sw, sh, channels, row_stride, frame_stride, w, h);
}

// ORB K2..K6 on the frames whose level 0 is resident
int orb_stages(evh_ctx* c, int nframes, int share_group) {
  int rc;
  { EvhProfScope ps(c, EVH_ST_PYRAMID); rc = evh_launch_pyramid(c, nframes); }
  if (rc) return rc;
  { EvhProfScope ps(c, EVH_ST_FAST); rc = evh_launch_fast(c, nframes, share_group); }
  if (rc) return rc;
  { EvhProfScope ps(c, EVH_ST_SELECT); rc = evh_launch_select(c, nframes); }
  if (rc) return rc;
  { EvhProfScope ps(c, EVH_ST_DESCRIBE); rc = evh_launch_describe(c, nframes); }
  if (rc) return rc;
  c->nframes_resident = nframes;
  return EVH_SUCCESS;
}

int detect_batch(evh_ctx* c, const uint8_t* d_frames, int nframes, int sw, int sh, int w, int h, int channels,
                        int64_t row_stride, int64_t frame_stride, int nfeatures) {
  if (!c) return EVH_ERR_INVALID;
  const int share_group = c->fast_share ? c->fast_share_group : 0;   // set by the pair / stream entries for THIS call only
  c->fast_share_group = 0;
  int rc = ingest_level0(c, "evh_orb_detect_batch", d_frames, nframes, sw, sh, w, h, channels, row_stride, frame_stride, nfeatures);
  if (rc) return rc;
  return orb_stages(c, nframes, share_group);
}


// ---- multi-type pairs (frame_processing.py:91-104) ---------------------------------------------------------------------------
int ensure_multitype(evh_ctx* c) {
  if (c->mt.cap) return EVH_SUCCESS;
  const int each = std::max(c->kcap, std::max(c->sift_cap, c->surf_cap)), cap = c->kcap + c->sift_cap + c->surf_cap;
  if (each > 65536)
    return evh_fail(c, EVH_ERR_CAPACITY, "multi-type pairs: at most 65536 key points per frame and type");
  const size_t P = (size_t)c->max_frames, K = (size_t)cap;
  int rc;
#define M_(call) if ((rc = (call)) != EVH_SUCCESS) return rc
  M_(dalloc(c, &c->mt.knn_idx, P * K * 2));
  M_(dalloc(c, &c->mt.knn_d2, P * K * 2));
  M_(dalloc(c, &c->mt.pts, P * K * 4));
  M_(dalloc(c, &c->mt.pts2, P * K * 4));
  M_(dalloc(c, &c->mt.crow, P * K * 4));
  M_(dalloc(c, &c->mt.npts, P));
  M_(dalloc(c, &c->mt.npts2, P));
  M_(dalloc(c, &c->mt.pstatus, P));
  M_(dalloc(c, &c->mt.H1, P * 9));
  M_(dalloc(c, &c->mt.mask, P * K));
  M_(dalloc(c, &c->mt.lm, P * K * 4));
  M_(dalloc(c, &c->mt.info, P * 8));
  M_(dalloc(c, &c->d_acc, P * K * 4));
  M_(dalloc(c, &c->d_nacc, P));
  M_(dalloc(c, &c->d_accstatus, P));
#undef M_
  c->mt.cap = cap;
  return EVH_SUCCESS;
}

EvhRansacArgs mt_ransac_args(evh_ctx* c, double thr, int max_iters, double conf, int force_max) {
  EvhRansacArgs R{};
  R.fast_solver = c->solver_mode;
  if (force_max && ensure_lane_scratch(c) == EVH_SUCCESS) R.lane_v = c->d_lane_v;
  const EvhPairBufs& B = c->mt;
  R.pts = B.pts; R.pts2 = B.pts2; R.row_stride = B.cap; R.npts = B.npts; R.npts2 = B.npts2;
  R.status = B.pstatus; R.thr = thr; R.max_iters = max_iters; R.conf = conf; R.force_max = force_max;
  R.mask = B.mask; R.crow = B.crow; R.lm = B.lm; R.H1 = B.H1; R.info = B.info;
  return R;
}

// K7 + glue of ONE feature type into the multi-type pair buffers
int match_pairs_type(evh_ctx* c, int type, int npairs, int q0, int qstep, int t0, int tstep) {
  const EvhPairBufs& B = c->mt;
  const bool sift = type == EVH_FEATURE_SIFT, surf = type == EVH_FEATURE_SURF;
  const int* counts = sift ? c->d_sift_count : surf ? c->d_surf_count : c->d_kp_count;
  const int tcap = sift ? c->sift_cap : surf ? c->surf_cap : c->kcap;
  int rc;
  if (surf) {            // real-valued float rows: the float matcher, distances carried as float bits
    EvhKnnF32Args K{};
    K.q = c->d_surf_desc; K.t = c->d_surf_desc; K.dim = 128; K.n_arr = counts; K.slot_floats = (int64_t)tcap * 128;
    K.q_slot0 = q0; K.q_slot_step = qstep; K.t_slot0 = t0; K.t_slot_step = tstep;
    K.idx = B.knn_idx; K.dist = reinterpret_cast<float*>(B.knn_d2); K.out_stride = B.cap;
    { EvhProfScope ps(c, EVH_ST_KNN); rc = evh_launch_knn2_f32(c, K, npairs); }
  } else {
    EvhKnnArgs K{};
    K.q = sift ? c->d_sift_desc : c->d_desc; K.t = K.q;
    K.slot_bytes = sift ? (int64_t)tcap * 128 : (int64_t)tcap * 32;
    K.desc_bytes = sift ? 128 : 32;
    K.nq_arr = counts; K.nt_arr = counts;
    K.q_slot0 = q0; K.q_slot_step = qstep; K.t_slot0 = t0; K.t_slot_step = tstep;
    K.idx = B.knn_idx; K.d2 = B.knn_d2; K.out_stride = B.cap; K.hamming = 0;
    { EvhProfScope ps(c, EVH_ST_KNN); rc = evh_launch_knn2(c, K, npairs); }
  }
  if (rc) return rc;
  EvhFilterArgs F{};
  F.idx = B.knn_idx; F.d2 = B.knn_d2; F.knn_stride = B.cap; F.d2_is_dist = surf ? 1 : 0;
  F.xy_q = sift ? c->d_sift_xy : surf ? c->d_surf_xy : c->d_kp_xy; F.xy_t = F.xy_q;
  F.xy_slot_floats = (int64_t)tcap * 2;
  F.nq_arr = counts; F.nt_arr = counts; F.flags_arr = sift ? c->d_sift_flags : surf ? c->d_surf_flags : c->d_frame_flags;
  F.q_slot0 = q0; F.q_slot_step = qstep; F.t_slot0 = t0; F.t_slot_step = tstep;
  F.ratio = 0.5; F.min_matches = 4;
  F.pts = B.pts; F.pts_stride = B.cap; F.npts = B.npts; F.status = B.pstatus;
  F.kcap = std::max(c->kcap, std::max(c->sift_cap, c->surf_cap));
  EvhProfScope ps(c, EVH_ST_FILTER);
  return evh_launch_filter(c, F, npairs);
}

// frames -> H with a LIST of feature types, in list order (the reference's default list is SURF, SIFT, ORB):
// per type detect, match, RANSAC #1, static filter; concatenate; remove_double_matching; compute_homography
int pairs_types(evh_ctx* c, const char* who, const uint8_t* d_frames, int nframes, int npairs, int stream_mode, int sw, int sh,
                int w, int h, int channels, int64_t row_stride, int64_t frame_stride, int nfeatures, const int* types,
                int ntypes, double thr, int max_iters, double conf, int force_max, const double* d_state_in, double* d_state_out,
                double* d_H, int32_t* d_status) {
  if (!types || ntypes < 1 || ntypes > 8) return evh_fail(c, EVH_ERR_INVALID, std::string(who) + ": bad feature type list");
  bool want_orb = false, want_sift = false, want_surf = false;
  for (int i = 0; i < ntypes; i++) {
    // the concatenation buffer holds one segment per detector (kcap + sift_cap + surf_cap rows): a type named twice would
    // overflow it, so it is refused (the reference would simply match the same key points twice and deduplicate them)
    bool* seen = types[i] == EVH_FEATURE_ORB ? &want_orb : types[i] == EVH_FEATURE_SIFT ? &want_sift :
                 types[i] == EVH_FEATURE_SURF ? &want_surf : nullptr;
    if (!seen) return evh_fail(c, EVH_ERR_INVALID, "unknown feature type");
    if (*seen) return evh_fail(c, EVH_ERR_INVALID, std::string(who) + ": a feature type appears twice in the list");
    *seen = true;
  }
  if (want_sift && !c->sift_cap) return evh_fail(c, EVH_ERR_INVALID, std::string(who) + ": SIFT in the list needs evh_sift_enable");
  if (want_surf && !c->surf_cap) return evh_fail(c, EVH_ERR_INVALID, std::string(who) + ": SURF in the list needs evh_surf_enable");
  if (c->mt.cap && c->mt.cap < c->kcap + c->sift_cap + c->surf_cap)
    return evh_fail(c, EVH_ERR_INVALID, std::string(who) + ": enable SIFT and SURF before the first multi-type call");
  int rc = ensure_multitype(c);
  if (rc) return rc;
  if ((rc = join_solve(c))) return rc;
  const int share_group = c->fast_share ? (stream_mode ? nframes : 2) : 0;
  c->fast_share_group = 0;
  if ((rc = ingest_level0(c, who, d_frames, nframes, sw, sh, w, h, channels, row_stride, frame_stride, nfeatures))) return rc;
  if (want_sift && (rc = evh_launch_sift(c, nframes, w, h))) return rc;       // reads level 0 before ORB's kernels run on it
  if (want_surf && (rc = evh_launch_surf(c, nframes, w, h, 400.f))) return rc; // SURF_create(extended=1, hessianThreshold=400)
  if (want_orb && (rc = orb_stages(c, nframes, share_group))) return rc;
  const int q0 = 1, qstep = stream_mode ? 1 : 2, t0 = 0, tstep = stream_mode ? 1 : 2;
  EvhRansacArgs R = mt_ransac_args(c, thr, max_iters, conf, force_max);
  for (int i = 0; i < ntypes; i++) {
    if ((rc = match_pairs_type(c, types[i], npairs, q0, qstep, t0, tstep))) return rc;
    { EvhProfScope ps(c, EVH_ST_RANSAC_STATIC); rc = evh_launch_ransac_static(c, R, npairs); }
    if (rc) return rc;
    EvhAccArgs A{};
    A.rows = c->mt.pts2; A.nrows = c->mt.npts2; A.status = c->mt.pstatus; A.row_stride = c->mt.cap;
    A.acc = c->d_acc; A.nacc = c->d_nacc; A.accstatus = c->d_accstatus; A.acc_stride = c->mt.cap; A.first = i == 0;
    if ((rc = evh_launch_accumulate(c, A, npairs))) return rc;
  }
  EvhMergeArgs M{};
  M.acc = c->d_acc; M.nacc = c->d_nacc; M.accstatus = c->d_accstatus; M.acc_stride = c->mt.cap;
  M.out = c->mt.pts2; M.nout = c->mt.npts2; M.status = c->mt.pstatus; M.out_stride = c->mt.cap;
  if ((rc = evh_launch_merge(c, M, npairs))) return rc;
  R.H = d_H; R.out_status = d_status;
  if (d_state_in) { R.Hsup0 = d_state_in; R.Hprev0 = d_state_in + 9; }
  R.state_out = d_state_out;
  EvhProfScope ps(c, EVH_ST_RANSAC_FINAL);
  return evh_launch_ransac_final(c, R, npairs, stream_mode ? 1 : 0, npairs);
}

}  // namespace

extern "C" {

int evh_version(void) { return 100; }

const char* evh_last_error_string(const evh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int evh_create(int device, int max_w, int max_h, int max_features, int max_frames, void* stream, evh_ctx** out) {
  if (!out) return evh_fail(nullptr, EVH_ERR_INVALID, "evh_create: out is NULL");
  *out = nullptr;
  // frames are a grid dimension of every per-frame kernel (rounded up to a multiple of 8 by the XCD-ordered ones)
  if (max_w < 64 || max_h < 64 || max_w >= 4096 || max_h >= 4096 || max_features < 1 || max_frames < 2 || max_frames > 65528)
    return evh_fail(nullptr, EVH_ERR_INVALID, "evh_create: sizes out of range (64 <= w,h < 4096, 2 <= frames <= 65528)");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return evh_fail(nullptr, EVH_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  evh_ctx* c = new evh_ctx();
  c->device = device; c->max_w = max_w; c->max_h = max_h; c->max_features = max_features; c->max_frames = max_frames;
  c->kcap = kcap_for(max_features);
  // k_filter keeps five int lists of kcap entries in LDS; k_select 8 * (EVH_K1CAP + kcap) bytes of dynamic LDS next to
  // ~5 KB of static arrays (both opt in to more than the default 64 KB at launch)
  if ((size_t)c->kcap * 5 * sizeof(int) > 150 * 1024 || 8 * ((size_t)EVH_K1CAP + c->kcap) + 8 * 1024 > 160 * 1024) {
    delete c;
    return evh_fail(nullptr, EVH_ERR_CAPACITY, "evh_create: max_features too large for the LDS lists of the matching filter / key-point selection (<= EVH_MAX_FEATURES = 5984)");
  }
  int rc = EVH_SUCCESS;
  auto fail = [&](int code) { g_create_error = c->err; evh_destroy(c); return code; };
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { c->err = std::string("hipStreamCreate: ") + hipGetErrorString(e); return fail(EVH_ERR_HIP); }
    c->own_stream = true;
  }
  EvhGeom gmax;
  compute_geometry(max_w, max_h, max_features, gmax);
  const size_t F = (size_t)max_frames, K = (size_t)c->kcap;
  int tabn = 0;
  for (int l = 1; l < EVH_NLEVELS; l++) tabn += 2 * gmax.lv[l].w + 2 * gmax.lv[l].h;
#define A_(call) if ((rc = (call)) != EVH_SUCCESS) return fail(rc)
  A_(dalloc(c, &c->d_pyr, F * (size_t)gmax.pyr_frame_bytes + 256));
  A_(dalloc(c, &c->d_cand, F * (size_t)gmax.cand_frame_entries));
  A_(dalloc(c, &c->d_cand_count, F * EVH_NLEVELS));
  A_(dalloc(c, &c->d_tabs, (size_t)tabn + 64));
  if (hipHostMalloc(reinterpret_cast<void**>(&c->h_tabs), ((size_t)tabn + 64) * sizeof(int), hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_tabs, hipEventDisableTiming) != hipSuccess) {
    c->err = "evh_create: pinned table staging (hipHostMalloc) or its event could not be created";
    return fail(EVH_ERR_HIP);
  }
  A_(dalloc(c, &c->d_kp_xy, F * K * 2));
  A_(dalloc(c, &c->d_kp_meta, F * K));
  A_(dalloc(c, &c->d_kp_resp, F * K));
  A_(dalloc(c, &c->d_kp_angle, F * K));
  A_(dalloc(c, &c->d_desc, F * K * 32));
  A_(dalloc(c, &c->d_kp_count, F));
  A_(dalloc(c, &c->d_frame_flags, F));
  A_(dalloc(c, &c->d_tmp_meta, F * EVH_NLEVELS * K));   // one frame-slot-sized segment per level
  A_(dalloc(c, &c->d_tmp_resp, F * EVH_NLEVELS * K));
  A_(dalloc(c, &c->d_lvl_count, F * EVH_NLEVELS));
  A_(dalloc(c, &c->d_fast_thr, F * EVH_NLEVELS));
  A_(dalloc(c, &c->d_fast_hist, F * EVH_NLEVELS * 256));
  A_(dalloc(c, &c->d_fast_redo, F * EVH_NLEVELS + 1));
  {
    int64_t mw = 0;
    for (int l = 0; l < EVH_NLEVELS; l++) mw += (int64_t)((gmax.lv[l].w + 31) / 32) * gmax.lv[l].h;
    c->cv_mask_frame_words = mw + 64;
  }
  A_(dalloc(c, &c->d_cv_seq, F * (size_t)gmax.cand_frame_entries));
  A_(dalloc(c, &c->d_cv_seq32, F * (size_t)gmax.cand_frame_entries));
  A_(dalloc(c, &c->d_cv_lpos, F * (size_t)gmax.cand_frame_entries));
  A_(dalloc(c, &c->d_cv_rpos, F * (size_t)gmax.cand_frame_entries));
  A_(dalloc(c, &c->d_cv_mask, F * 2 * (size_t)c->cv_mask_frame_words));
  A_(dalloc(c, &c->d_cv_tdesc, F * 8 * (size_t)gmax.total_tiles + 64));
  A_(dalloc(c, &c->d_fast_hint, 16 + 8 * 256 + 8));
  if (hipMemset(c->d_fast_hint, 0, sizeof(int) * (16 + 8 * 256 + 8)) != hipSuccess) {
    c->err = "hipMemset(d_fast_hint) failed";
    return fail(EVH_ERR_HIP);
  }
  A_(dalloc(c, &c->d_knn_idx, F * K * 2));
  A_(dalloc(c, &c->d_knn_d2, F * K * 2));
  A_(dalloc(c, &c->d_pts, F * K * 4));
  A_(dalloc(c, &c->d_pts2, F * K * 4));
  A_(dalloc(c, &c->d_crow, F * K * 4));
  A_(dalloc(c, &c->d_npts, F));
  A_(dalloc(c, &c->d_npts2, F));
  A_(dalloc(c, &c->d_pstatus, F));
  A_(dalloc(c, &c->d_H1, F * 9));
  A_(dalloc(c, &c->d_mask, F * K));
  A_(dalloc(c, &c->d_lm, F * K * 4));
  A_(dalloc(c, &c->d_info, F * 8));
  A_(dalloc(c, &c->d_small, 64));
#undef A_
  e = hipMemset(c->d_kp_count, 0, F * sizeof(int));
  if (e == hipSuccess) e = hipMemset(c->d_frame_flags, 0, F * sizeof(int));
  if (e != hipSuccess) { c->err = std::string("hipMemset: ") + hipGetErrorString(e); return fail(EVH_ERR_HIP); }
  *out = c;
  return EVH_SUCCESS;
}

void evh_destroy(evh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->d_pyr, c->d_cand, c->d_cand_count, c->d_tabs, c->d_kp_xy, c->d_kp_meta, c->d_kp_resp, c->d_kp_angle,
                  c->d_desc, c->d_kp_count, c->d_frame_flags, c->d_tmp_meta, c->d_tmp_resp, c->d_lvl_count, c->d_fast_thr, c->d_fast_hist, c->d_fast_redo, c->d_cv_seq, c->d_cv_seq32, c->d_cv_lpos, c->d_cv_rpos, c->d_cv_mask, c->d_cv_tdesc, c->d_area_tab, c->d_lane_v, c->d_fast_hint, c->d_knn_idx, c->d_knn_d2, c->d_pts, c->d_pts2, c->d_crow,
                  c->d_npts, c->d_npts2, c->d_pstatus, c->d_H1, c->d_mask, c->d_lm, c->d_info, c->d_small, c->d_scratch, c->d_scan_ws, c->d_filter_ws, c->d_merge_ws};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  evh_sift_free(c);
  evh_surf_free(c);
  {
    void* mp[] = {c->mt.knn_idx, c->mt.knn_d2, c->mt.pts, c->mt.pts2, c->mt.crow, c->mt.npts, c->mt.npts2, c->mt.pstatus, c->mt.H1,
                  c->mt.mask, c->mt.lm, c->mt.info, c->d_acc, c->d_nacc, c->d_accstatus};
    for (void* p : mp) if (p) (void)hipFree(p);
  }
  for (auto& s : c->prof_spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (auto e : c->prof_pool) (void)hipEventDestroy(e);
  if (c->solve_stream) { (void)hipStreamSynchronize(c->solve_stream); (void)hipStreamDestroy(c->solve_stream); }
  if (c->h_tabs) (void)hipHostFree(c->h_tabs);
  if (c->ev_tabs) (void)hipEventDestroy(c->ev_tabs);
  if (c->ev_match_done) (void)hipEventDestroy(c->ev_match_done);
  if (c->ev_solve_done) (void)hipEventDestroy(c->ev_solve_done);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int evh_set_fast_lift(evh_ctx* c, int on) {
  if (!c) return EVH_ERR_INVALID;
  c->fast_lift = on != 0;
  return EVH_SUCCESS;
}

int evh_set_solver_mode(evh_ctx* c, int mode) {
  if (!c || (mode != EVH_SOLVER_EXACT && mode != EVH_SOLVER_FAST)) return EVH_ERR_INVALID;
  c->solver_mode = mode;
  return EVH_SUCCESS;
}

int evh_get_solver_mode(const evh_ctx* c) { return c ? c->solver_mode : EVH_ERR_INVALID; }

int evh_set_keypoint_order(evh_ctx* c, int mode) {
  if (!c || (mode != EVH_ORDER_CANONICAL && mode != EVH_ORDER_OPENCV)) return EVH_ERR_INVALID;
  c->order_mode = mode;
  return EVH_SUCCESS;
}

int evh_get_keypoint_order(const evh_ctx* c) { return c ? c->order_mode : EVH_ERR_INVALID; }

int evh_set_fast_hint(evh_ctx* c, int on) {
  if (!c) return EVH_ERR_INVALID;
  c->fast_hint = on != 0;
  return EVH_SUCCESS;
}

int evh_set_fast_share(evh_ctx* c, int on) {
  if (!c) return EVH_ERR_INVALID;
  c->fast_share = on != 0;
  return EVH_SUCCESS;
}

int evh_profile_enable(evh_ctx* c, int on) {
  if (!c) return EVH_ERR_INVALID;
  c->profiling = on != 0;
  return EVH_SUCCESS;
}

const char* evh_profile_stage_name(int stage) {
  static const char* names[EVH_NSTAGES] = {"gray", "pyramid", "fast", "select", "describe", "knn2", "filter",
                                           "ransac_static", "ransac_final"};
  return stage >= 0 && stage < EVH_NSTAGES ? names[stage] : "";
}

int evh_profile_read(evh_ctx* c, float* h_total_ms, int* h_counts) {
  if (!c || !h_total_ms || !h_counts) return evh_fail(c, EVH_ERR_INVALID, "evh_profile_read: bad argument");
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  if (c->solve_stream) EVH_HIP(c, hipStreamSynchronize(c->solve_stream));
  for (int i = 0; i < EVH_NSTAGES; i++) { h_total_ms[i] = 0.f; h_counts[i] = 0; }
  for (auto& s : c->prof_spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { h_total_ms[s.stage] += ms; h_counts[s.stage]++; }
    c->prof_pool.push_back(s.a); c->prof_pool.push_back(s.b);
  }
  c->prof_spans.clear();
  return EVH_SUCCESS;
}

void* evh_stream(const evh_ctx* c) { return c ? (void*)c->stream : nullptr; }

int evh_synchronize(evh_ctx* c) {
  if (!c) return EVH_ERR_INVALID;
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  if (c->solve_stream) EVH_HIP(c, hipStreamSynchronize(c->solve_stream));
  c->solve_pending = false;
  return EVH_SUCCESS;
}

int evh_set_async_solve(evh_ctx* c, int on) {
  if (!c) return EVH_ERR_INVALID;
  if (on && !c->solve_stream) {
    EVH_HIP(c, hipStreamCreateWithFlags(&c->solve_stream, hipStreamNonBlocking));
    EVH_HIP(c, hipEventCreateWithFlags(&c->ev_match_done, hipEventDisableTiming));
    EVH_HIP(c, hipEventCreateWithFlags(&c->ev_solve_done, hipEventDisableTiming));
  }
  if (!on && c->solve_stream) { EVH_HIP(c, hipStreamSynchronize(c->solve_stream)); c->solve_pending = false; }
  c->async_solve = on != 0;
  return EVH_SUCCESS;
}

int evh_solve_wait(evh_ctx* c, void* stream) {
  if (!c) return EVH_ERR_INVALID;
  if (c->solve_pending) EVH_HIP(c, hipStreamWaitEvent(stream ? (hipStream_t)stream : c->stream, c->ev_solve_done, 0));
  return EVH_SUCCESS;
}

int evh_resize_area_u8(evh_ctx* c, const uint8_t* d_src, int nimg, int sw, int sh, int cn, int64_t src_stride,
                       int64_t src_img_stride, uint8_t* d_dst, int dw, int dh, int64_t dst_stride,
                       int64_t dst_img_stride) {
  if (!c || !d_src || !d_dst || nimg < 1 || sw < 1 || sh < 1 || dw < 1 || dh < 1 || (cn != 1 && cn != 3))
    return evh_fail(c, EVH_ERR_INVALID, "evh_resize_area_u8: bad argument");
  if (nimg > 65535 || dh > 65535) return evh_fail(c, EVH_ERR_CAPACITY, "evh_resize_area_u8: too many images/rows");
  return evh_launch_resize_area(c, d_src, nimg, sw, sh, cn, src_stride, src_img_stride, d_dst, dw, dh, dst_stride,
                                dst_img_stride);
}

int evh_resize_area_u8c3(evh_ctx* c, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh) {
  return evh_resize_area_u8(c, d_src, 1, sw, sh, 3, (int64_t)sw * 3, (int64_t)sw * sh * 3, d_dst, dw, dh, (int64_t)dw * 3,
                            (int64_t)dw * dh * 3);
}

int evh_fixed_plane_field(evh_ctx* c, const double* h_Hsup, int n, int w, int h, double* d_field, double* h_max) {
  if (!c || !h_Hsup || !h_max || n < 1 || w < 1 || h < 1) return evh_fail(c, EVH_ERR_INVALID, "evh_fixed_plane_field: bad argument");
  if (n > 65535) return evh_fail(c, EVH_ERR_CAPACITY, "evh_fixed_plane_field: at most 65535 matrices per call");
  { int sr = ensure_scratch(c, (sizeof(double) * 9 + sizeof(unsigned long long)) * (size_t)n); if (sr) return sr; }
  double* d_H = reinterpret_cast<double*>(c->d_scratch);
  unsigned long long* d_max = reinterpret_cast<unsigned long long*>(c->d_scratch + sizeof(double) * 9 * (size_t)n);
  std::vector<unsigned long long> keys(n);
  int rc = EVH_SUCCESS;
  hipError_t e = hipMemcpyAsync(d_H, h_Hsup, sizeof(double) * 9 * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) rc = evh_launch_fixed_plane(c, d_H, n, w, h, d_field, d_max);
  if (e == hipSuccess && rc == EVH_SUCCESS)
    e = hipMemcpyAsync(keys.data(), d_max, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return evh_fail(c, EVH_ERR_HIP, std::string("evh_fixed_plane_field: ") + hipGetErrorString(e));
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    unsigned long long b = keys[i];
    b = (b >> 63) ? (b & 0x7FFFFFFFFFFFFFFFull) : ~b;
    memcpy(&h_max[i], &b, sizeof(double));
  }
  return EVH_SUCCESS;
}

int evh_superposition_scan(evh_ctx* c, const double* h_H, int n, double* h_out) {
  if (!c || !h_H || !h_out || n < 1) return evh_fail(c, EVH_ERR_INVALID, "evh_superposition_scan: bad argument");
  const size_t bytes = sizeof(double) * 9 * (size_t)n;
  { int sr = ensure_scratch(c, 2 * bytes); if (sr) return sr; }
  double* d = reinterpret_cast<double*>(c->d_scratch);
  int rc = EVH_SUCCESS;
  hipError_t e = hipMemcpyAsync(d, h_H, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) rc = evh_launch_superposition_scan(c, d, n, d + 9 * (size_t)n);
  if (e == hipSuccess && rc == EVH_SUCCESS) e = hipMemcpyAsync(h_out, d + 9 * (size_t)n, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return evh_fail(c, EVH_ERR_HIP, std::string("evh_superposition_scan: ") + hipGetErrorString(e));
  return rc;
}

int evh_transform_points(evh_ctx* c, const double* h_M, int nmat, const int32_t* h_idx, const double* h_pts, int n, double kx,
                         double ky, int decimals, double* h_out) {
  if (!c || !h_M || !h_idx || !h_pts || !h_out || nmat < 1 || n < 0 || decimals > 15)
    return evh_fail(c, EVH_ERR_INVALID, "evh_transform_points: bad argument");
  if (n == 0) return EVH_SUCCESS;
  for (int i = 0; i < n; i++)
    if (h_idx[i] < 0 || h_idx[i] >= nmat) return evh_fail(c, EVH_ERR_INVALID, "evh_transform_points: matrix index out of range");
  const size_t bm = sizeof(double) * 9 * (size_t)nmat, bp = sizeof(double) * 2 * (size_t)n, bi = sizeof(int32_t) * (size_t)n;
  { int sr = ensure_scratch(c, bm + 2 * bp + bi); if (sr) return sr; }
  char* d = c->d_scratch;
  double* d_M = reinterpret_cast<double*>(d);
  double* d_pts = reinterpret_cast<double*>(d + bm);
  double* d_out = reinterpret_cast<double*>(d + bm + bp);
  int* d_idx = reinterpret_cast<int*>(d + bm + 2 * bp);
  int rc = EVH_SUCCESS;
  hipError_t e = hipMemcpyAsync(d_M, h_M, bm, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pts, h_pts, bp, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, h_idx, bi, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) rc = evh_launch_transform_points(c, d_M, d_idx, d_pts, n, kx, ky, decimals, d_out);
  if (e == hipSuccess && rc == EVH_SUCCESS) e = hipMemcpyAsync(h_out, d_out, bp, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return evh_fail(c, EVH_ERR_HIP, std::string("evh_transform_points: ") + hipGetErrorString(e));
  return rc;
}

int evh_orb_detect_batch(evh_ctx* c, const uint8_t* d_frames, int nframes, int w, int h, int channels,
                         int64_t row_stride, int64_t frame_stride, int nfeatures) {
  return detect_batch(c, d_frames, nframes, w, h, w, h, channels, row_stride, frame_stride, nfeatures);
}

int evh_orb_detect_batch_resized(evh_ctx* c, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                                 int64_t row_stride, int64_t frame_stride, int w, int h, int nfeatures) {
  return detect_batch(c, d_frames, nframes, src_w, src_h, w, h, channels, row_stride, frame_stride, nfeatures);
}

int evh_orb_capacity(const evh_ctx* c) { return c ? c->kcap : EVH_ERR_INVALID; }

int evh_orb_count(evh_ctx* c, int frame) {
  if (!c || frame < 0 || frame >= c->nframes_resident) return evh_fail(c, EVH_ERR_INVALID, "bad frame slot");
  int n = 0, fl = 0;
  EVH_HIP(c, hipMemcpyAsync(&n, c->d_kp_count + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipMemcpyAsync(&fl, c->d_frame_flags + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  if (fl & 2) return evh_fail(c, EVH_ERR_CAPACITY, "key-point selection: nth_element's depth limit was reached for this frame (heap-select fall-back)");
  if (fl) return evh_fail(c, EVH_ERR_CAPACITY, "a fixed-capacity keypoint list overflowed for this frame");
  return n;
}

int evh_orb_download(evh_ctx* c, int frame, float* h_xy, uint8_t* h_desc, int32_t* h_octave, int32_t* h_lxy,
                     float* h_response, float* h_angle) {
  int n = evh_orb_count(c, frame);
  if (n <= 0) return n;
  const size_t o = (size_t)frame * c->kcap;
  std::vector<uint32_t> meta;
  if (h_xy) EVH_HIP(c, hipMemcpyAsync(h_xy, c->d_kp_xy + 2 * o, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, c->stream));
  if (h_desc) EVH_HIP(c, hipMemcpyAsync(h_desc, c->d_desc + 32 * o, 32 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  if (h_response) EVH_HIP(c, hipMemcpyAsync(h_response, c->d_kp_resp + o, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
  if (h_angle) EVH_HIP(c, hipMemcpyAsync(h_angle, c->d_kp_angle + o, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
  if (h_octave || h_lxy) {
    meta.resize(n);
    EVH_HIP(c, hipMemcpyAsync(meta.data(), c->d_kp_meta + o, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
  }
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n && !meta.empty(); i++) {
    if (h_octave) h_octave[i] = (int32_t)(meta[i] >> 24);
    if (h_lxy) { h_lxy[2 * i] = (int32_t)(meta[i] & 0xFFFu); h_lxy[2 * i + 1] = (int32_t)((meta[i] >> 12) & 0xFFFu); }
  }
  return n;
}

int evh_orb_detect_compute(evh_ctx* c, const uint8_t* d_frame, int w, int h, int channels, int nfeatures, float* h_xy,
                           uint8_t* h_desc, int32_t* h_octave, int* h_count) {
  if (!c) return EVH_ERR_INVALID;
  int rc = evh_orb_detect_batch(c, d_frame, 1, w, h, channels, (int64_t)w * channels, (int64_t)w * h * channels, nfeatures);
  if (rc != EVH_SUCCESS) return rc;
  const int n = evh_orb_download(c, 0, h_xy, h_desc, h_octave, nullptr, nullptr, nullptr);
  if (n < 0) return n;
  if (h_count) *h_count = n;
  return EVH_SUCCESS;
}

int evh_orb_level_info(const evh_ctx* c, int level, int* w, int* h, int* quota, float* scale) {
  if (!c || !c->geom_valid || level < 0 || level >= EVH_NLEVELS) return EVH_ERR_INVALID;
  const EvhLevel& L = c->g.lv[level];
  if (w) *w = L.w; if (h) *h = L.h; if (quota) *quota = L.quota; if (scale) *scale = L.scale;
  return EVH_SUCCESS;
}

int evh_orb_download_level(evh_ctx* c, int frame, int level, uint8_t* h_pixels) {
  if (!c || !c->geom_valid || level < 0 || level >= EVH_NLEVELS || frame < 0 || frame >= c->nframes_resident || !h_pixels)
    return evh_fail(c, EVH_ERR_INVALID, "evh_orb_download_level: bad argument");
  const EvhLevel& L = c->g.lv[level];
  EVH_HIP(c, hipMemcpy2DAsync(h_pixels, L.w, c->d_pyr + (size_t)frame * c->g.pyr_frame_bytes + L.off, L.stride, L.w, L.h,
                              hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  return EVH_SUCCESS;
}

int evh_orb_download_candidates(evh_ctx* c, int frame, int level, uint32_t* h_packed, int cap) {
  if (!c || !c->geom_valid || level < 0 || level >= EVH_NLEVELS || frame < 0 || frame >= c->nframes_resident)
    return evh_fail(c, EVH_ERR_INVALID, "evh_orb_download_candidates: bad argument");
  const EvhLevel& L = c->g.lv[level];
  int n = 0;
  EVH_HIP(c, hipMemcpyAsync(&n, c->d_cand_count + frame * EVH_NLEVELS + level, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  int m = std::min(std::min(n, cap), L.cand_cap);
  if (m > 0 && h_packed) {
    EVH_HIP(c, hipMemcpyAsync(h_packed, c->d_cand + (size_t)frame * c->g.cand_frame_entries + L.cand_off,
                              sizeof(uint32_t) * m, hipMemcpyDeviceToHost, c->stream));
    EVH_HIP(c, hipStreamSynchronize(c->stream));
  }
  return n;
}

static int knn_generic(evh_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx, uint32_t* d_d2,
                       int hamming, int desc_bytes = 32) {
  if (!c || !d_idx || !d_d2 || nq < 0 || nt < 0) return evh_fail(c, EVH_ERR_INVALID, "evh_match_knn2: bad argument");
  if (nq == 0) return EVH_SUCCESS;
  if (((uintptr_t)d_q | (uintptr_t)d_t) & 15) return evh_fail(c, EVH_ERR_INVALID, "descriptor buffers must be 16-byte aligned");
  EvhKnnArgs K{};
  K.q = d_q; K.t = d_t; K.slot_bytes = 0; K.nq_fixed = nq; K.nt_fixed = nt;
  K.idx = d_idx; K.d2 = d_d2; K.out_stride = nq; K.hamming = hamming; K.desc_bytes = desc_bytes;
  return evh_launch_knn2(c, K, 1);
}

int evh_match_knn2_l2u8(evh_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx, uint32_t* d_d2) {
  return knn_generic(c, d_q, nq, d_t, nt, d_idx, d_d2, 0);
}
int evh_match_knn2_l2u8x128(evh_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx, uint32_t* d_d2) {
  return knn_generic(c, d_q, nq, d_t, nt, d_idx, d_d2, 0, 128);
}
int evh_match_knn2_hamming(evh_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx,
                           uint32_t* d_d2) {
  return knn_generic(c, d_q, nq, d_t, nt, d_idx, d_d2, 1);
}

int evh_ratio_unique_filter(evh_ctx* c, const int32_t* d_idx, const uint32_t* d_d2, int nq, int nt, const float* d_xy_q,
                            const float* d_xy_t, double ratio, int min_matches, float* d_pts, int* h_count, int* h_status) {
  if (!c || !d_idx || !d_d2 || !d_xy_q || !d_xy_t || !d_pts || !h_count || !h_status || nq < 0 || nt < 0)
    return evh_fail(c, EVH_ERR_INVALID, "evh_ratio_unique_filter: bad argument");
  const int kc = std::max(std::max(nq, nt), 1);
  if (kc > 65535) return evh_fail(c, EVH_ERR_CAPACITY, "evh_ratio_unique_filter: too many rows");
  if (((uintptr_t)d_pts) & 15) return evh_fail(c, EVH_ERR_INVALID, "d_pts must be 16-byte aligned");
  int* d_cnt = reinterpret_cast<int*>(c->d_small);
  EvhFilterArgs F{};
  F.idx = d_idx; F.d2 = d_d2; F.knn_stride = nq; F.xy_q = d_xy_q; F.xy_t = d_xy_t; F.xy_slot_floats = 0;
  F.nq_fixed = nq; F.nt_fixed = nt; F.ratio = ratio; F.min_matches = min_matches;
  F.pts = d_pts; F.pts_stride = nq; F.npts = d_cnt; F.status = d_cnt + 1; F.kcap = kc;
  int rc = evh_launch_filter(c, F, 1);
  if (rc) return rc;
  int host[2];
  EVH_HIP(c, hipMemcpyAsync(host, d_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  *h_count = host[0]; *h_status = host[1];
  return EVH_SUCCESS;
}

static int find_homography_entry(evh_ctx* c, const float* d_pts, int n, double thr, int max_iters, double conf, int force_max,
                                 double* h_H, uint8_t* h_mask, int* h_found, int* h_info) {
  if (!c || (!d_pts && n > 0) || n < 0 || !h_H || !h_found) return evh_fail(c, EVH_ERR_INVALID, "evh_find_homography_ransac: bad argument");
  if (n > c->kcap * c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "evh_find_homography_ransac: too many rows");
  if (((uintptr_t)d_pts) & 15) return evh_fail(c, EVH_ERR_INVALID, "d_pts must be 16-byte aligned");
  { int jr = join_solve(c); if (jr) return jr; }
  // scratch: the per-pair buffers viewed as one big problem
  EvhRansacArgs R{};
  R.fast_solver = c->solver_mode;
  R.pts = const_cast<float*>(d_pts); R.n_fixed = n; R.thr = thr; R.max_iters = max_iters; R.conf = conf; R.force_max = force_max;
  if (force_max) { int lr = ensure_lane_scratch(c); if (lr) return lr; R.lane_v = c->d_lane_v; }
  R.mask = c->d_mask; R.crow = c->d_crow; R.lm = c->d_lm;
  R.H = c->d_small; R.found = reinterpret_cast<int*>(c->d_small + 16); R.info = reinterpret_cast<int*>(c->d_small + 17);
  int rc = evh_launch_find_homography(c, R);
  if (rc) return rc;
  double Hh[9]; int found = 0, info[3] = {0, 0, 0};
  EVH_HIP(c, hipMemcpyAsync(Hh, c->d_small, sizeof(Hh), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipMemcpyAsync(&found, c->d_small + 16, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipMemcpyAsync(info, c->d_small + 17, sizeof(info), hipMemcpyDeviceToHost, c->stream));
  if (h_mask && n > 0) EVH_HIP(c, hipMemcpyAsync(h_mask, c->d_mask, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(h_H, Hh, sizeof(Hh));
  *h_found = found;
  if (h_info) memcpy(h_info, info, sizeof(info));
  return EVH_SUCCESS;
}

int evh_find_homography_ransac(evh_ctx* c, const float* d_pts, int n, double thr, int max_iters, double conf, double* h_H,
                               uint8_t* h_mask, int* h_found, int* h_info) {
  return find_homography_entry(c, d_pts, n, thr, max_iters, conf, 0, h_H, h_mask, h_found, h_info);
}
int evh_find_homography_ransac_fixed(evh_ctx* c, const float* d_pts, int n, double thr, int max_iters, double conf,
                                     double* h_H, uint8_t* h_mask, int* h_found, int* h_info) {
  return find_homography_entry(c, d_pts, n, thr, max_iters, conf, 1, h_H, h_mask, h_found, h_info);
}

int evh_static_filter(evh_ctx* c, const double* h_H, const float* d_pts, int n, float* d_out_pts, int* h_count) {
  if (!c || !h_H || (!d_pts && n > 0) || !d_out_pts || !h_count || n < 0) return evh_fail(c, EVH_ERR_INVALID, "evh_static_filter: bad argument");
  if (n > c->kcap * c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "evh_static_filter: too many rows");
  if ((((uintptr_t)d_pts) | ((uintptr_t)d_out_pts)) & 15) return evh_fail(c, EVH_ERR_INVALID, "row buffers must be 16-byte aligned");
  { int jr = join_solve(c); if (jr) return jr; }
  EVH_HIP(c, hipMemcpyAsync(c->d_small, h_H, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int* d_cnt = reinterpret_cast<int*>(c->d_small + 16);
  int rc = evh_launch_static_filter(c, c->d_small, d_pts, n, reinterpret_cast<int*>(c->d_lm), d_out_pts, d_cnt);
  if (rc) return rc;
  EVH_HIP(c, hipMemcpyAsync(h_count, d_cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  return EVH_SUCCESS;
}

int evh_pair_homography_batch(evh_ctx* c, const uint8_t* d_frames, int npairs, int mode, int w, int h, int channels,
                              int64_t row_stride, int64_t frame_stride, int nfeatures, double ransac_thr,
                              int ransac_max_iters, double ransac_conf, int force_max_iters, double* d_H, int32_t* d_status) {
  if (!c || !d_frames || !d_H || !d_status || npairs < 1) return evh_fail(c, EVH_ERR_INVALID, "evh_pair_homography_batch: bad argument");
  if (mode != EVH_MODE_INDEPENDENT_PAIRS && mode != EVH_MODE_STREAM) return evh_fail(c, EVH_ERR_INVALID, "unknown mode");
  const int nframes = mode == EVH_MODE_INDEPENDENT_PAIRS ? 2 * npairs : npairs + 1;
  if (nframes > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "batch needs more frame slots than max_frames");
  c->fast_share_group = mode == EVH_MODE_INDEPENDENT_PAIRS ? 2 : nframes;   // the two frames of a pair / one stream
  int rc = evh_orb_detect_batch(c, d_frames, nframes, w, h, channels, row_stride, frame_stride, nfeatures);
  if (rc) return rc;
  if (mode == EVH_MODE_INDEPENDENT_PAIRS) rc = match_pairs(c, npairs, 1, 2, 0, 2);
  else rc = match_pairs(c, npairs, 1, 1, 0, 1);
  if (rc) return rc;
  EvhRansacArgs R = pair_ransac_args(c, ransac_thr, ransac_max_iters, ransac_conf, force_max_iters);
  R.H = d_H; R.out_status = d_status;
  return solve_pairs(c, R, npairs, mode == EVH_MODE_STREAM ? 1 : 0, npairs, npairs);
}

int evh_stream_homography_batch(evh_ctx* c, const uint8_t* d_frames, int nframes, int w, int h, int channels,
                                int64_t row_stride, int64_t frame_stride, int nfeatures, double ransac_thr,
                                int ransac_max_iters, double ransac_conf, int force_max_iters, const double* d_state_in,
                                double* d_state_out, double* d_H, int32_t* d_status) {
  if (!c || !d_frames || !d_H || !d_status || nframes < 2) return evh_fail(c, EVH_ERR_INVALID, "evh_stream_homography_batch: bad argument");
  if (nframes > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "chunk needs more frame slots than max_frames");
  const int npairs = nframes - 1;
  c->fast_share_group = nframes;
  int rc = evh_orb_detect_batch(c, d_frames, nframes, w, h, channels, row_stride, frame_stride, nfeatures);
  if (rc) return rc;
  if ((rc = match_pairs(c, npairs, 1, 1, 0, 1))) return rc;
  EvhRansacArgs R = pair_ransac_args(c, ransac_thr, ransac_max_iters, ransac_conf, force_max_iters);
  R.H = d_H; R.out_status = d_status;
  if (d_state_in) { R.Hsup0 = d_state_in; R.Hprev0 = d_state_in + 9; }
  R.state_out = d_state_out;
  return solve_pairs(c, R, npairs, 1, npairs, npairs);
}

int evh_stream_homography_batch_resized(evh_ctx* c, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                                        int64_t row_stride, int64_t frame_stride, int w, int h, int nfeatures,
                                        double ransac_thr, int ransac_max_iters, double ransac_conf, int force_max_iters,
                                        const double* d_state_in, double* d_state_out, double* d_H, int32_t* d_status) {
  if (!c || !d_frames || !d_H || !d_status || nframes < 2) return evh_fail(c, EVH_ERR_INVALID, "evh_stream_homography_batch_resized: bad argument");
  if (nframes > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "chunk needs more frame slots than max_frames");
  const int npairs = nframes - 1;
  c->fast_share_group = nframes;
  int rc = detect_batch(c, d_frames, nframes, src_w, src_h, w, h, channels, row_stride, frame_stride, nfeatures);
  if (rc) return rc;
  if ((rc = match_pairs(c, npairs, 1, 1, 0, 1))) return rc;
  EvhRansacArgs R = pair_ransac_args(c, ransac_thr, ransac_max_iters, ransac_conf, force_max_iters);
  R.H = d_H; R.out_status = d_status;
  if (d_state_in) { R.Hsup0 = d_state_in; R.Hprev0 = d_state_in + 9; }
  R.state_out = d_state_out;
  return solve_pairs(c, R, npairs, 1, npairs, npairs);
}

int evh_match_static_from_slots(evh_ctx* c, int cur_slot, int prev_slot, float* h_pts, int cap, int* h_count, int* h_status) {
  if (!c || !h_count || !h_status || cur_slot < 0 || prev_slot < 0 || cur_slot >= c->nframes_resident ||
      prev_slot >= c->nframes_resident)
    return evh_fail(c, EVH_ERR_INVALID, "evh_match_static_from_slots: bad argument");
  { int jr = join_solve(c); if (jr) return jr; }
  int rc = match_pairs(c, 1, cur_slot, 0, prev_slot, 0);
  if (rc) return rc;
  EvhRansacArgs R = pair_ransac_args(c, 3.0, 2000, 0.995, 0);  // constants.py:22 THRESHOLD_FOR_FIND_HOMOGRAPHY
  if ((rc = evh_launch_ransac_static(c, R, 1))) return rc;
  int st = 0, n = 0;
  EVH_HIP(c, hipMemcpyAsync(&st, c->d_pstatus, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipMemcpyAsync(&n, c->d_npts2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  *h_status = st; *h_count = n;
  if (st == EVH_PAIR_OK && n > 0 && h_pts) {
    if (n > cap) return evh_fail(c, EVH_ERR_CAPACITY, "h_pts too small");
    EVH_HIP(c, hipMemcpy(h_pts, c->d_pts2, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost));
  }
  return EVH_SUCCESS;
}

int evh_compute_homography(evh_ctx* c, const float* h_pts, int n, const double* h_Hsup, double* h_H, int* h_status) {
  if (!c || (!h_pts && n > 0) || !h_H || !h_status || n < 0) return evh_fail(c, EVH_ERR_INVALID, "evh_compute_homography: bad argument");
  if (n > c->kcap) return evh_fail(c, EVH_ERR_CAPACITY, "evh_compute_homography: too many rows");
  { int jr = join_solve(c); if (jr) return jr; }
  const int zero = 0;
  EVH_HIP(c, hipMemcpyAsync(c->d_pts2, h_pts, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  EVH_HIP(c, hipMemcpyAsync(c->d_npts2, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
  EVH_HIP(c, hipMemcpyAsync(c->d_pstatus, &zero, sizeof(int), hipMemcpyHostToDevice, c->stream));
  EvhRansacArgs R = pair_ransac_args(c, 3.0, 2000, 0.995, 0);
  R.H = c->d_small; R.out_status = reinterpret_cast<int*>(c->d_small + 32);
  if (h_Hsup) {
    EVH_HIP(c, hipMemcpyAsync(c->d_small + 16, h_Hsup, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    R.Hsup0 = c->d_small + 16;
  }
  // the stream kernel with one pair applies the optional pre-transform; Hprev0 = Hsup0 only marks "not first"
  R.Hprev0 = R.Hsup0;
  int rc = evh_launch_ransac_final(c, R, 1, h_Hsup ? 1 : 0, 1);
  if (rc) return rc;
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  EVH_HIP(c, hipMemcpy(h_H, c->d_small, 9 * sizeof(double), hipMemcpyDeviceToHost));
  EVH_HIP(c, hipMemcpy(h_status, c->d_small + 32, sizeof(int), hipMemcpyDeviceToHost));
  return EVH_SUCCESS;
}

int evh_multi_stream_homography_batch(evh_ctx* c, const uint8_t* d_frames, int nstreams, int frames_per_stream, int w,
                                      int h, int channels, int64_t row_stride, int64_t frame_stride, int nfeatures,
                                      double ransac_thr, int ransac_max_iters, double ransac_conf, int force_max_iters,
                                      const double* d_state_in, double* d_state_out, double* d_H, int32_t* d_status) {
  if (!c || !d_frames || !d_H || !d_status || nstreams < 1 || frames_per_stream < 2)
    return evh_fail(c, EVH_ERR_INVALID, "evh_multi_stream_homography_batch: bad argument");
  const int64_t nframes64 = (int64_t)nstreams * frames_per_stream;
  if (nframes64 > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "batch needs more frame slots than max_frames");
  const int nframes = (int)nframes64;
  c->fast_share_group = frames_per_stream;
  int rc = evh_orb_detect_batch(c, d_frames, nframes, w, h, channels, row_stride, frame_stride, nfeatures);
  if (rc) return rc;
  // pair slot p = (frame p + 1, frame p): the slot that straddles two streams is computed and never read
  if ((rc = match_pairs(c, nframes - 1, 1, 1, 0, 1))) return rc;
  EvhRansacArgs R = pair_ransac_args(c, ransac_thr, ransac_max_iters, ransac_conf, force_max_iters);
  R.H = d_H; R.out_status = d_status;
  if (d_state_in) { R.Hsup0 = d_state_in; R.Hprev0 = d_state_in + 9; }
  R.state_out = d_state_out;
  return solve_pairs(c, R, nframes - 1, nstreams, frames_per_stream - 1, frames_per_stream);
}

int evh_stream_static_batch(evh_ctx* c, const uint8_t* d_frames, int nframes, int w, int h, int channels,
                            int64_t row_stride, int64_t frame_stride, int nfeatures, double ransac_thr,
                            int ransac_max_iters, double ransac_conf, int force_max_iters, float* d_rows, int row_cap,
                            int32_t* d_counts, int32_t* d_status1) {
  if (!c || !d_frames || !d_rows || !d_counts || !d_status1 || nframes < 2)
    return evh_fail(c, EVH_ERR_INVALID, "evh_stream_static_batch: bad argument");
  if (nframes > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "block needs more frame slots than max_frames");
  if (row_cap != c->kcap) return evh_fail(c, EVH_ERR_INVALID, "row_cap must equal evh_orb_capacity()");
  const int npairs = nframes - 1;
  c->fast_share_group = nframes;
  int rc = evh_orb_detect_batch(c, d_frames, nframes, w, h, channels, row_stride, frame_stride, nfeatures);
  if (rc) return rc;
  if ((rc = match_pairs(c, npairs, 1, 1, 0, 1))) return rc;      // orders itself behind a pending async solve
  EvhRansacArgs R = pair_ransac_args(c, ransac_thr, ransac_max_iters, ransac_conf, force_max_iters);
  { EvhProfScope ps(c, EVH_ST_RANSAC_STATIC); rc = evh_launch_ransac_static(c, R, npairs); }
  if (rc) return rc;
  EVH_HIP(c, hipMemcpyAsync(d_rows, c->d_pts2, sizeof(float) * 4 * (size_t)c->kcap * npairs, hipMemcpyDeviceToDevice, c->stream));
  EVH_HIP(c, hipMemcpyAsync(d_counts, c->d_npts2, sizeof(int) * (size_t)npairs, hipMemcpyDeviceToDevice, c->stream));
  EVH_HIP(c, hipMemcpyAsync(d_status1, c->d_pstatus, sizeof(int) * (size_t)npairs, hipMemcpyDeviceToDevice, c->stream));
  return EVH_SUCCESS;
}

int evh_stream_scan(evh_ctx* c, const float* d_rows, int row_cap, const int32_t* d_counts, const int32_t* d_status1,
                    int npairs, double ransac_thr, int ransac_max_iters, double ransac_conf, int force_max_iters,
                    const double* d_state_in, double* d_state_out, double* d_H, int32_t* d_status) {
  if (!c || !d_rows || !d_counts || !d_status1 || !d_H || !d_status || npairs < 1)
    return evh_fail(c, EVH_ERR_INVALID, "evh_stream_scan: bad argument");
  if (row_cap < 1 || row_cap > c->kcap) return evh_fail(c, EVH_ERR_CAPACITY, "row_cap larger than evh_orb_capacity()");
  if (((uintptr_t)d_rows) & 15) return evh_fail(c, EVH_ERR_INVALID, "d_rows must be 16-byte aligned");
  { int jr = join_solve(c); if (jr) return jr; }                  // the scan uses slot 0 of the pair scratch
  EvhRansacArgs R = pair_ransac_args(c, ransac_thr, ransac_max_iters, ransac_conf, force_max_iters);
  R.pts2 = const_cast<float*>(d_rows); R.npts2 = const_cast<int*>(d_counts); R.status = const_cast<int*>(d_status1);
  R.row_stride = row_cap; R.info = nullptr;
  R.H = d_H; R.out_status = d_status;
  if (d_state_in) { R.Hsup0 = d_state_in; R.Hprev0 = d_state_in + 9; }
  R.state_out = d_state_out;
  int rc;
  { EvhProfScope ps(c, EVH_ST_RANSAC_FINAL); rc = evh_launch_ransac_final(c, R, npairs, 1, npairs); }
  return rc;
}

int evh_pair_from_slots(evh_ctx* c, int cur_slot, int prev_slot, const double* h_Hsup, double* h_H, int* h_status) {
  if (!c || !h_H || !h_status) return evh_fail(c, EVH_ERR_INVALID, "evh_pair_from_slots: bad argument");
  int n = 0, st = 0;
  int rc = evh_match_static_from_slots(c, cur_slot, prev_slot, nullptr, 0, &n, &st);
  if (rc) return rc;
  if (st != EVH_PAIR_OK) { *h_status = st; memset(h_H, 0, 9 * sizeof(double)); return EVH_SUCCESS; }
  // static rows are already resident in d_pts2 / d_npts2 / d_pstatus
  EvhRansacArgs R = pair_ransac_args(c, 3.0, 2000, 0.995, 0);
  R.H = c->d_small; R.out_status = reinterpret_cast<int*>(c->d_small + 32);
  if (h_Hsup) {
    EVH_HIP(c, hipMemcpyAsync(c->d_small + 16, h_Hsup, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    R.Hsup0 = c->d_small + 16; R.Hprev0 = R.Hsup0;
  }
  rc = evh_launch_ransac_final(c, R, 1, h_Hsup ? 1 : 0, 1);
  if (rc) return rc;
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  EVH_HIP(c, hipMemcpy(h_H, c->d_small, 9 * sizeof(double), hipMemcpyDeviceToHost));
  EVH_HIP(c, hipMemcpy(h_status, c->d_small + 32, sizeof(int), hipMemcpyDeviceToHost));
  return EVH_SUCCESS;
}


// ---- N4: SIFT ------------------------------------------------------------------------------------------------------------------
int evh_sift_enable(evh_ctx* c, int max_sift_features) {
  if (!c) return EVH_ERR_INVALID;
  return evh_sift_allocate(c, max_sift_features);
}

int evh_sift_capacity(const evh_ctx* c) { return c ? c->sift_cap : EVH_ERR_INVALID; }

int evh_sift_detect_batch(evh_ctx* c, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                          int64_t row_stride, int64_t frame_stride, int w, int h) {
  if (!c) return EVH_ERR_INVALID;
  if (!c->sift_cap) return evh_fail(c, EVH_ERR_INVALID, "evh_sift_detect_batch: call evh_sift_enable first");
  int rc = join_solve(c);
  if (rc) return rc;
  const int nf = c->geom_valid ? c->g.nfeatures : std::min(500, c->max_features);
  if ((rc = ingest_level0(c, "evh_sift_detect_batch", d_frames, nframes, src_w, src_h, w, h, channels, row_stride, frame_stride, nf)))
    return rc;
  c->nframes_resident = 0;            // level 0 was rewritten: the ORB results of an earlier call no longer match it
  return evh_launch_sift(c, nframes, w, h);
}

int evh_sift_count(evh_ctx* c, int frame) {
  if (!c || frame < 0 || frame >= c->sift_frames_resident) return evh_fail(c, EVH_ERR_INVALID, "bad SIFT frame slot");
  int n = 0, fl = 0;
  EVH_HIP(c, hipMemcpyAsync(&n, c->d_sift_count + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipMemcpyAsync(&fl, c->d_sift_flags + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  if (fl) return evh_fail(c, EVH_ERR_CAPACITY, "more SIFT key points (or scale-space extrema) than evh_sift_enable reserved for a frame");
  return n;
}

int evh_sift_download(evh_ctx* c, int frame, float* h_xy, float* h_desc, int32_t* h_octave, float* h_size, float* h_angle,
                      float* h_response) {
  const int n = evh_sift_count(c, frame);
  if (n <= 0) return n;
  const size_t o = (size_t)frame * c->sift_cap;
  std::vector<float> rec((size_t)n * 8);
  std::vector<uint8_t> d8;
  EVH_HIP(c, hipMemcpyAsync(rec.data(), c->d_sift_kp + o * 8, sizeof(float) * 8 * n, hipMemcpyDeviceToHost, c->stream));
  if (h_desc) {
    d8.resize((size_t)n * 128);
    EVH_HIP(c, hipMemcpyAsync(d8.data(), c->d_sift_desc + o * 128, (size_t)n * 128, hipMemcpyDeviceToHost, c->stream));
  }
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) {
    const float* r = &rec[(size_t)i * 8];
    if (h_xy) { h_xy[2 * i] = r[0]; h_xy[2 * i + 1] = r[1]; }
    if (h_size) h_size[i] = r[2];
    if (h_angle) h_angle[i] = r[3];
    if (h_response) h_response[i] = r[4];
    if (h_octave) memcpy(&h_octave[i], &r[5], 4);
  }
  if (h_desc) for (size_t i = 0; i < (size_t)n * 128; i++) h_desc[i] = (float)d8[i];
  return n;
}

int evh_sift_octave_info(const evh_ctx* c, int octave, int* w, int* h) {
  if (!c || !c->sift_geom_valid || octave < 0) return EVH_ERR_INVALID;
  if (octave >= c->sg.noct) return 1;
  if (w) *w = c->sg.ow[octave]; if (h) *h = c->sg.oh[octave];
  return EVH_SUCCESS;
}

int evh_sift_download_gauss(evh_ctx* c, int frame, int octave, int layer, float* h_pixels) {
  if (!c || !c->sift_geom_valid || !h_pixels || octave < 0 || octave >= c->sg.noct || layer < 0 || layer > 5 || frame < 0)
    return evh_fail(c, EVH_ERR_INVALID, "evh_sift_download_gauss: bad argument");
  // only the LAST group's scale space is resident
  const int g0 = ((c->sift_frames_resident - 1) / c->sift_group) * c->sift_group;
  if (frame < g0 || frame >= c->sift_frames_resident) return evh_fail(c, EVH_ERR_INVALID, "evh_sift_download_gauss: that frame's scale space is no longer resident");
  const EvhSiftGeom& g = c->sg;
  const float* src = c->d_sift_pyr + (int64_t)(frame - g0) * c->sift_pyr_frame_floats + g.ooff[octave] + (int64_t)layer * g.os[octave] * g.oh[octave];
  EVH_HIP(c, hipMemcpy2DAsync(h_pixels, sizeof(float) * g.ow[octave], src, sizeof(float) * g.os[octave], sizeof(float) * g.ow[octave],
                              g.oh[octave], hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  return EVH_SUCCESS;
}

int evh_match_knn2_l2f32(evh_ctx* c, const float* d_q, int nq, const float* d_t, int nt, int dim, int32_t* d_idx, float* d_dist) {
  if (!c || !d_idx || !d_dist || nq < 0 || nt < 0) return evh_fail(c, EVH_ERR_INVALID, "evh_match_knn2_l2f32: bad argument");
  if (nq == 0) return EVH_SUCCESS;
  if (((uintptr_t)d_q | (uintptr_t)d_t) & 15) return evh_fail(c, EVH_ERR_INVALID, "descriptor buffers must be 16-byte aligned");
  EvhKnnF32Args A{d_q, d_t, nq, nt, dim, d_idx, d_dist};
  return evh_launch_knn2_f32(c, A);
}

int evh_ratio_unique_filter_f32(evh_ctx* c, const int32_t* d_idx, const float* d_dist, int nq, int nt, const float* d_xy_q,
                                const float* d_xy_t, double ratio, int min_matches, float* d_pts, int* h_count, int* h_status) {
  if (!c || !d_idx || !d_dist || !d_xy_q || !d_xy_t || !d_pts || !h_count || !h_status || nq < 0 || nt < 0)
    return evh_fail(c, EVH_ERR_INVALID, "evh_ratio_unique_filter_f32: bad argument");
  const int kc = std::max(std::max(nq, nt), 1);
  if (kc > 65535) return evh_fail(c, EVH_ERR_CAPACITY, "evh_ratio_unique_filter_f32: too many rows");
  if (((uintptr_t)d_pts) & 15) return evh_fail(c, EVH_ERR_INVALID, "d_pts must be 16-byte aligned");
  int* d_cnt = reinterpret_cast<int*>(c->d_small);
  EvhFilterArgs F{};
  F.idx = d_idx; F.d2 = reinterpret_cast<const uint32_t*>(d_dist); F.d2_is_dist = 1; F.knn_stride = nq;
  F.xy_q = d_xy_q; F.xy_t = d_xy_t; F.xy_slot_floats = 0;
  F.nq_fixed = nq; F.nt_fixed = nt; F.ratio = ratio; F.min_matches = min_matches;
  F.pts = d_pts; F.pts_stride = nq; F.npts = d_cnt; F.status = d_cnt + 1; F.kcap = kc;
  int rc = evh_launch_filter(c, F, 1);
  if (rc) return rc;
  int host[2];
  EVH_HIP(c, hipMemcpyAsync(host, d_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  *h_count = host[0]; *h_status = host[1];
  return EVH_SUCCESS;
}

int evh_pair_homography_batch_types(evh_ctx* c, const uint8_t* d_frames, int npairs, int mode, int src_w, int src_h,
                                    int channels, int64_t row_stride, int64_t frame_stride, int w, int h, int nfeatures,
                                    const int32_t* h_types, int ntypes, double ransac_thr, int ransac_max_iters,
                                    double ransac_conf, int force_max_iters, double* d_H, int32_t* d_status) {
  if (!c || !d_frames || !d_H || !d_status || npairs < 1) return evh_fail(c, EVH_ERR_INVALID, "evh_pair_homography_batch_types: bad argument");
  if (mode != EVH_MODE_INDEPENDENT_PAIRS && mode != EVH_MODE_STREAM) return evh_fail(c, EVH_ERR_INVALID, "unknown mode");
  const int nframes = mode == EVH_MODE_INDEPENDENT_PAIRS ? 2 * npairs : npairs + 1;
  if (nframes > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "batch needs more frame slots than max_frames");
  return pairs_types(c, "evh_pair_homography_batch_types", d_frames, nframes, npairs, mode == EVH_MODE_STREAM, src_w, src_h, w, h,
                     channels, row_stride, frame_stride, nfeatures, h_types, ntypes, ransac_thr, ransac_max_iters, ransac_conf,
                     force_max_iters, nullptr, nullptr, d_H, d_status);
}

int evh_stream_homography_batch_types(evh_ctx* c, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                                      int64_t row_stride, int64_t frame_stride, int w, int h, int nfeatures,
                                      const int32_t* h_types, int ntypes, double ransac_thr, int ransac_max_iters,
                                      double ransac_conf, int force_max_iters, const double* d_state_in, double* d_state_out,
                                      double* d_H, int32_t* d_status) {
  if (!c || !d_frames || !d_H || !d_status || nframes < 2) return evh_fail(c, EVH_ERR_INVALID, "evh_stream_homography_batch_types: bad argument");
  if (nframes > c->max_frames) return evh_fail(c, EVH_ERR_CAPACITY, "chunk needs more frame slots than max_frames");
  return pairs_types(c, "evh_stream_homography_batch_types", d_frames, nframes, nframes - 1, 1, src_w, src_h, w, h, channels,
                     row_stride, frame_stride, nfeatures, h_types, ntypes, ransac_thr, ransac_max_iters, ransac_conf,
                     force_max_iters, d_state_in, d_state_out, d_H, d_status);
}


// ---- N4: SURF --------------------------------------------------------------------------------------------------------------------
int evh_surf_enable(evh_ctx* c, int max_surf_features) {
  if (!c) return EVH_ERR_INVALID;
  return evh_surf_allocate(c, max_surf_features);
}
int evh_surf_capacity(const evh_ctx* c) { return c ? c->surf_cap : EVH_ERR_INVALID; }

int evh_surf_detect_batch(evh_ctx* c, const uint8_t* d_frames, int nframes, int src_w, int src_h, int channels,
                          int64_t row_stride, int64_t frame_stride, int w, int h, double hessian_threshold) {
  if (!c) return EVH_ERR_INVALID;
  if (!c->surf_cap) return evh_fail(c, EVH_ERR_INVALID, "evh_surf_detect_batch: call evh_surf_enable first");
  if (!(hessian_threshold >= 0)) return evh_fail(c, EVH_ERR_INVALID, "evh_surf_detect_batch: hessian_threshold must be >= 0");
  int rc = join_solve(c);
  if (rc) return rc;
  const int nf = c->geom_valid ? c->g.nfeatures : std::min(500, c->max_features);
  if ((rc = ingest_level0(c, "evh_surf_detect_batch", d_frames, nframes, src_w, src_h, w, h, channels, row_stride, frame_stride, nf)))
    return rc;
  c->nframes_resident = 0;
  return evh_launch_surf(c, nframes, w, h, (float)hessian_threshold);
}

int evh_surf_count(evh_ctx* c, int frame) {
  if (!c || frame < 0 || frame >= c->surf_frames_resident) return evh_fail(c, EVH_ERR_INVALID, "bad SURF frame slot");
  int n = 0, fl = 0;
  EVH_HIP(c, hipMemcpyAsync(&n, c->d_surf_count + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipMemcpyAsync(&fl, c->d_surf_flags + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  if (fl) return evh_fail(c, EVH_ERR_CAPACITY, "more SURF key points than evh_surf_enable reserved for a frame");
  return n;
}

int evh_surf_download(evh_ctx* c, int frame, float* h_xy, float* h_desc, float* h_size, float* h_angle, float* h_response,
                      int32_t* h_octave, int32_t* h_laplacian) {
  const int n = evh_surf_count(c, frame);
  if (n <= 0) return n;
  const size_t o = (size_t)frame * c->surf_cap;
  std::vector<float> rec((size_t)n * 8);
  EVH_HIP(c, hipMemcpyAsync(rec.data(), c->d_surf_kp + o * 8, sizeof(float) * 8 * n, hipMemcpyDeviceToHost, c->stream));
  if (h_desc) EVH_HIP(c, hipMemcpyAsync(h_desc, c->d_surf_desc + o * 128, sizeof(float) * 128 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) {
    const float* r = &rec[(size_t)i * 8];
    if (h_xy) { h_xy[2 * i] = r[0]; h_xy[2 * i + 1] = r[1]; }
    if (h_size) h_size[i] = r[2];
    if (h_angle) h_angle[i] = r[3];
    if (h_response) h_response[i] = r[4];
    if (h_octave) memcpy(&h_octave[i], &r[5], 4);
    if (h_laplacian) memcpy(&h_laplacian[i], &r[6], 4);
  }
  return n;
}

int evh_surf_download_integral(evh_ctx* c, int frame, int32_t* h_sum) {
  if (!c || !h_sum || !c->surf_tab_w || frame < 0) return evh_fail(c, EVH_ERR_INVALID, "evh_surf_download_integral: bad argument");
  const int g0 = ((c->surf_frames_resident - 1) / c->surf_group) * c->surf_group;
  if (frame < g0 || frame >= c->surf_frames_resident) return evh_fail(c, EVH_ERR_INVALID, "evh_surf_download_integral: that frame's integral image is no longer resident");
  const int w = c->surf_tab_w, h = c->surf_tab_h, st = (w + 1 + 15) & ~15;
  EVH_HIP(c, hipMemcpy2DAsync(h_sum, sizeof(int) * (w + 1), c->d_surf_sum + (int64_t)(frame - g0) * c->surf_sum_frame_ints, sizeof(int) * st,
                              sizeof(int) * (w + 1), h + 1, hipMemcpyDeviceToHost, c->stream));
  EVH_HIP(c, hipStreamSynchronize(c->stream));
  return EVH_SUCCESS;
}

}  // extern "C"
