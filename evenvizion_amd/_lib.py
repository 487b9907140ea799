"""ctypes binding of libevhip.so (include/evhip.h).  Host code stays Python; every kernel is reached through this
thin C-ABI layer.  PyTorch-ROCm is used only to own device memory (tensor.data_ptr()) and for torch.distributed.

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVHIP_LIBRARY selects another build of the same C ABI (A/B measurements of two kernel variants on one box)
LIB_PATH = os.environ.get("EVHIP_LIBRARY") or os.path.join(_HERE, "libevhip.so")

EVH_SUCCESS = 0
PAIR_OK, PAIR_NO_DESCRIPTORS, PAIR_FEW_MATCHES, PAIR_NO_PROVISIONAL_H, PAIR_LOW_INLIER_RATIO, PAIR_NO_FINAL_H, \
    PAIR_CAPACITY = range(7)
ORDER_CANONICAL, ORDER_OPENCV = 0, 1
SOLVER_EXACT, SOLVER_FAST = 0, 1
MAX_FEATURES = 5984   # EVH_MAX_FEATURES (include/evhip.h): largest max_features a context accepts
MODE_INDEPENDENT_PAIRS, MODE_STREAM = 0, 1

# every symbol include/evhip.h declares, with its ctypes signature
_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pi = C.POINTER(C.c_int)
SIGNATURES = {
    "evh_create": (_i, [_i, _i, _i, _i, _i, _vp, C.POINTER(_vp)]),
    "evh_destroy": (None, [_vp]),
    "evh_last_error_string": (C.c_char_p, [_vp]),
    "evh_stream": (_vp, [_vp]),
    "evh_synchronize": (_i, [_vp]),
    "evh_version": (_i, []),
    "evh_set_async_solve": (_i, [_vp, _i]),
    "evh_solve_wait": (_i, [_vp, _vp]),
    "evh_profile_enable": (_i, [_vp, _i]),
    "evh_profile_read": (_i, [_vp, _vp, _vp]),
    "evh_profile_stage_name": (C.c_char_p, [_i]),
    "evh_resize_area_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _vp, _i, _i, _i64, _i64]),
    "evh_fixed_plane_field": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "evh_superposition_scan": (_i, [_vp, _vp, _i, _vp]),
    "evh_transform_points": (_i, [_vp, _vp, _i, _vp, _vp, _i, _d, _d, _i, _vp]),
    "evh_orb_detect_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i]),
    "evh_orb_detect_batch_resized": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i, _i]),
    "evh_stream_homography_batch_resized": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i, _i, _d, _i, _d, _i, _vp, _vp, _vp, _vp]),
    "evh_set_fast_lift": (_i, [_vp, _i]),
    "evh_set_solver_mode": (_i, [_vp, _i]),
    "evh_get_solver_mode": (_i, [_vp]),
    "evh_set_keypoint_order": (_i, [_vp, _i]),
    "evh_get_keypoint_order": (_i, [_vp]),
    "evh_set_fast_share": (_i, [_vp, _i]),
    "evh_set_fast_hint": (_i, [_vp, _i]),
    "evh_orb_count": (_i, [_vp, _i]),
    "evh_orb_capacity": (_i, [_vp]),
    "evh_orb_download": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evh_orb_detect_compute": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "evh_resize_area_u8c3": (_i, [_vp, _vp, _i, _i, _vp, _i, _i]),
    "evh_orb_level_info": (_i, [_vp, _i, _pi, _pi, _pi, C.POINTER(C.c_float)]),
    "evh_orb_download_level": (_i, [_vp, _i, _i, _vp]),
    "evh_orb_download_candidates": (_i, [_vp, _i, _i, _vp, _i]),
    "evh_match_knn2_l2u8": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "evh_match_knn2_l2u8x128": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "evh_match_knn2_hamming": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "evh_ratio_unique_filter": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _d, _i, _vp, _pi, _pi]),
    "evh_find_homography_ransac": (_i, [_vp, _vp, _i, _d, _i, _d, _vp, _vp, _pi, _vp]),
    "evh_find_homography_ransac_fixed": (_i, [_vp, _vp, _i, _d, _i, _d, _vp, _vp, _pi, _vp]),
    "evh_static_filter": (_i, [_vp, _vp, _vp, _i, _vp, _pi]),
    "evh_pair_homography_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i, _d, _i, _d, _i, _vp, _vp]),
    "evh_stream_homography_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _d, _i, _d, _i, _vp, _vp, _vp, _vp]),
    "evh_multi_stream_homography_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i, _d, _i, _d, _i, _vp, _vp, _vp, _vp]),
    "evh_stream_static_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _d, _i, _d, _i, _vp, _i, _vp, _vp]),
    "evh_stream_scan": (_i, [_vp, _vp, _i, _vp, _vp, _i, _d, _i, _d, _i, _vp, _vp, _vp, _vp]),
    "evh_pair_from_slots": (_i, [_vp, _i, _i, _vp, _vp, _pi]),
    "evh_match_static_from_slots": (_i, [_vp, _i, _i, _vp, _i, _pi, _pi]),
    "evh_compute_homography": (_i, [_vp, _vp, _i, _vp, _vp, _pi]),
    # N4: SIFT + multi-type pairs
    "evh_sift_enable": (_i, [_vp, _i]),
    "evh_sift_capacity": (_i, [_vp]),
    "evh_sift_detect_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i]),
    "evh_sift_count": (_i, [_vp, _i]),
    "evh_sift_download": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evh_sift_octave_info": (_i, [_vp, _i, _pi, _pi]),
    "evh_sift_download_gauss": (_i, [_vp, _i, _i, _i, _vp]),
    "evh_surf_enable": (_i, [_vp, _i]),
    "evh_surf_capacity": (_i, [_vp]),
    "evh_surf_detect_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i, _d]),
    "evh_surf_count": (_i, [_vp, _i]),
    "evh_surf_download": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evh_surf_download_integral": (_i, [_vp, _i, _vp]),
    "evh_match_knn2_l2f32": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "evh_ratio_unique_filter_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _d, _i, _vp, _pi, _pi]),
    "evh_pair_homography_batch_types": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i, _i, _i, _vp, _i, _d, _i, _d, _i, _vp, _vp]),
    "evh_stream_homography_batch_types": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i, _i, _vp, _i, _d, _i, _d, _i, _vp, _vp, _vp, _vp]),
}
FEATURE_ORB, FEATURE_SIFT, FEATURE_SURF = 0, 1, 2
FEATURE_CODES = {"ORB": FEATURE_ORB, "SIFT": FEATURE_SIFT, "SURF": FEATURE_SURF}


class EvhError(RuntimeError):
    pass


def build(force=False):
    """Compile libevhip.so for gfx950 with hipcc (evenvizion_amd/csrc/Makefile)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def load():
    """Load libevhip.so and bind every declared symbol.  Raises if the library is missing -- there is no fallback."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and must be loaded FIRST so that
        # libevhip.so binds to that same runtime (two runtimes in one process cannot both see the device).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise EvhError("libevhip.so not found at %s: build it with `make -C evenvizion_amd/csrc` "
                           "(the HIP library is the only compute backend)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if an include/evhip.h symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _hp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One evh_ctx: one device, one HIP stream, device buffers sized at creation (reused across calls)."""

    def __init__(self, device=0, max_w=1280, max_h=720, max_features=500, max_frames=2, stream=None):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.evh_create(int(device), int(max_w), int(max_h), int(max_features), int(max_frames),
                                 C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != EVH_SUCCESS:
            raise EvhError("evh_create failed (%d): %s" % (rc, self.lib.evh_last_error_string(None).decode()))
        self.h = h
        self.device = device
        self.max_frames = max_frames
        self.max_features = max_features
        self.max_w, self.max_h = max_w, max_h

    def close(self):
        if getattr(self, "h", None):
            self.lib.evh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _enter(self):
        """Order this context's (non-blocking) stream behind the work already queued on torch's current stream: the
        tensors a call receives may still be being filled there (torch.zeros, copies, a previous op's output)."""
        self.order_after_torch()

    def _check(self, rc):
        if rc < 0:
            raise EvhError("libevhip error %d: %s" % (rc, self.lib.evh_last_error_string(self.h).decode()))
        return rc

    @property
    def stream(self):
        return self.lib.evh_stream(self.h)

    def synchronize(self):
        self._check(self.lib.evh_synchronize(self.h))

    def set_async_solve(self, on=True):
        self._check(self.lib.evh_set_async_solve(self.h, int(bool(on))))

    def solve_wait(self, stream=None):
        """Make `stream` (a hipStream_t handle; None = the context's main stream) wait for the pending solve."""
        self._check(self.lib.evh_solve_wait(self.h, C.c_void_p(stream) if stream else None))

    def profile_enable(self, on=True):
        self._check(self.lib.evh_profile_enable(self.h, int(bool(on))))

    def profile_read(self):
        """-> {stage: (launch_groups, total_ms)} since the last read (synchronises the stream)."""
        ms = np.zeros(9, np.float32); cnt = np.zeros(9, np.int32)
        self._check(self.lib.evh_profile_read(self.h, _hp(ms), _hp(cnt)))
        return {self.lib.evh_profile_stage_name(i).decode(): (int(cnt[i]), float(ms[i])) for i in range(9)}

    # ---- K0 ----
    def resize_area(self, src, dst):
        """src/dst: CUDA uint8 tensors [n,h,w] or [n,h,w,c], contiguous."""
        self._enter()
        n, sh, sw = src.shape[:3]
        cn = 1 if src.dim() == 3 else src.shape[3]
        dh, dw = dst.shape[1:3]
        self._check(self.lib.evh_resize_area_u8(self.h, src.data_ptr(), n, sw, sh, cn, sw * cn, sw * sh * cn,
                                                dst.data_ptr(), dw, dh, dw * cn, dw * dh * cn))

    def resize_area_bgr(self, src, dst):
        """One BGR image: src CUDA uint8 [sh,sw,3] -> dst [dh,dw,3] (imutils.resize / INTER_AREA)."""
        self._enter()
        sh, sw = src.shape[:2]; dh, dw = dst.shape[:2]
        self._check(self.lib.evh_resize_area_u8c3(self.h, src.data_ptr(), sw, sh, dst.data_ptr(), dw, dh))

    def set_fast_lift(self, on=True):
        self._check(self.lib.evh_set_fast_lift(self.h, int(bool(on))))

    def set_solver_mode(self, mode):
        """SOLVER_EXACT (default): LM's 8x8 systems by the operator's Jacobi eigen-solve (H bit-identical to the oracle);
        SOLVER_FAST: by LDL^T (stream pairs ~45 % cheaper; H within ~1e-3 px of the exact mode's, see include/evhip.h)."""
        self._check(self.lib.evh_set_solver_mode(self.h, int(mode)))

    def get_solver_mode(self):
        return int(self.lib.evh_get_solver_mode(self.h))

    def set_keypoint_order(self, mode):
        """ORDER_OPENCV (default): key points leave retainBest in the order (and set) OpenCV 3.4.2 on libstdc++ leaves them --
        the reference's; ORDER_CANONICAL: all ties kept, (level, y, x) order (faster: FAST threshold lifting applies)."""
        self._check(self.lib.evh_set_keypoint_order(self.h, int(mode)))

    def get_keypoint_order(self):
        return int(self.lib.evh_get_keypoint_order(self.h))

    def set_fast_hint(self, on=True):
        self._check(self.lib.evh_set_fast_hint(self.h, int(bool(on))))

    def set_fast_share(self, on=True):
        self._check(self.lib.evh_set_fast_share(self.h, int(bool(on))))

    def fixed_plane_max(self, Hsup, w, h, field=None):
        """Hsup f64[n,3,3] -> f64[n]: max fixed-plane coordinate over the w x h grid of each matrix."""
        Hs = np.ascontiguousarray(Hsup, np.float64).reshape(-1, 9)
        out = np.zeros(len(Hs), np.float64)
        self._check(self.lib.evh_fixed_plane_field(self.h, _hp(Hs), len(Hs), int(w), int(h),
                                                   field.data_ptr() if field is not None else None, _hp(out)))
        return out

    # ---- N1 ----
    def superposition_scan(self, Hs):
        """Hs f64[n,3,3] per-frame H in frame order -> f64[n,3,3] running superposition (utils.superposition_dict)."""
        Hs = np.ascontiguousarray(Hs, np.float64).reshape(-1, 9)
        out = np.zeros_like(Hs)
        self._check(self.lib.evh_superposition_scan(self.h, _hp(Hs), len(Hs), _hp(out)))
        return out.reshape(-1, 3, 3)

    def transform_points(self, mats, idx, pts, kx=1.0, ky=1.0, decimals=-1):
        """pts f64[n,2], idx i32[n] (row of mats f64[m,3,3] per point) -> f64[n,2] transformed (and rounded) points."""
        mats = np.ascontiguousarray(mats, np.float64).reshape(-1, 9)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
        idx = np.ascontiguousarray(idx, np.int32).reshape(-1)
        out = np.zeros_like(pts)
        self._check(self.lib.evh_transform_points(self.h, _hp(mats), len(mats), _hp(idx), _hp(pts), len(pts), float(kx),
                                                  float(ky), int(decimals), _hp(out)))
        return out

    # ---- K1..K6 ----
    def orb_detect_batch(self, frames, nfeatures=500, resize_to=None):
        """frames: CUDA uint8 tensor [n,h,w] (gray) or [n,h,w,3] (BGR), contiguous.  resize_to=(w, h): the frames are
        shrunk to that working size inside the ingest kernel (imutils.resize fused into level 0)."""
        self._enter()
        n, h, w = frames.shape[:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        if resize_to is not None and tuple(resize_to) != (w, h):
            self._check(self.lib.evh_orb_detect_batch_resized(self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn,
                                                              int(resize_to[0]), int(resize_to[1]), nfeatures))
            return
        self._check(self.lib.evh_orb_detect_batch(self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn, nfeatures))

    def orb_detect_compute(self, frame, nfeatures=500):
        """One frame (CUDA uint8 [h,w] or [h,w,3]) -> (xy f32[n,2], desc u8[n,32], octave i32[n]); the single-frame
        form of cv2.ORB_create().detectAndCompute (frame_processing.py:60-61)."""
        self._enter()
        h, w = frame.shape[:2]
        cn = 1 if frame.dim() == 2 else frame.shape[2]
        cap = self.lib.evh_orb_capacity(self.h)
        xy = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 32), np.uint8); oc = np.zeros(cap, np.int32)
        n = C.c_int(0)
        self._check(self.lib.evh_orb_detect_compute(self.h, frame.data_ptr(), w, h, cn, nfeatures, _hp(xy), _hp(desc),
                                                    _hp(oc), C.byref(n)))
        return xy[:n.value].copy(), desc[:n.value].copy(), oc[:n.value].copy()

    def orb_download(self, frame):
        cap = self.lib.evh_orb_capacity(self.h)
        xy = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 32), np.uint8)
        oc = np.zeros(cap, np.int32); lxy = np.zeros((cap, 2), np.int32)
        rs = np.zeros(cap, np.float32); an = np.zeros(cap, np.float32)
        n = self._check(self.lib.evh_orb_download(self.h, frame, _hp(xy), _hp(desc), _hp(oc), _hp(lxy), _hp(rs), _hp(an)))
        return dict(xy=xy[:n].copy(), desc=desc[:n].copy(), octave=oc[:n].copy(), lx=lxy[:n, 0].copy(),
                    ly=lxy[:n, 1].copy(), response=rs[:n].copy(), angle=an[:n].copy())

    def level_info(self, level):
        w = C.c_int(); h = C.c_int(); q = C.c_int(); s = C.c_float()
        self._check(self.lib.evh_orb_level_info(self.h, level, C.byref(w), C.byref(h), C.byref(q), C.byref(s)))
        return w.value, h.value, q.value, s.value

    def download_level(self, frame, level):
        w, h, _, _ = self.level_info(level)
        out = np.zeros((h, w), np.uint8)
        self._check(self.lib.evh_orb_download_level(self.h, frame, level, _hp(out)))
        return out

    def download_candidates(self, frame, level):
        w, h, _, _ = self.level_info(level)
        cap = (w // 2 + 1) * (h // 2 + 1) + 64
        buf = np.zeros(cap, np.uint32)
        n = self._check(self.lib.evh_orb_download_candidates(self.h, frame, level, _hp(buf), cap))
        p = buf[:min(n, cap)]
        return (p & 0xFFF).astype(np.int32), ((p >> 12) & 0xFFF).astype(np.int32), (p >> 24).astype(np.int32)

    # ---- K7 + glue ----
    def knn2(self, q, t, idx, d2, hamming=False):
        self._enter()
        f = self.lib.evh_match_knn2_hamming if hamming else (
            self.lib.evh_match_knn2_l2u8x128 if q.shape[1] == 128 else self.lib.evh_match_knn2_l2u8)
        self._check(f(self.h, q.data_ptr(), q.shape[0], t.data_ptr(), t.shape[0], idx.data_ptr(), d2.data_ptr()))

    def ratio_unique_filter(self, idx, d2, xy_q, xy_t, pts, ratio=0.5, min_matches=4):
        self._enter()
        n = C.c_int(); st = C.c_int()
        self._check(self.lib.evh_ratio_unique_filter(self.h, idx.data_ptr(), d2.data_ptr(), idx.shape[0], xy_t.shape[0],
                                                     xy_q.data_ptr(), xy_t.data_ptr(), float(ratio), int(min_matches),
                                                     pts.data_ptr(), C.byref(n), C.byref(st)))
        return n.value, st.value

    # ---- K8/K9 ----
    def find_homography(self, pts, thr=3.0, max_iters=2000, conf=0.995, force_max_iters=False):
        self._enter()
        n = pts.shape[0]
        H = np.zeros(9, np.float64); mask = np.zeros(max(n, 1), np.uint8); info = np.zeros(3, np.int32)
        found = C.c_int()
        f = self.lib.evh_find_homography_ransac_fixed if force_max_iters else self.lib.evh_find_homography_ransac
        self._check(f(self.h, pts.data_ptr() if n else None, n, float(thr), int(max_iters), float(conf), _hp(H),
                      _hp(mask), C.byref(found), _hp(info)))
        return (H.reshape(3, 3) if found.value else None), mask[:n].copy(), info

    def static_filter(self, H, pts, out):
        self._enter()
        H = np.ascontiguousarray(H, np.float64).reshape(9)
        n = C.c_int()
        self._check(self.lib.evh_static_filter(self.h, _hp(H), pts.data_ptr(), pts.shape[0], out.data_ptr(), C.byref(n)))
        return n.value

    # ---- fused ----
    def pair_homography_batch(self, frames, npairs, mode, out_H, out_status, nfeatures=500, thr=3.0, max_iters=2000,
                              conf=0.995, force_max_iters=False):
        self._enter()
        h, w = frames.shape[1:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        self._check(self.lib.evh_pair_homography_batch(self.h, frames.data_ptr(), npairs, mode, w, h, cn, w * cn,
                                                       w * h * cn, nfeatures, float(thr), int(max_iters), float(conf),
                                                       int(bool(force_max_iters)), out_H.data_ptr(),
                                                       out_status.data_ptr()))

    def stream_homography_batch(self, frames, out_H, out_status, state_in=None, state_out=None, nfeatures=500, thr=3.0,
                                max_iters=2000, conf=0.995, force_max_iters=False, resize_to=None):
        """frames: CUDA uint8 [n,h,w(,3)], n >= 2 consecutive frames of one stream -> n-1 pairs (stream semantics).
        resize_to=(w, h): full-size frames, the reference's resize_width fused into the ingest kernel."""
        self._enter()
        n, h, w = frames.shape[:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        if resize_to is not None and tuple(resize_to) != (w, h):
            self._check(self.lib.evh_stream_homography_batch_resized(
                self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn, int(resize_to[0]), int(resize_to[1]), nfeatures,
                float(thr), int(max_iters), float(conf), int(bool(force_max_iters)),
                state_in.data_ptr() if state_in is not None else None,
                state_out.data_ptr() if state_out is not None else None, out_H.data_ptr(), out_status.data_ptr()))
            return
        self._check(self.lib.evh_stream_homography_batch(
            self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn, nfeatures, float(thr), int(max_iters), float(conf),
            int(bool(force_max_iters)), state_in.data_ptr() if state_in is not None else None,
            state_out.data_ptr() if state_out is not None else None, out_H.data_ptr(), out_status.data_ptr()))

    def multi_stream_homography_batch(self, frames, out_H, out_status, state_in=None, state_out=None, nfeatures=500,
                                      thr=3.0, max_iters=2000, conf=0.995, force_max_iters=False):
        """frames: CUDA uint8 [S,F,h,w(,3)] -- S independent streams of F consecutive frames each; out_H f64[S,F-1,9],
        out_status i32[S,F-1]; state_in / state_out f64[S,18] carry {H_sup, H_prev} of every stream between calls."""
        self._enter()
        S, F, h, w = frames.shape[:4]
        cn = 1 if frames.dim() == 4 else frames.shape[4]
        self._check(self.lib.evh_multi_stream_homography_batch(
            self.h, frames.data_ptr(), S, F, w, h, cn, w * cn, w * h * cn, nfeatures, float(thr), int(max_iters),
            float(conf), int(bool(force_max_iters)), state_in.data_ptr() if state_in is not None else None,
            state_out.data_ptr() if state_out is not None else None, out_H.data_ptr(), out_status.data_ptr()))

    # ---- N4: SIFT + multi-type pairs ----
    def sift_enable(self, max_sift_features=8192):
        self._check(self.lib.evh_sift_enable(self.h, int(max_sift_features)))

    def sift_detect_batch(self, frames, resize_to=None):
        """frames: CUDA uint8 [n,h,w] or [n,h,w,3]; SIFT_create().detectAndCompute on each (frame_processing.py:62-64)."""
        self._enter()
        n, h, w = frames.shape[:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        dw, dh = (w, h) if resize_to is None else (int(resize_to[0]), int(resize_to[1]))
        self._check(self.lib.evh_sift_detect_batch(self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn, dw, dh))

    def sift_download(self, frame):
        cap = self.lib.evh_sift_capacity(self.h)
        xy = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 128), np.float32); oc = np.zeros(cap, np.int32)
        sz = np.zeros(cap, np.float32); an = np.zeros(cap, np.float32); rs = np.zeros(cap, np.float32)
        n = self._check(self.lib.evh_sift_download(self.h, frame, _hp(xy), _hp(desc), _hp(oc), _hp(sz), _hp(an), _hp(rs)))
        return dict(xy=xy[:n].copy(), desc=desc[:n].copy(), octave=oc[:n].copy(), size=sz[:n].copy(), angle=an[:n].copy(),
                    response=rs[:n].copy())

    def sift_octaves(self):
        out = []
        while True:
            w = C.c_int(); h = C.c_int()
            rc = self.lib.evh_sift_octave_info(self.h, len(out), C.byref(w), C.byref(h))
            if rc != 0:
                break
            out.append((w.value, h.value))
        return out

    def sift_download_gauss(self, frame, octave, layer):
        w, h = self.sift_octaves()[octave]
        out = np.zeros((h, w), np.float32)
        self._check(self.lib.evh_sift_download_gauss(self.h, frame, octave, layer, _hp(out)))
        return out

    def surf_enable(self, max_surf_features=4096):
        self._check(self.lib.evh_surf_enable(self.h, int(max_surf_features)))

    def surf_detect_batch(self, frames, resize_to=None, hessian_threshold=400.0):
        """frames: CUDA uint8 [n,h,w] or [n,h,w,3]; SURF_create(extended=1, hessianThreshold=400).detectAndCompute
        (frame_processing.py:65-67)."""
        self._enter()
        n, h, w = frames.shape[:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        dw, dh = (w, h) if resize_to is None else (int(resize_to[0]), int(resize_to[1]))
        self._surf_shape = (dh, dw)
        self._check(self.lib.evh_surf_detect_batch(self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn, dw, dh,
                                                   float(hessian_threshold)))

    def surf_download(self, frame):
        cap = self.lib.evh_surf_capacity(self.h)
        xy = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 128), np.float32); sz = np.zeros(cap, np.float32)
        an = np.zeros(cap, np.float32); rs = np.zeros(cap, np.float32); oc = np.zeros(cap, np.int32); lp = np.zeros(cap, np.int32)
        n = self._check(self.lib.evh_surf_download(self.h, frame, _hp(xy), _hp(desc), _hp(sz), _hp(an), _hp(rs), _hp(oc), _hp(lp)))
        return dict(xy=xy[:n].copy(), desc=desc[:n].copy(), size=sz[:n].copy(), angle=an[:n].copy(), response=rs[:n].copy(),
                    octave=oc[:n].copy(), laplacian=lp[:n].copy())

    def surf_download_integral(self, frame):
        h, w = self._surf_shape
        out = np.zeros((h + 1, w + 1), np.int32)
        self._check(self.lib.evh_surf_download_integral(self.h, frame, _hp(out)))
        return out

    def knn2_f32(self, q, t, idx, dist):
        """q, t: CUDA float32 [n,dim] (dim 64 or 128); idx int32 [nq,2], dist float32 [nq,2]."""
        self._enter()
        self._check(self.lib.evh_match_knn2_l2f32(self.h, q.data_ptr(), q.shape[0], t.data_ptr(), t.shape[0], q.shape[1],
                                                  idx.data_ptr(), dist.data_ptr()))

    def ratio_unique_filter_f32(self, idx, dist, xy_q, xy_t, pts, ratio=0.5, min_matches=4):
        self._enter()
        n = C.c_int(); st = C.c_int()
        self._check(self.lib.evh_ratio_unique_filter_f32(self.h, idx.data_ptr(), dist.data_ptr(), idx.shape[0], xy_t.shape[0],
                                                         xy_q.data_ptr(), xy_t.data_ptr(), float(ratio), int(min_matches),
                                                         pts.data_ptr(), C.byref(n), C.byref(st)))
        return n.value, st.value

    @staticmethod
    def _types(features):
        codes = np.ascontiguousarray([FEATURE_CODES[f] if isinstance(f, str) else int(f) for f in features], np.int32)
        return codes

    def pair_homography_batch_types(self, frames, npairs, mode, out_H, out_status, features, nfeatures=500, thr=3.0,
                                    max_iters=2000, conf=0.995, force_max_iters=False, resize_to=None):
        self._enter()
        h, w = frames.shape[1:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        dw, dh = (w, h) if resize_to is None else (int(resize_to[0]), int(resize_to[1]))
        t = self._types(features)
        self._multi_used = True
        self._check(self.lib.evh_pair_homography_batch_types(
            self.h, frames.data_ptr(), npairs, mode, w, h, cn, w * cn, w * h * cn, dw, dh, nfeatures, _hp(t), len(t), float(thr),
            int(max_iters), float(conf), int(bool(force_max_iters)), out_H.data_ptr(), out_status.data_ptr()))

    def stream_homography_batch_types(self, frames, out_H, out_status, features, state_in=None, state_out=None, nfeatures=500,
                                      thr=3.0, max_iters=2000, conf=0.995, force_max_iters=False, resize_to=None):
        """stream_homography_batch with a list of feature types ("SIFT", "ORB", ... in the order the reference's
        FrameProcessing would loop over them, frame_processing.py:91-104)."""
        self._enter()
        n, h, w = frames.shape[:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        dw, dh = (w, h) if resize_to is None else (int(resize_to[0]), int(resize_to[1]))
        t = self._types(features)
        self._multi_used = True
        self._check(self.lib.evh_stream_homography_batch_types(
            self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn, dw, dh, nfeatures, _hp(t), len(t), float(thr),
            int(max_iters), float(conf), int(bool(force_max_iters)), state_in.data_ptr() if state_in is not None else None,
            state_out.data_ptr() if state_out is not None else None, out_H.data_ptr(), out_status.data_ptr()))

    def _torch_stream(self):
        """The context's (non-blocking) HIP stream as a torch stream, for ordering against torch work."""
        import torch
        if getattr(self, "_ext_stream", None) is None:
            self._ext_stream = torch.cuda.ExternalStream(self.stream, device=self.device)
        return self._ext_stream

    def order_after_torch(self):
        """Kernels enqueued by this context from now on wait for the work already on torch's current stream."""
        import torch
        self._torch_stream().wait_stream(torch.cuda.current_stream(self.device))

    def order_torch_after(self):
        """torch's current stream waits for everything this context has enqueued so far (main and solve stream)."""
        import torch
        self.solve_wait()
        torch.cuda.current_stream(self.device).wait_stream(self._torch_stream())

    def stream_static_batch(self, frames, nfeatures=500, thr=3.0, max_iters=2000, conf=0.995, force_max_iters=False):
        """Phase 1 of the two-phase stream path (evh_stream_static_batch): frames CUDA uint8 [n,h,w(,3)] ->
        (rows f32[n-1,cap,4], counts i32[n-1], status1 i32[n-1]) as CUDA tensors.  Asynchronous, but ordered with
        torch's current stream on both sides (inputs produced by torch ops, outputs consumed by torch ops / RCCL)."""
        import torch
        n, h, w = frames.shape[:3]
        cn = 1 if frames.dim() == 3 else frames.shape[3]
        cap = self.lib.evh_orb_capacity(self.h)
        rows = torch.zeros((max(n - 1, 0), cap, 4), dtype=torch.float32, device=frames.device)
        counts = torch.zeros(max(n - 1, 0), dtype=torch.int32, device=frames.device)
        status1 = torch.zeros(max(n - 1, 0), dtype=torch.int32, device=frames.device)
        if n < 2:
            return rows, counts, status1          # an empty block (more ranks than pairs)
        self.order_after_torch()
        self._check(self.lib.evh_stream_static_batch(
            self.h, frames.data_ptr(), n, w, h, cn, w * cn, w * h * cn, nfeatures, float(thr), int(max_iters), float(conf),
            int(bool(force_max_iters)), rows.data_ptr(), cap, counts.data_ptr(), status1.data_ptr()))
        self.order_torch_after()
        return rows, counts, status1

    def stream_scan(self, rows, counts, status1, state_in=None, state_out=None, thr=3.0, max_iters=2000, conf=0.995,
                    force_max_iters=False):
        """Phase 2 (evh_stream_scan): the sequential compute_homography / matrix_superposition scan over all pairs in
        stream order -> (H f64[npairs,9], status i32[npairs]) CUDA tensors.  Asynchronous, ordered with torch's
        current stream like stream_static_batch."""
        import torch
        npairs, cap = rows.shape[0], rows.shape[1]
        rows = rows.contiguous(); counts = counts.contiguous(); status1 = status1.contiguous()
        H = torch.zeros((npairs, 9), dtype=torch.float64, device=rows.device)
        st = torch.zeros(npairs, dtype=torch.int32, device=rows.device)
        self.order_after_torch()
        self._check(self.lib.evh_stream_scan(
            self.h, rows.data_ptr(), cap, counts.data_ptr(), status1.data_ptr(), npairs, float(thr), int(max_iters),
            float(conf), int(bool(force_max_iters)), state_in.data_ptr() if state_in is not None else None,
            state_out.data_ptr() if state_out is not None else None, H.data_ptr(), st.data_ptr()))
        self.order_torch_after()
        return H, st

    def match_static_from_slots(self, cur_slot, prev_slot):
        cap = self.lib.evh_orb_capacity(self.h)
        pts = np.zeros((cap, 4), np.float32)
        n = C.c_int(); st = C.c_int()
        self._check(self.lib.evh_match_static_from_slots(self.h, cur_slot, prev_slot, _hp(pts), cap, C.byref(n), C.byref(st)))
        return st.value, pts[:n.value].copy()

    def compute_homography(self, pts, Hsup=None):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 4)
        H = np.zeros(9, np.float64); st = C.c_int()
        hs = None if Hsup is None else np.ascontiguousarray(Hsup, np.float64).reshape(9)
        self._check(self.lib.evh_compute_homography(self.h, _hp(pts), pts.shape[0], _hp(hs), _hp(H), C.byref(st)))
        return st.value, H.reshape(3, 3)

    def pair_from_slots(self, cur_slot, prev_slot, Hsup=None):
        H = np.zeros(9, np.float64); st = C.c_int()
        hs = None if Hsup is None else np.ascontiguousarray(Hsup, np.float64).reshape(9)
        self._check(self.lib.evh_pair_from_slots(self.h, cur_slot, prev_slot, _hp(hs), _hp(H), C.byref(st)))
        return st.value, H.reshape(3, 3)
