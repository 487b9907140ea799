"""ctypes front-end of the CPU ORACLE (oracle/libevz_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.  See oracle/evz_oracle.h for what is restated and its parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("EVZ_ORACLE_SO") or os.path.join(_HERE, "libevz_oracle.so")   # override: instrumented builds (tools/hygiene.sh)

OK, NO_DESCRIPTORS, FEW_MATCHES, NO_PROVISIONAL_H, LOW_INLIER_RATIO, NO_FINAL_H = range(6)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.evo_fast_atan2.restype = C.c_float
        _lib.evo_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.evo_orb_pyramid.restype = C.c_int64
        _lib.evo_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return _lib


def set_orb_order(mode):
    """1 (default): key points leave retainBest in the order OpenCV 3.4.2 + libstdc++ leave them; 0: row-major, all ties kept."""
    lib().evo_set_orb_order(int(mode))


def get_orb_order():
    return int(lib().evo_get_orb_order())


def set_sift_blur_mode(mode):
    """2 (default, pinned): SIFT's Gaussian filter fuses its multiply-adds in the vector bodies only; 0: nowhere; 1: everywhere."""
    lib().evo_set_sift_blur_mode(int(mode))


def get_sift_blur_mode():
    return int(lib().evo_get_sift_blur_mode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def bgr2gray(bgr):
    bgr = _u8(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().evo_bgr2gray(_p(bgr), w, h, w * 3, _p(out))
    return out


def resize_dims(w0, h0, width):
    """imutils.resize(width=): r = width / float(w); dim = (width, int(h * r))."""
    r = width / float(w0)
    return width, int(h0 * r)


def resize_area(img, dw, dh):
    img = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    sh, sw = img.shape[:2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    rc = lib().evo_resize_area(_p(img), sw, sh, cn, _p(out), dw, dh)
    assert rc == 0
    return out


def resize_linear_exact(img, dw, dh):
    img = _u8(img)
    sh, sw = img.shape
    out = np.empty((dh, dw), np.uint8)
    lib().evo_resize_linear_exact(_p(img), sw, sh, _p(out), dw, dh)
    return out


def orb_layout(w, h, nfeatures=500):
    lw = np.zeros(8, np.int32); lh = np.zeros(8, np.int32)
    ls = np.zeros(8, np.float32); lq = np.zeros(8, np.int32)
    lib().evo_orb_layout(w, h, nfeatures, _p(lw), _p(lh), _p(ls), _p(lq))
    return lw, lh, ls, lq


def orb_pyramid(gray):
    gray = _u8(gray)
    h, w = gray.shape
    lw, lh, _, _ = orb_layout(w, h)
    buf = np.empty(int((lw.astype(np.int64) * lh).sum()), np.uint8)
    lib().evo_orb_pyramid(_p(gray), w, h, _p(buf))
    out, off = [], 0
    for l in range(8):
        n = int(lw[l]) * int(lh[l])
        out.append(buf[off:off + n].reshape(int(lh[l]), int(lw[l])))
        off += n
    return out


def fast_nms(img, threshold=20):
    img = _u8(img)
    h, w = img.shape
    cap = max(1, (w * h) // 4 + 16)
    xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32)
    n = lib().evo_fast_nms(_p(img), w, h, threshold, _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def fast_score_map(img, threshold=20):
    """FAST-9/16 corner score of every pixel at `threshold` (0 = not a corner), before the 3x3 non-maximum suppression."""
    img = _u8(img)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().evo_fast_score_map(_p(img), w, h, int(threshold), _p(out))
    return out


def orb_level_candidates(img, quota):
    img = _u8(img)
    h, w = img.shape
    cap = max(1, (w * h) // 4 + 16)
    xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32)
    n = lib().evo_orb_level_candidates(_p(img), w, h, int(quota), _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def gaussian_blur7(img):
    img = _u8(img)
    h, w = img.shape
    out = np.empty_like(img)
    lib().evo_gaussian_blur7(_p(img), w, h, _p(out))
    return out


def orb_detect(gray, nfeatures=500):
    """-> dict(xy f32[N,2], desc u8[N,32], octave, lx, ly, response, angle)"""
    gray = _u8(gray)
    h, w = gray.shape
    cap = 2 * nfeatures + 4096
    xy = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 32), np.uint8)
    oc = np.zeros(cap, np.int32); lx = np.zeros(cap, np.int32); ly = np.zeros(cap, np.int32)
    rs = np.zeros(cap, np.float32); an = np.zeros(cap, np.float32)
    n = lib().evo_orb_detect(_p(gray), w, h, nfeatures, _p(xy), _p(desc), _p(oc), _p(lx), _p(ly), _p(rs), _p(an), cap)
    return dict(xy=xy[:n].copy(), desc=desc[:n].copy(), octave=oc[:n].copy(), lx=lx[:n].copy(), ly=ly[:n].copy(),
                response=rs[:n].copy(), angle=an[:n].copy())


def sincos(x):
    s = C.c_double(); c = C.c_double()
    lib().evo_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def fast_atan2(y, x):
    return float(lib().evo_fast_atan2(float(y), float(x)))


def knn2(q, t, hamming=False):
    q = _u8(q).reshape(-1, 32); t = _u8(t).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32); d2 = np.zeros((len(q), 2), np.uint32)
    f = lib().evo_knn2_hamming if hamming else lib().evo_knn2_l2
    f(_p(q), len(q), _p(t), len(t), _p(idx), _p(d2))
    return idx, d2


def ratio_unique(idx, d2, ratio=0.5):
    idx = np.ascontiguousarray(idx, np.int32); d2 = np.ascontiguousarray(d2, np.uint32)
    n = len(idx)
    oq = np.zeros(max(n, 1), np.int32); ot = np.zeros(max(n, 1), np.int32)
    f = lib().evo_ratio_unique
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    m = f(_p(idx), _p(d2), n, float(ratio), _p(oq), _p(ot))
    return oq[:m].copy(), ot[:m].copy()


def remove_double(a, b):
    a = _f32(a).reshape(-1, 2); b = _f32(b).reshape(-1, 2)
    oa = np.zeros_like(a); ob = np.zeros_like(b)
    m = lib().evo_remove_double(_p(a), _p(b), len(a), _p(oa), _p(ob))
    return oa[:m].copy(), ob[:m].copy()


def find_homography(a, b, thr=3.0, max_iters=2000, conf=0.995, force_max_iters=False):
    """-> (H f64[3,3] or None, mask u8[n], info)"""
    a = _f32(a).reshape(-1, 2); b = _f32(b).reshape(-1, 2)
    n = len(a)
    H = np.zeros(9, np.float64); mask = np.zeros(max(n, 1), np.uint8); info = np.zeros(3, np.int32)
    f = lib().evo_find_homography_ex
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                  C.c_void_p]
    ok = f(_p(a), _p(b), n, float(thr), int(max_iters), float(conf), int(bool(force_max_iters)), _p(H), _p(mask), _p(info))
    return (H.reshape(3, 3) if ok else None), mask[:n].copy(), info


def dlt(src, dst):
    src = _f32(src).reshape(-1, 2); dst = _f32(dst).reshape(-1, 2)
    H = np.zeros(9, np.float64)
    ok = lib().evo_dlt(_p(src), _p(dst), len(src), _p(H))
    return H.reshape(3, 3) if ok else None


def jacobi(A):
    A = np.array(A, np.float64, order="C")
    n = A.shape[0]
    W = np.zeros(n); V = np.zeros((n, n))
    lib().evo_jacobi(_p(A), n, _p(W), _p(V))
    return W, V


def static_filter(H, a, b):
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    a = _f32(a).reshape(-1, 2); b = _f32(b).reshape(-1, 2)
    oa = np.zeros_like(a); ob = np.zeros_like(b)
    m = lib().evo_static_filter(_p(H), _p(a), _p(b), len(a), _p(oa), _p(ob))
    return oa[:m].copy(), ob[:m].copy()


def compute_homography(a, b, Hsup=None, force_max_iters=False):
    """-> (status, H f64[3,3])"""
    a = _f32(a).reshape(-1, 2); b = _f32(b).reshape(-1, 2)
    H = np.zeros(9, np.float64)
    hs = None if Hsup is None else np.ascontiguousarray(Hsup, np.float64).reshape(9)
    st = lib().evo_compute_homography_ex(_p(a), _p(b), len(a), None if hs is None else _p(hs),
                                         int(bool(force_max_iters)), _p(H))
    return st, H.reshape(3, 3)


def matrix_superposition(H, Hsup, first=False):
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    out = np.zeros(9)
    hs = np.zeros(9) if Hsup is None else np.ascontiguousarray(Hsup, np.float64).reshape(9)
    lib().evo_matrix_superposition(_p(H), _p(hs), int(bool(first)), _p(out))
    return out.reshape(3, 3)


def match_static(xy_a, desc_a, xy_b, desc_b):
    """KeyPoints(a).match_static_kps(KeyPoints(b)) -> (status, pts_a, pts_b)"""
    xy_a = _f32(xy_a).reshape(-1, 2); xy_b = _f32(xy_b).reshape(-1, 2)
    desc_a = _u8(desc_a).reshape(-1, 32); desc_b = _u8(desc_b).reshape(-1, 32)
    na = len(xy_a)
    oa = np.zeros((max(na, 1), 2), np.float32); ob = np.zeros((max(na, 1), 2), np.float32)
    n = C.c_int(0)
    st = lib().evo_match_static(_p(xy_a), _p(desc_a), na, _p(xy_b), _p(desc_b), len(xy_b), _p(oa), _p(ob), C.byref(n))
    return st, oa[:n.value].copy(), ob[:n.value].copy()


def pair_gray(cur, prev, nfeatures=500, Hsup=None):
    cur = _u8(cur); prev = _u8(prev)
    h, w = cur.shape
    H = np.zeros(9, np.float64)
    hs = None if Hsup is None else np.ascontiguousarray(Hsup, np.float64).reshape(9)
    st = lib().evo_pair_gray(_p(cur), _p(prev), w, h, nfeatures, None if hs is None else _p(hs), _p(H))
    return st, H.reshape(3, 3)


def pairs_gray_batch(frames, nfeatures=500, threads=1):
    """frames u8[2B,h,w] laid out (prev0, cur0, prev1, cur1, ...) -> (H f64[B,3,3], status i32[B])"""
    frames = _u8(frames)
    nb = frames.shape[0] // 2
    h, w = frames.shape[1:]
    H = np.zeros((nb, 9), np.float64); st = np.zeros(nb, np.int32)
    lib().evo_pairs_gray_batch(_p(frames), nb, w, h, nfeatures, int(threads), _p(H), _p(st))
    return H.reshape(nb, 3, 3), st


def stream_gray(frames, nfeatures=500, force_max_iters=False):
    """frames u8[F,h,w] -> (H f64[F-1,3,3], status i32[F-1], failed_first_pair_index or -1)"""
    frames = _u8(frames)
    nf, h, w = frames.shape
    H = np.zeros((nf - 1, 9), np.float64); st = np.zeros(nf - 1, np.int32)
    rc = lib().evo_stream_gray_ex(_p(frames), nf, w, h, nfeatures, int(bool(force_max_iters)), _p(H), _p(st))
    return H.reshape(-1, 3, 3), st, rc


# ---- N4: SIFT (oracle/evz_sift.cpp; restated from recall; pinned jointly by tests/test_capture_golden.py) -------------------------------------------
def sift_layout(w, h):
    ow = np.zeros(16, np.int32); oh = np.zeros(16, np.int32)
    n = lib().evo_sift_layout(int(w), int(h), _p(ow), _p(oh), 16)
    return [(int(ow[o]), int(oh[o])) for o in range(n)]


def sift_gauss_pyramid(gray):
    """-> list over octaves of float32[6, h_o, w_o] (the Gaussian scale space of SIFT_create().detectAndCompute)."""
    gray = _u8(gray)
    h, w = gray.shape
    lay = sift_layout(w, h)
    total = sum(6 * a * b for a, b in lay)
    buf = np.zeros(total, np.float32)
    f = lib().evo_sift_gauss_pyramid
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    n = f(_p(gray), w, h, _p(buf), total)
    assert n == total
    out, off = [], 0
    for (a, b) in lay:
        out.append(buf[off:off + 6 * a * b].reshape(6, b, a))
        off += 6 * a * b
    return out


def sift_detect(gray, cap=20000):
    """-> dict(xy f32[N,2], desc f32[N,128] (integer values 0..255, as the operator returns them), octave i32[N] (packed
    octave | layer << 8 | ...), size, angle, response), in the operator's own order (removeDuplicatedSorted)."""
    gray = _u8(gray)
    h, w = gray.shape
    xy = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 128), np.uint8); oc = np.zeros(cap, np.int32)
    sz = np.zeros(cap, np.float32); an = np.zeros(cap, np.float32); rs = np.zeros(cap, np.float32)
    n = lib().evo_sift_detect(_p(gray), w, h, _p(xy), _p(desc), _p(oc), _p(sz), _p(an), _p(rs), cap)
    assert n <= cap, "raise cap"
    return dict(xy=xy[:n].copy(), desc=desc[:n].astype(np.float32), octave=oc[:n].copy(), size=sz[:n].copy(),
                angle=an[:n].copy(), response=rs[:n].copy())


def knn2_f32(q, t):
    """BruteForce knnMatch(q, t, 2) on float32 descriptors -> (idx i32[nq,2], dist f32[nq,2])."""
    q = _f32(q); t = _f32(t)
    dim = q.shape[1]
    idx = np.zeros((len(q), 2), np.int32); dist = np.zeros((len(q), 2), np.float32)
    lib().evo_knn2_l2f32(_p(q), len(q), _p(t), len(t), dim, _p(idx), _p(dist))
    return idx, dist


def ratio_unique_f32(idx, dist, ratio=0.5):
    idx = np.ascontiguousarray(idx, np.int32); dist = _f32(dist)
    n = len(idx)
    oq = np.zeros(max(n, 1), np.int32); ot = np.zeros(max(n, 1), np.int32)
    f = lib().evo_ratio_unique_f32
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    m = f(_p(idx), _p(dist), n, float(ratio), _p(oq), _p(ot))
    return oq[:m].copy(), ot[:m].copy()


def match_static_f32(xy_a, desc_a, xy_b, desc_b):
    """KeyPoints(a).match_static_kps(KeyPoints(b)) on float descriptors -> (status, pts_a, pts_b)"""
    xy_a = _f32(xy_a).reshape(-1, 2); xy_b = _f32(xy_b).reshape(-1, 2)
    desc_a = _f32(desc_a); desc_b = _f32(desc_b)
    na = len(xy_a)
    dim = desc_a.shape[1] if desc_a.ndim == 2 and na else (desc_b.shape[1] if desc_b.ndim == 2 else 128)
    oa = np.zeros((max(na, 1), 2), np.float32); ob = np.zeros((max(na, 1), 2), np.float32)
    n = C.c_int(0)
    st = lib().evo_match_static_f32(_p(xy_a), _p(desc_a), na, _p(xy_b), _p(desc_b), len(xy_b), int(dim), _p(oa), _p(ob),
                                    C.byref(n))
    return st, oa[:n.value].copy(), ob[:n.value].copy()


FEATURE_CODES = {"ORB": 0, "SIFT": 1, "SURF": 2}


def stream_gray_types(frames, features, nfeatures=500, force_max_iters=False, Hsup_forced=None, return_npts=False):
    """frames u8[F,h,w], features e.g. ["SIFT", "ORB"] -> (H f64[F-1,3,3], status i32[F-1], failed_first_pair_index or -1).
    Hsup_forced f64[F-1,3,3]: pair k (0-based, k >= 1) is solved in the plane Hsup_forced[k-1] instead of the plane the stream
    accumulated itself (pair-by-pair comparison with a recorded run); return_npts adds the RANSAC input sizes."""
    frames = _u8(frames)
    nf, h, w = frames.shape
    t = np.ascontiguousarray([FEATURE_CODES[f] for f in features], np.int32)
    H = np.zeros((nf - 1, 9), np.float64); st = np.zeros(nf - 1, np.int32); npts = np.zeros(nf - 1, np.int32)
    hs = None
    if Hsup_forced is not None:
        hs = np.ascontiguousarray(Hsup_forced, np.float64).reshape(-1, 9)
        assert hs.shape[0] >= nf - 1
    rc = lib().evo_stream_gray_types_ex(_p(frames), nf, w, h, nfeatures, _p(t), len(t), 1 if force_max_iters else 0,
                                        _p(hs) if hs is not None else None, _p(H), _p(st), _p(npts))
    if return_npts:
        return H.reshape(-1, 3, 3), st, rc, npts
    return H.reshape(-1, 3, 3), st, rc


# ---- N4: SURF (oracle/evz_surf.cpp; restated from recall; pinned jointly by tests/test_capture_golden.py) -------------------------------------------
def integral(gray):
    gray = _u8(gray)
    h, w = gray.shape
    out = np.zeros((h + 1, w + 1), np.int32)
    lib().evo_integral(_p(gray), w, h, _p(out))
    return out


def surf_detect(gray, cap=65536):
    """SURF_create(extended=1, hessianThreshold=400).detectAndCompute -> dict(xy, desc f32[N,128], size, angle, response,
    octave, laplacian) in the operator's own order (response descending)."""
    gray = _u8(gray)
    h, w = gray.shape
    xy = np.zeros((cap, 2), np.float32); desc = np.zeros((cap, 128), np.float32); sz = np.zeros(cap, np.float32)
    an = np.zeros(cap, np.float32); rs = np.zeros(cap, np.float32); oc = np.zeros(cap, np.int32); lp = np.zeros(cap, np.int32)
    n = lib().evo_surf_detect(_p(gray), w, h, _p(xy), _p(desc), _p(sz), _p(an), _p(rs), _p(oc), _p(lp), cap)
    assert n <= cap, "raise cap"
    return dict(xy=xy[:n].copy(), desc=desc[:n].copy(), size=sz[:n].copy(), angle=an[:n].copy(), response=rs[:n].copy(),
                octave=oc[:n].copy(), laplacian=lp[:n].copy())
