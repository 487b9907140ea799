"""Independent checks of the oracle's restated operators (parity with OpenCV itself is unpinned -- no cv2 in the
image and no vectors in the reference; these tests pin the oracle against definitions, numpy and analytic truth)."""
import math
import os

import numpy as np
import pytest

from evenvizion_amd import synthetic as S
from oracle import oracle as O

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
        (-3, 1), (-2, 2), (-1, 3)]


def fast_by_definition(img, thr):
    """FAST-9/16 straight from its definition: corner, score = largest t keeping it a corner, 3x3 NMS."""
    h, w = img.shape
    sc = np.zeros((h, w), np.int32)
    I = img.astype(np.int32)

    def is_corner(y, x, t):
        v = I[y, x]
        r = [I[y + dy, x + dx] for dx, dy in RING]
        for sign in (1, -1):
            flags = [(sign * (p - v)) > t for p in r] * 2
            run = 0
            for f in flags:
                run = run + 1 if f else 0
                if run >= 9:
                    return True
        return False

    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(y, x, thr):
                t = thr
                while is_corner(y, x, t + 1):
                    t += 1
                sc[y, x] = t
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = sc[y, x]
            if s and all(s > sc[y + j, x + i] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i, j) != (0, 0)):
                out.append((x, y, int(s)))
    return out


def test_fast_matches_definition():
    img = np.ascontiguousarray(S.make_pair(3, 400, 224)[1][40:110, 60:150])
    xs, ys, sc = O.fast_nms(img, 20)
    assert len(xs) > 20
    assert sorted(zip(xs.tolist(), ys.tolist(), sc.tolist())) == sorted(fast_by_definition(img, 20))


def test_sincos_float_results_equal_libm():
    worst = 0.0
    for deg in np.linspace(0, 360, 200001, dtype=np.float32):
        rad = np.float32(deg * np.float32(math.pi / np.float32(180.0)))
        s, c = O.sincos(float(rad))
        worst = max(worst, abs(s - math.sin(float(rad))), abs(c - math.cos(float(rad))))
        assert np.float32(s) == np.float32(math.sin(float(rad))) and np.float32(c) == np.float32(math.cos(float(rad)))
    assert worst < 4e-16


def test_fast_atan2_accuracy():
    for y, x in [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (3, 4), (-5, 2), (1e5, 3)]:
        want = math.degrees(math.atan2(y, x)) % 360
        assert abs(O.fast_atan2(y, x) - want) < 0.3


def test_layout_matches_survey_tables():
    lw, lh, ls, lq = O.orb_layout(1280, 720, 500)
    assert list(zip(lw, lh)) == [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)]
    assert lq.tolist() == [109, 90, 75, 63, 52, 44, 36, 31]
    assert O.orb_layout(1280, 720, 2000)[3].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert O.orb_layout(3840, 2160, 4000)[3].tolist() == [869, 724, 603, 503, 419, 349, 291, 242]


def test_linear_exact_resize_properties():
    flat = np.full((50, 70), 93, np.uint8)
    assert (O.resize_linear_exact(flat, 58, 42) == 93).all()
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (60, 1))       # horizontal ramp stays monotone
    r = O.resize_linear_exact(ramp, 100, 50)
    assert (np.diff(r.astype(int), axis=1) >= 0).all() and (r[0] == r[-1]).all()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    half = O.resize_linear_exact(img, 32, 32)                            # exact 2:1 = mean of 2x2 with rounding
    want = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.abs(half.astype(int) - want).max() <= 1


def test_area_resize_properties():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    assert np.array_equal(O.resize_area(img, 80, 60), img)                                # identity = copy
    q = O.resize_area(img, 20, 15)                                                        # integer 4x4 -> block mean
    want = img.reshape(15, 4, 20, 4, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(q - want).max() <= 0.5 + 1e-9
    g = O.resize_area(np.full((658, 1170), 201, np.uint8), 400, 224)                     # the reference's geometry
    assert g.shape == (224, 400) and (g == 201).all()
    # enlarging (resize_width > frame width): INTER_AREA's bilinear emulation.  Definitions: a constant image stays
    # constant; an exact 2x enlargement duplicates every sample (area-mode coefficients: fx = 0 for both copies)
    up = O.resize_area(np.full((45, 80, 3), 77, np.uint8), 400, 225)
    assert up.shape == (225, 400, 3) and (up == 77).all()
    up2 = O.resize_area(img, 160, 120)
    assert np.array_equal(up2, np.repeat(np.repeat(img, 2, axis=0), 2, axis=1))
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8), (10, 1))                          # 100 wide, slope 2
    r = O.resize_area(ramp, 150, 15).astype(int)                                           # x1.5: monotone, within range
    assert (np.diff(r, axis=1) >= 0).all() and r.min() == 0 and r.max() == 198 and (r[0] == r[-1]).all()


def test_gaussian_blur_kernel():
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    b = O.gaussian_blur7(imp).astype(int)
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    want = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(b[7:14, 7:14], want)
    assert (O.gaussian_blur7(np.full((30, 30), 100, np.uint8)) == ((257 * 257 * 100 + 32768) >> 16)).all()


def test_jacobi_against_numpy():
    rng = np.random.default_rng(2)
    for n in (8, 9):
        M = rng.normal(size=(n, n)); A = M @ M.T
        W, V = O.jacobi(A)
        assert np.all(np.diff(W) <= 1e-12)
        assert np.allclose(np.sort(W), np.linalg.eigvalsh(A), rtol=1e-10, atol=1e-10)
        assert np.allclose(V @ V.T, np.eye(n), atol=1e-12)
        assert np.allclose(V @ A @ V.T, np.diag(W), atol=1e-9)


def test_dlt_and_ransac_recover_ground_truth():
    rng = np.random.default_rng(5)
    Ht = S.random_h(rng, 400)
    a = rng.uniform(0, 400, (200, 2))
    p = (Ht @ np.c_[a, np.ones(200)].T).T
    b = p[:, :2] / p[:, 2:]
    H = O.dlt(a, b)
    assert np.allclose(H / H[2, 2], Ht / Ht[2, 2], rtol=0, atol=2e-4)   # float32 inputs
    bn = b + rng.normal(0, 0.3, b.shape)
    out = rng.choice(200, 60, replace=False)
    bn[out] = rng.uniform(0, 400, (60, 2))
    H, mask, info = O.find_homography(a, bn)
    assert H is not None and mask[out].sum() <= 3 and mask.sum() >= 130
    c = np.array([[0, 0, 1], [400, 0, 1], [0, 224, 1], [400, 224, 1]], float).T
    pa = H @ c; pb = Ht @ c
    assert np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max() < 0.5
    assert info[0] < 50 and info[2] >= 1                                # adaptive stop, LM ran
    H4, m4, _ = O.find_homography(a[:4], b[:4])
    assert m4.tolist() == [1, 1, 1, 1] and np.allclose(H4 / H4[2, 2], Ht / Ht[2, 2], atol=1e-3)
    assert O.find_homography(a[:3], b[:3])[0] is None
    line = np.c_[np.arange(20.), np.arange(20.)]
    assert O.find_homography(line, line)[0] is None                      # every sample collinear -> no model


def test_knn2_bruteforce_and_ties():
    rng = np.random.default_rng(6)
    q = rng.integers(0, 256, (40, 32), dtype=np.uint8); t = rng.integers(0, 256, (55, 32), dtype=np.uint8)
    t[9] = t[2]; q[0] = t[2]
    idx, d2 = O.knn2(q, t)
    D = ((q[:, None, :].astype(int) - t[None].astype(int)) ** 2).sum(-1)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order) and np.array_equal(d2, np.take_along_axis(D, order, 1))
    assert idx[0].tolist() == [2, 9]
    i1, _ = O.knn2(q, t[:1])
    assert (i1[:, 1] == -1).all()


def test_pair_and_stream_semantics():
    prev, cur, Ht = S.make_pair(2000, 400, 224)
    st, H = O.pair_gray(cur, prev)
    assert st == 0
    c = np.array([[0, 0, 1], [400, 0, 1], [0, 224, 1], [400, 224, 1]], float).T
    pa = H @ c; pb = Ht @ c
    assert np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max() < 8.0
    flat = np.full((224, 400), 50, np.uint8)
    assert O.pair_gray(flat, flat)[0] == O.NO_DESCRIPTORS
    frames, _ = S.make_stream(5, 4, 400, 224)
    Hs, sts, rc = O.stream_gray(frames)
    assert rc == -1 and sts.tolist() == [0, 0, 0]
    bad = frames.copy(); bad[2] = 50                                     # a flat frame in the middle: 2 failing pairs
    Hs2, st2, rc2 = O.stream_gray(bad)
    assert rc2 == -1 and st2.tolist() == [0, 1, 1]
    assert np.array_equal(Hs2[1], Hs2[0]) and np.array_equal(Hs2[2], Hs2[0])   # none_H_processing: previous H
    bad0 = frames.copy(); bad0[0] = 50
    assert O.stream_gray(bad0)[2] == 0                                   # failing FIRST pair: the reference raises


def test_sift_and_surf_restatements_are_geometrically_sound():
    """The SIFT / SURF halves of the oracle are restated from recall (parity unpinned).  What CAN be checked without the
    operator: on a synthetic pair with a known H_true (cur pixel -> prev pixel) the ratio-test matches of each detector
    reproject within a fraction of a pixel, key points come out in the operator's documented order, descriptors have the
    documented norms, and the helper functions agree with their definitions."""
    from evenvizion_amd import synthetic as S
    prev, cur, Ht = S.make_pair(7, 400, 224)
    for name, det in (("sift", O.sift_detect), ("surf", O.surf_detect)):
        a, b = det(cur), det(prev)
        assert len(a["xy"]) > 300 and len(b["xy"]) > 300
        idx, dist = O.knn2_f32(a["desc"], b["desc"])
        good = dist[:, 0] < 0.5 * dist[:, 1]
        assert good.sum() > 100
        pa = a["xy"][good].astype(np.float64); pb = b["xy"][idx[good, 0]].astype(np.float64)
        pp = np.c_[pa, np.ones(len(pa))] @ Ht.T
        err = np.linalg.norm(pp[:, :2] / pp[:, 2:] - pb, axis=1)
        assert np.median(err) < 0.3 and (err < 1.5).mean() > 0.97, (name, np.median(err))
    s = O.sift_detect(cur)
    key = list(zip(s["xy"][:, 0].tolist(), s["xy"][:, 1].tolist()))
    assert key == sorted(key)                                              # removeDuplicatedSorted: by x, then y
    nrm = np.linalg.norm(s["desc"], axis=1)
    assert np.all(s["desc"] == np.rint(s["desc"])) and s["desc"].max() <= 255 and abs(np.median(nrm) - 512) < 4
    u = O.surf_detect(cur)
    assert np.all(np.diff(u["response"]) <= 0) and u["response"].min() > 400   # KeypointGreater order, hessianThreshold
    assert np.allclose(np.linalg.norm(u["desc"], axis=1), 1.0, atol=1e-5)
    assert set(np.unique(u["laplacian"]).tolist()) <= {-1, 0, 1}
    ii = O.integral(cur)
    assert np.array_equal(ii[1:, 1:], cur.astype(np.int64).cumsum(0).cumsum(1)) and not ii[0].any() and not ii[:, 0].any()
    lib = O.lib()
    lib.evo_sift_exp32f.restype = lib.evo_sift_exp2.restype = __import__("ctypes").c_float
    lib.evo_sift_exp32f.argtypes = lib.evo_sift_exp2.argtypes = [__import__("ctypes").c_float]
    for x in (-30.0, -7.25, -1.0, -0.01, 0.0, 0.5, 3.0):
        assert abs(lib.evo_sift_exp32f(x) - np.exp(np.float32(x))) <= 4e-7 * np.exp(x)       # hal::exp32f: ~1e-7 relative
    for x in (0.0, 1 / 3, 0.5, 1.1666, -0.4):
        assert lib.evo_sift_exp2(x) == np.float32(2.0 ** np.float64(np.float32(x)))          # 2^x rounded to float


def test_fast_corner_predicate_equals_skimage():
    """Independent pin of the FAST-9/16 corner test (frame_processing.py:59-61 -> ORB -> FAST): the oracle's corner mask
    equals, pixel for pixel, the mask of skimage.feature.corner_fast(n=9) -- scikit-image 0.18.3's own Cython
    implementation, captured in the build container by tools/make_skimage_fixture.py into tests/golden/skimage_fast9.npz
    (images included; blocks with plateaus and ties, smooth texture with isolated points and lines, noise at three
    contrasts so that differences exactly at the threshold are frequent)."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_fast9.npz"))
    total = 0
    for i in range(3):
        img = d["img%d" % i]
        h, w = img.shape
        want = np.unpackbits(d["mask%d" % i])[:h * w].reshape(h, w).astype(bool)
        score = O.fast_score_map(img, 20)
        assert np.array_equal(score > 0, want), "image %d" % i
        # the score is the largest threshold at which the pixel is still a corner (cornerScore: best arc's min |diff| - 1)
        ys, xs = np.nonzero(score)
        for y, x in list(zip(ys, xs))[::97]:
            s = int(score[y, x])
            assert s >= 20 and O.fast_score_map(img, s)[y, x] == s and O.fast_score_map(img, s + 1)[y, x] == 0
        total += int(want.sum())
    assert total > 10000


def test_sift_scale_space_against_scipy():
    """Independent sanity of the SIFT scale space (sigma schedule, octave chain; parity with OpenCV stays unpinned): layer i
    of octave o equals a direct Gaussian of the documented absolute blur sigma_i = 1.6 * 2^(i/3) * 2^o (in pixels of the
    doubled image, which itself carries blur 1.0), computed by scipy.ndimage in float64 from the doubled image.  The two
    differ only by kernel truncation and float32 rounding: a few hundredths of a gray level."""
    from scipy import ndimage
    from evenvizion_amd import synthetic as S
    _, cur, _ = S.make_pair(5, 160, 120)
    pyr = O.sift_gauss_pyramid(cur)
    assert pyr[0].shape == (6, 240, 320)
    # the doubled image: recover it by undoing nothing -- blur it ourselves from the 2x bilinear upsample of the frame
    up = ndimage.zoom(cur.astype(np.float64), 2, order=1, mode="nearest", grid_mode=True)
    assert up.shape == (240, 320)
    for i in range(6):
        sig_abs = 1.6 * 2.0 ** (i / 3.0)
        want = ndimage.gaussian_filter(up, np.sqrt(sig_abs ** 2 - 1.0), mode="mirror", truncate=5.0)
        got = pyr[0][i].astype(np.float64)
        inner = (slice(16, -16), slice(16, -16))                     # away from the border conventions of the upsample
        assert np.abs(got[inner] - want[inner]).max() < 0.75, (i, np.abs(got[inner] - want[inner]).max())
        assert np.abs(got[inner] - want[inner]).mean() < 0.05, (i, np.abs(got[inner] - want[inner]).mean())
    # next octave = every second pixel of layer 3 (blur 2 * 1.6), then the same schedule
    for o in range(1, len(pyr)):
        hh, ww = pyr[o][0].shape                                     # (odd sizes round down)
        assert np.array_equal(pyr[o][0], pyr[o - 1][3][::2, ::2][:hh, :ww])


def test_orb_orientation_and_brief_equal_skimage():
    """Independent pin of ORB's orientation and descriptor stages (K5, K6; frame_processing.py:59-61): on the oracle's own
    key points of a stored frame, scikit-image's corner_orientations gives the same intensity-centroid angle (within
    fastAtan2's approximation error) and its Cython _orb_loop -- steered by the same angles, reading the oracle's blurred
    level image -- gives the same 256 bits for every key point; the 1024 numbers of the sampling pattern (restated here
    from the published table) equal scikit-image's copy.  Fixture: tests/golden/skimage_orb.npz, made by
    tools/make_skimage_fixture.py in the build container."""
    import os, re
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_orb.npz"))
    # the fixture lists the key points retainBest keeps under "all ties, row-major order" (oracle order mode 0); orientation and
    # descriptor bits do not depend on which order mode selected the points
    O.set_orb_order(0)
    try:
        kp = O.orb_detect(d["gray"], 500)
    finally:
        O.set_orb_order(1)
    # the oracle's key points, regrouped by level like the fixture (stable within a level)
    order = np.concatenate([np.nonzero(kp["octave"] == l)[0] for l in range(8)])
    assert np.array_equal(kp["octave"][order], d["octave"]) and np.array_equal(kp["lx"][order], d["lx"]) \
        and np.array_equal(kp["ly"][order], d["ly"]), "fixture is stale: regenerate it"
    ang = np.deg2rad(kp["angle"][order].astype(np.float64))
    diff = np.abs((ang - d["skimage_angle_rad"] + np.pi) % (2 * np.pi) - np.pi)
    assert np.rad2deg(diff.max()) < 0.02                      # fastAtan2: <= 0.3 degrees by its contract, 0.01 in practice
    assert np.array_equal(kp["desc"][order], d["skimage_desc"])
    assert len(order) > 400 and len(np.unique(kp["octave"])) == 8
    # the pattern table itself, both copies in the repo
    root = os.path.join(os.path.dirname(__file__), "..")
    txt = [l for l in open(os.path.join(root, "oracle", "orb_pattern.inc")) if not l.strip().startswith("//")]
    mine = np.array([int(x) for x in re.findall(r"-?\d+", "".join(txt))][:1024]).reshape(256, 4)
    assert np.array_equal(mine, d["skimage_pattern"].astype(int))
    data = np.loadtxt(os.path.join(root, "evenvizion_amd", "data", "orb_pattern_31.txt"), dtype=int)
    assert np.array_equal(data, d["skimage_pattern"].astype(int))


def test_pyramid_resize_against_scipy():
    """Independent sanity of the pyramid's `resize` (INTER_LINEAR_EXACT: pixel centres aligned, 8.8 fixed-point weights;
    conventions, not bits) against scipy.ndimage.zoom(order=1, grid_mode=True) in float64: within one gray level
    everywhere and unbiased.  (The descriptor stage's 8-bit Gaussian is NOT checked this way: its kernel, rounded tap by tap
    as the operator's fixed-point path does, sums to 257/256 per pass -- a +0.8 % gain a float blur does not have.)"""
    from scipy import ndimage
    from evenvizion_amd import synthetic as S
    _, cur, _ = S.make_pair(9, 400, 224)
    pyr = O.orb_pyramid(cur)
    for l in range(1, 4):
        lh, lw = pyr[l].shape
        ph, pw = pyr[l - 1].shape
        want = ndimage.zoom(pyr[l - 1].astype(np.float64), (lh / ph, lw / pw), order=1, mode="nearest", grid_mode=True)
        assert want.shape == (lh, lw)
        diff = pyr[l].astype(np.float64) - want
        assert np.abs(diff).max() <= 1.0 and abs(diff.mean()) < 0.05, (l, np.abs(diff).max(), diff.mean())


def test_introselect_statement_equals_libstdcxx(tmp_path):
    """The oracle's statement of libstdc++'s introselect (what fixes the ORDER retainBest leaves ORB's key points in) against
    the real std::nth_element of this image's libstdc++: 20 300 sequences with few distinct keys, every permutation equal.
    (At the depth limit the oracle calls std::nth_element itself; the device's heap-select is checked against that in
    tests/test_gpu_order.py.)"""
    import subprocess
    exe = str(tmp_path / "nth_check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "nth_check.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, src])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("mismatches 0 of "), r.stdout + r.stderr


def test_sift_filter_fusion_modes_differ_only_where_stated():
    """evo_set_sift_blur_mode: 2 (the pinned split) fuses the multiply-adds of the float Gaussian filter in columns
    [0, w & ~7) of the row pass and [0, w & ~15) of the column pass.  So against mode 1 (fusion everywhere) the FIRST blurred
    image may differ only right of w & ~15, a width that is a multiple of 16 gives identical scale spaces, and mode 0 (no
    fusion) differs from both nearly everywhere."""
    rng = np.random.default_rng(3)
    img = (S.make_pair(5, 96, 64)[0][:57, :53]).copy()          # doubled: 106 x 114 -> 106 & ~15 = 96
    assert img.shape == (57, 53)
    pyr = {}
    try:
        for m in (0, 1, 2):
            O.set_sift_blur_mode(m)
            pyr[m] = [p.copy() for p in O.sift_gauss_pyramid(img)]
    finally:
        O.set_sift_blur_mode(2)
    assert O.get_sift_blur_mode() == 2
    base1, base2 = pyr[1][0][0], pyr[2][0][0]                      # octave 0, layer 0: one blur of the doubled frame
    assert base1.shape[1] == 106
    assert np.array_equal(base1[:, :96], base2[:, :96]) and not np.array_equal(base1[:, 96:], base2[:, 96:])
    assert (pyr[0][0][0] != base2).mean() > 0.3
    wide = S.make_pair(6, 96, 64)[0][:40, :64].copy()             # doubled: 128 columns, octaves 128 / 64 / 32 / 16: all multiples of 16
    try:
        O.set_sift_blur_mode(1); a = O.sift_gauss_pyramid(wide)
        O.set_sift_blur_mode(2); b = O.sift_gauss_pyramid(wide)
    finally:
        O.set_sift_blur_mode(2)
    assert all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4]))
