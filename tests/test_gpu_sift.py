"""GPU parity of the SIFT branch (frame_processing.py:62-64, matching.py:102-108 on float descriptors,
frame_processing.py:91-104 over a type list) against the CPU oracle (oracle/evz_sift.cpp -- restated from recall, parity
unpinned: what is checked here is HIP == oracle).

Bar, stated: the scale space, key points (x, y, size, angle, response, packed octave) and descriptors are compared
BIT FOR BIT (float32 arrays with np.array_equal; no tolerance is needed because both sides evaluate every float
expression one IEEE operation at a time in the same order).  H of a ["SIFT", "ORB"] stream: rtol 1e-9 against the
oracle stream (north-star bar 1e-3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from evenvizion_amd import synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_ctx(w, h, frames=4, sift=4096, feats=500):
    from evenvizion_amd._lib import Context
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    c = Context(device=0, max_w=w, max_h=h, max_features=feats, max_frames=frames)
    c.sift_enable(sift)
    return c


@pytest.mark.parametrize("w,h", [(400, 224), (333, 217)])
def test_sift_scale_space(w, h):
    prev, cur, _ = S.make_pair(21, w, h)
    c = make_ctx(w, h)
    try:
        c.sift_detect_batch(dev(np.stack([prev, cur])))
        assert c.sift_octaves() == O.sift_layout(w, h)
        for f, img in enumerate((prev, cur)):
            want = O.sift_gauss_pyramid(img)
            for o in range(len(want)):
                for l in range(6):
                    got = c.sift_download_gauss(f, o, l)
                    assert np.array_equal(got, want[o][l]), "frame %d octave %d layer %d: max |diff| %g" % (
                        f, o, l, np.abs(got - want[o][l]).max())
    finally:
        c.close()


def _same_keypoints(g, o):
    assert len(g["xy"]) == len(o["xy"]), (len(g["xy"]), len(o["xy"]))
    for k in ("xy", "size", "angle", "response"):
        assert np.array_equal(g[k].view(np.uint32), o[k].view(np.uint32)), k
    assert np.array_equal(g["octave"], o["octave"])
    assert np.array_equal(g["desc"], o["desc"])


@pytest.mark.parametrize("w,h,cap", [(400, 224, 4096), (333, 217, 4096), (1280, 720, 40960)])
def test_sift_keypoints_and_descriptors(w, h, cap):
    prev, cur, _ = S.make_pair(23, w, h)
    bgr = S.gray_to_bgr(cur)
    c = make_ctx(w, h, frames=2, sift=cap)
    try:
        c.sift_detect_batch(dev(np.stack([prev, cur])))
        for f, img in enumerate((prev, cur)):
            _same_keypoints(c.sift_download(f), O.sift_detect(img, cap=65536))
        c.sift_detect_batch(dev(np.stack([bgr, bgr])))                      # BGR entry: cvtColor first
        _same_keypoints(c.sift_download(1), O.sift_detect(cur, cap=65536))
    finally:
        c.close()


def test_sift_flat_and_tiny_frames():
    c = make_ctx(128, 96, frames=2, sift=1024)
    try:
        flat = np.full((2, 96, 128), 90, np.uint8)
        c.sift_detect_batch(dev(flat))
        assert len(c.sift_download(0)["xy"]) == 0 and len(O.sift_detect(flat[0])["xy"]) == 0
        a, b, _ = S.make_pair(3, 64, 64)
        c.sift_detect_batch(dev(np.stack([a, b])))
        for f, img in enumerate((a, b)):
            _same_keypoints(c.sift_download(f), O.sift_detect(img))
    finally:
        c.close()


def test_rare_exit_inputs_sift_and_surf():
    """The inputs of tools/hygiene_inputs.py (chosen so that the rarely taken exits of the SIFT / SURF statements are taken:
    singular and near-singular refinement systems, refinement leaving the layer range or the border or not converging, blobs
    at the border, windows larger than the image, flat and saturated content; profiles/r04_hygiene.txt lists the execution
    counts on the oracle's side) through the device kernels: key points, order, descriptors bit for bit."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("hygiene_inputs", os.path.join(os.path.dirname(__file__), "..", "tools", "hygiene_inputs.py"))
    hy = importlib.util.module_from_spec(spec); spec.loader.exec_module(hy)
    for img in hy.detector_inputs():
        h, w = img.shape
        if w < 64 or h < 64:
            continue
        c = make_ctx(w, h, frames=2, sift=16384)
        try:
            c.surf_enable(8192)
            c.sift_detect_batch(dev(np.stack([img, img])))
            o = O.sift_detect(img, cap=65536)
            if len(o["xy"]) == 0:
                assert c.lib.evh_sift_count(c.h, 0) == 0
            else:
                _same_keypoints(c.sift_download(0), o)
            c.surf_detect_batch(dev(np.stack([img, img])))
            os_ = O.surf_detect(img)
            if len(os_["xy"]) == 0:
                assert c.lib.evh_surf_count(c.h, 0) == 0
            else:
                _same_surf(c.surf_download(0), os_)
        finally:
            c.close()


def test_sift_capacity_is_flagged():
    from evenvizion_amd._lib import EvhError
    a, b, _ = S.make_pair(5, 400, 224)
    c = make_ctx(400, 224, frames=2, sift=512)              # the frame has ~2 400 key points
    try:
        c.sift_detect_batch(dev(np.stack([a, b])))
        with pytest.raises(EvhError):
            c.sift_download(0)
    finally:
        c.close()


@pytest.mark.parametrize("dim", [128, 64])
def test_knn2_float_descriptors(dim):
    from evenvizion_amd._lib import Context
    rng = np.random.default_rng(dim)
    c = Context(device=0, max_w=64, max_h=64, max_features=500, max_frames=2)
    try:
        cases = [(rng.random((300, dim), dtype=np.float32) - 0.5, rng.random((411, dim), dtype=np.float32) - 0.5),
                 (rng.integers(0, 256, (257, dim)).astype(np.float32), rng.integers(0, 256, (190, dim)).astype(np.float32)),
                 (rng.random((5, dim), dtype=np.float32), rng.random((1, dim), dtype=np.float32))]
        q2 = cases[0][0].copy(); t2 = np.concatenate([q2[:40], q2[:40], cases[0][1]])      # exact ties -> lowest index
        cases.append((q2, t2))
        for q, t in cases:
            idx = torch.zeros(len(q), 2, dtype=torch.int32, device="cuda")
            dist = torch.zeros(len(q), 2, dtype=torch.float32, device="cuda")
            c.knn2_f32(dev(q), dev(t), idx, dist)
            c.synchronize()
            oi, od = O.knn2_f32(q, t)
            assert np.array_equal(idx.cpu().numpy(), oi)
            assert np.array_equal(dist.cpu().numpy().view(np.uint32), od.view(np.uint32))
    finally:
        c.close()


def test_sift_descriptor_matching_float_equals_byte_path():
    """SIFT's float32 descriptors hold integers 0..255: the fused path matches them as bytes (exact integer squared
    distances, compared after sqrt like the operator does); the float matcher must give the same neighbours."""
    a, b, _ = S.make_pair(29, 400, 224)
    oa, ob = O.sift_detect(a), O.sift_detect(b)
    c = make_ctx(400, 224, frames=2)
    try:
        idx = torch.zeros(len(oa["desc"]), 2, dtype=torch.int32, device="cuda")
        dist = torch.zeros(len(oa["desc"]), 2, dtype=torch.float32, device="cuda")
        c.knn2_f32(dev(oa["desc"]), dev(ob["desc"]), idx, dist)
        c.synchronize()
        oi, od = O.knn2_f32(oa["desc"], ob["desc"])
        assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(dist.cpu().numpy(), od)
        d2 = (od.astype(np.float64) ** 2)
        assert np.allclose(d2, np.rint(d2), atol=1e-2)        # integer squared distances
    finally:
        c.close()


@pytest.mark.parametrize("features", [["SIFT"], ["SIFT", "ORB"], ["ORB", "SIFT"]])
def test_multi_type_stream_vs_oracle(features):
    """frame_processing.py:91-104 over a type list + the stream semantics of video_processing.py:67-105."""
    w, h = 400, 224
    frames, _ = S.make_stream(31, 6, w, h)
    flat = np.full((h, w), 128, np.uint8)
    frames = np.stack([frames[0], frames[1], frames[2], flat, frames[3], frames[4], frames[5]])
    n = len(frames) - 1
    Ho, so, rc = O.stream_gray_types(frames, features)
    assert rc == -1 and list(so) == [0, 0, 1, 1, 0, 0]
    c = make_ctx(w, h, frames=len(frames), sift=4096)
    try:
        H = torch.zeros(n, 9, dtype=torch.float64, device="cuda")
        st = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        c.stream_homography_batch_types(dev(frames), H, st, features)
        c.synchronize()
        assert np.array_equal(st.cpu().numpy(), so)
        Hg = H.cpu().numpy().reshape(-1, 3, 3)
        assert np.allclose(Hg, Ho, rtol=1e-9, atol=1e-12)
        # chunked, state carried on the device
        state = torch.zeros(18, dtype=torch.float64, device="cuda")
        H1 = torch.zeros(3, 9, dtype=torch.float64, device="cuda"); s1 = torch.zeros(3, dtype=torch.int32, device="cuda")
        H2 = torch.zeros(3, 9, dtype=torch.float64, device="cuda"); s2 = torch.zeros(3, dtype=torch.int32, device="cuda")
        c.stream_homography_batch_types(dev(frames[:4]), H1, s1, features, state_out=state)
        c.stream_homography_batch_types(dev(frames[3:]), H2, s2, features, state_in=state, state_out=state)
        c.synchronize()
        assert np.array_equal(np.concatenate([H1.cpu().numpy(), H2.cpu().numpy()]).reshape(-1, 3, 3), Hg)
    finally:
        c.close()


def test_multi_type_fixed_iteration_forms_agree(monkeypatch):
    """The fixed-iteration mode through the multi-type path: RANSAC #1 of every type and the scan spread their samples over
    workgroups (k_static_hyp / k_scan_hyp + the finishing kernels, 1 000-2 000 rows per pair here, so the four-wave
    refinement passes run too).  Every form -- the single-workgroup kernels (EVH_SCAN_ONE_WG=1), the spread form, and the
    spread form with a short hypothesis table (EVH_SCAN_CHUNKS=5: the rest of the samples inside the finishing kernels) --
    against the oracle's fixed-iteration multi-type stream (evo_stream_gray_types_ex, force_max = 1)."""
    w, h = 400, 224
    frames, _ = S.make_stream(33, 4, w, h)
    n = len(frames) - 1
    c = make_ctx(w, h, frames=len(frames), sift=4096)
    try:
        outs = []
        for env in ({"EVH_SCAN_ONE_WG": "1"}, {}, {"EVH_SCAN_CHUNKS": "5"}):
            monkeypatch.delenv("EVH_SCAN_CHUNKS", raising=False); monkeypatch.delenv("EVH_SCAN_ONE_WG", raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            H = torch.zeros(n, 9, dtype=torch.float64, device="cuda")
            st = torch.full((n,), -1, dtype=torch.int32, device="cuda")
            c.stream_homography_batch_types(dev(frames), H, st, ["SIFT", "ORB"], force_max_iters=True)
            c.synchronize()
            outs.append((H.cpu().numpy(), st.cpu().numpy()))
        Ho, so, rc = O.stream_gray_types(frames, ["SIFT", "ORB"], force_max_iters=True)
        assert rc == -1 and list(so) == [0] * n
        for H, st in outs:
            assert np.array_equal(st, so) and np.allclose(H.reshape(-1, 3, 3), Ho, rtol=1e-9, atol=1e-12)
        for H, st in outs[1:]:
            assert np.array_equal(H, outs[0][0])
    finally:
        c.close()


def test_multi_type_stream_beyond_the_lds_filter():
    """More than 7 680 key points of one type per frame (SIFT on an 800x450 frame): the matching filter then keeps its work
    arrays in a global scratch (k_filter<true>) instead of LDS; stream of three frames against the oracle, through the C
    ABI and through get_homography_dict (whose SIFT slots scale with the frame area)."""
    from evenvizion_amd import runtime
    from evenvizion_amd.processing import get_homography_dict
    w, h = 800, 450
    frames, _ = S.make_stream(47, 3, w, h)
    n = len(frames) - 1
    Ho, so, rc = O.stream_gray_types(frames, ["SIFT", "ORB"])
    assert rc == -1 and list(so) == [0, 0]
    assert len(O.sift_detect(frames[1])["xy"]) > 7680          # the case this test is about
    c = make_ctx(w, h, frames=len(frames), sift=16384)
    try:
        H = torch.zeros(n, 9, dtype=torch.float64, device="cuda")
        st = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        c.stream_homography_batch_types(dev(frames), H, st, ["SIFT", "ORB"])
        c.synchronize()
        assert np.array_equal(st.cpu().numpy(), so)
        assert np.allclose(H.cpu().numpy().reshape(-1, 3, 3), Ho, rtol=1e-9, atol=1e-12)
    finally:
        c.close()
    runtime.reset()
    d = get_homography_dict(S.SyntheticCapture([S.gray_to_bgr(f) for f in frames]), resize_width=w, features_type_list=["SIFT", "ORB"])
    got = np.array([d[k]["H"] for k in range(2, len(frames) + 1)])
    assert np.allclose(got, Ho, rtol=1e-9, atol=1e-12)
    assert runtime.sift_features_for(w, h) >= 16384 and runtime.sift_features_for(400, 224) == 7168 and runtime.sift_features_for(640, 360) == 17408
    runtime.reset()


def test_multi_type_independent_pairs_and_bgr():
    w, h = 400, 224
    fr, _ = S.make_pair_batch(7, 2, w, h)
    c = make_ctx(w, h, frames=4, sift=4096)
    try:
        H = torch.zeros(2, 9, dtype=torch.float64, device="cuda")
        st = torch.full((2,), -1, dtype=torch.int32, device="cuda")
        c.pair_homography_batch_types(dev(S.gray_to_bgr(fr)), 2, 0, H, st, ["SIFT", "ORB"])
        c.synchronize()
        Hg = H.cpu().numpy().reshape(-1, 3, 3)
        for p in range(2):
            Ho, so, rc = O.stream_gray_types(fr[2 * p:2 * p + 2], ["SIFT", "ORB"])
            assert rc == -1 and so[0] == 0 and st.cpu().numpy()[p] == 0
            assert np.allclose(Hg[p], Ho[0], rtol=1e-9, atol=1e-12)
    finally:
        c.close()


def test_python_mirror_sift_and_type_list():
    """FrameProcessing(frame, ["SIFT", "ORB"]) / KeyPoints on float descriptors / get_homography_dict(features_type_list=)
    -- the reference's API with its own argument (frame_processing.py:37-108, matching.py:75-163) -- against the oracle."""
    from evenvizion_amd import runtime
    from evenvizion_amd.processing import FrameProcessing, KeyPoints, get_homography_dict, compute_homography
    runtime.reset()
    w, h = 400, 224
    frames, _ = S.make_stream(37, 5, w, h)
    bgr = [S.gray_to_bgr(f) for f in frames]
    fa, fb = FrameProcessing(bgr[1], ["SIFT", "ORB"]), FrameProcessing(bgr[0], ["SIFT", "ORB"])
    xy, desc = fa.detect_and_describe_features("SIFT")
    o1, o0 = O.sift_detect(frames[1]), O.sift_detect(frames[0])
    assert xy.dtype == np.float32 and desc.dtype == np.float32 and desc.shape[1] == 128
    assert np.array_equal(xy, o1["xy"]) and np.array_equal(desc, o1["desc"])
    # KeyPoints on float descriptors: match_kps / match_static_kps
    pa, pb = KeyPoints(o1["xy"], o1["desc"]).match_kps(KeyPoints(o0["xy"], o0["desc"]))
    oi, od = O.knn2_f32(o1["desc"], o0["desc"])
    mq, mt = O.ratio_unique_f32(oi, od)
    wa, wb = O.remove_double(o1["xy"][mq], o0["xy"][mt])
    assert np.array_equal(np.array(pa), wa) and np.array_equal(np.array(pb), wb)
    sa, sb = KeyPoints(o1["xy"], o1["desc"]).match_static_kps(KeyPoints(o0["xy"], o0["desc"]))
    st, ea, eb = O.match_static_f32(o1["xy"], o1["desc"], o0["xy"], o0["desc"])
    assert st == 0 and np.array_equal(sa, ea) and np.array_equal(sb, eb)
    # the per-pair body over both types, then compute_homography
    ca, cb = fa.concatenate_all_features_types(fb)
    H = compute_homography(ca, cb, None)
    Ho, so, rc = O.stream_gray_types(frames[:2], ["SIFT", "ORB"])
    assert rc == -1 and so[0] == 0 and np.allclose(H, Ho[0], rtol=1e-9, atol=1e-12)
    # the stream driver with a type list (chunked)
    Hs, ss, rc = O.stream_gray_types(frames, ["SIFT", "ORB"])
    d = get_homography_dict(S.SyntheticCapture(bgr), resize_width=w, chunk_frames=3, features_type_list=["SIFT", "ORB"])
    got = np.array([d[k]["H"] for k in range(2, len(frames) + 1)])
    assert rc == -1 and (ss == 0).all() and np.allclose(got, Hs, rtol=1e-9, atol=1e-12)
    runtime.reset()


def test_mirror_matches_more_float_rows_than_a_context_holds():
    """KeyPoints.match_kps / match_static_kps on SIFT rows of a 720p frame (more than EVH_MAX_FEATURES = 5984 per frame): the
    float 2-NN and filter take any row count, so the mirror must not size the context by it (the reference has no bound,
    matching.py:75-163); the context is sized by the MATCHES that reach findHomography."""
    from evenvizion_amd import runtime
    from evenvizion_amd._lib import MAX_FEATURES
    from evenvizion_amd.processing import KeyPoints
    runtime.reset()
    prev, cur, _ = S.make_pair(91, 1280, 720)
    o1, o0 = O.sift_detect(cur, cap=65535), O.sift_detect(prev, cap=65535)
    assert len(o1["xy"]) > MAX_FEATURES and len(o0["xy"]) > MAX_FEATURES
    pa, pb = KeyPoints(o1["xy"], o1["desc"]).match_kps(KeyPoints(o0["xy"], o0["desc"]))
    oi, od = O.knn2_f32(o1["desc"], o0["desc"])
    mq, mt = O.ratio_unique_f32(oi, od)
    wa, wb = O.remove_double(o1["xy"][mq], o0["xy"][mt])
    assert len(pa) > 500 and np.array_equal(np.array(pa), wa) and np.array_equal(np.array(pb), wb)
    sa, sb = KeyPoints(o1["xy"], o1["desc"]).match_static_kps(KeyPoints(o0["xy"], o0["desc"]))
    st, ea, eb = O.match_static_f32(o1["xy"], o1["desc"], o0["xy"], o0["desc"])
    assert st == 0 and np.array_equal(sa, ea) and np.array_equal(sb, eb)
    runtime.reset()


def test_a_feature_type_named_twice_is_refused():
    from evenvizion_amd._lib import Context, EvhError
    frames, _ = S.make_stream(5, 3, 400, 224)
    c = Context(device=0, max_w=400, max_h=224, max_features=500, max_frames=4)
    try:
        c.sift_enable(4096)
        H = torch.zeros(2, 9, dtype=torch.float64, device="cuda")
        st = torch.full((2,), -1, dtype=torch.int32, device="cuda")
        with pytest.raises(EvhError):
            c.stream_homography_batch_types(torch.from_numpy(frames).cuda(), H, st, ["SIFT", "ORB", "SIFT"])
    finally:
        c.close()


# ---- SURF (frame_processing.py:65-67) ---------------------------------------------------------------------------------
def make_surf_ctx(w, h, frames=4, surf=4096, sift=0):
    from evenvizion_amd._lib import Context
    c = Context(device=0, max_w=w, max_h=h, max_features=500, max_frames=frames)
    c.surf_enable(surf)
    if sift:
        c.sift_enable(sift)
    return c


def _same_surf(g, o):
    assert len(g["xy"]) == len(o["xy"]), (len(g["xy"]), len(o["xy"]))
    for k in ("xy", "size", "angle", "response", "desc"):
        assert np.array_equal(g[k].view(np.uint32), o[k].view(np.uint32)), k
    assert np.array_equal(g["octave"], o["octave"]) and np.array_equal(g["laplacian"], o["laplacian"])


@pytest.mark.parametrize("w,h,cap", [(400, 224, 4096), (333, 217, 4096), (1280, 720, 16384), (97, 131, 1024)])
def test_surf_keypoints_and_descriptors(w, h, cap):
    """SURF_create(extended=1, hessianThreshold=400).detectAndCompute: the integral image, the key points in the operator's
    order (x, y, size, angle, response, octave, laplacian) and the 128-float descriptors, all BIT FOR BIT."""
    prev, cur, _ = S.make_pair(41, w, h)
    c = make_surf_ctx(w, h, frames=2, surf=cap)
    try:
        c.surf_detect_batch(dev(np.stack([prev, cur])))
        for f, img in enumerate((prev, cur)):
            assert np.array_equal(c.surf_download_integral(f), O.integral(img))
            _same_surf(c.surf_download(f), O.surf_detect(img))
        c.surf_detect_batch(dev(np.stack([S.gray_to_bgr(prev), S.gray_to_bgr(cur)])))
        _same_surf(c.surf_download(1), O.surf_detect(cur))
        flat = np.full((2, h, w), 77, np.uint8)
        c.surf_detect_batch(dev(flat))
        assert len(c.surf_download(0)["xy"]) == 0 and len(O.surf_detect(flat[0])["xy"]) == 0
    finally:
        c.close()


def test_surf_large_scales_and_image_borders():
    """Big blobs: key points of the upper octaves (window sides of several hundred pixels, sampled past the image border,
    the integer-ratio INTER_AREA path among them) and key points whose orientation samples partly fall outside."""
    w, h = 640, 480
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 128.0)
    rng = np.random.default_rng(3)
    for k in range(160):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, 45)
        a = rng.uniform(30, 90) * (1 if k % 2 else -1)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
    img = np.clip(img + rng.normal(0, 1.5, img.shape), 0, 255).astype(np.uint8)
    o = O.surf_detect(img)
    assert (o["octave"] >= 2).sum() >= 30 and o["size"].max() > 200          # 67 key points, window sides up to 632
    c = make_surf_ctx(w, h, frames=2)
    try:
        c.surf_detect_batch(dev(np.stack([img, img[::-1].copy()])))
        _same_surf(c.surf_download(0), o)
        _same_surf(c.surf_download(1), O.surf_detect(img[::-1].copy()))
    finally:
        c.close()


@pytest.mark.parametrize("features", [["SURF"], ["SURF", "SIFT", "ORB"]])
def test_reference_default_type_list_stream_vs_oracle(features):
    """The reference's default FrameProcessing list (frame_processing.py:40) end to end in stream semantics."""
    w, h = 400, 224
    frames, _ = S.make_stream(43, 5, w, h)
    flat = np.full((h, w), 128, np.uint8)
    frames = np.stack([frames[0], frames[1], flat, frames[2], frames[3], frames[4]])
    n = len(frames) - 1
    Ho, so, rc = O.stream_gray_types(frames, features)
    assert rc == -1 and list(so) == [0, 1, 1, 0, 0]
    c = make_surf_ctx(w, h, frames=len(frames), surf=4096, sift=4096)
    try:
        H = torch.zeros(n, 9, dtype=torch.float64, device="cuda")
        st = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        c.stream_homography_batch_types(dev(frames), H, st, features)
        c.synchronize()
        assert np.array_equal(st.cpu().numpy(), so)
        assert np.allclose(H.cpu().numpy().reshape(-1, 3, 3), Ho, rtol=1e-9, atol=1e-12)
    finally:
        c.close()


def test_python_mirror_default_is_the_reference_default():
    """get_homography_dict(capture) / FrameProcessing(frame) with NO feature list = SURF + SIFT + ORB, like the reference
    (video_processing.py:69,74 -> frame_processing.py:40)."""
    from evenvizion_amd import runtime
    from evenvizion_amd.processing import FrameProcessing, get_homography_dict
    runtime.reset()
    w, h = 400, 224
    frames, _ = S.make_stream(47, 4, w, h)
    bgr = [S.gray_to_bgr(f) for f in frames]
    fa = FrameProcessing(bgr[1])
    assert fa.features_types == ["SURF", "SIFT", "ORB"]
    xy, desc = fa.detect_and_describe_features("SURF")
    o = O.surf_detect(frames[1])
    assert desc.dtype == np.float32 and np.array_equal(xy, o["xy"]) and np.array_equal(desc, o["desc"])
    Hs, ss, rc = O.stream_gray_types(frames, ["SURF", "SIFT", "ORB"])
    d = get_homography_dict(S.SyntheticCapture(bgr), resize_width=w, chunk_frames=3)
    got = np.array([d[k]["H"] for k in range(2, len(frames) + 1)])
    assert rc == -1 and (ss == 0).all() and np.allclose(got, Hs, rtol=1e-9, atol=1e-12)
    # per-pair body through the class API
    ca, cb = fa.concatenate_all_features_types(FrameProcessing(bgr[0]))
    from evenvizion_amd.processing import compute_homography
    assert np.allclose(compute_homography(ca, cb, None), Hs[0], rtol=1e-9, atol=1e-12)
    runtime.reset()


def test_byte_matcher_on_128_byte_rows_ties_after_sqrt():
    """evh_match_knn2_l2u8x128 (the fused path's SIFT matcher): exact integer squared distances, but neighbours are ranked
    like the operator ranks them -- by sqrt in float32, where two different D above 2^22 can collide and then the LOWER
    train index wins.  112 saturated, complementary dimensions put D near 7e6; 16 neutral dimensions carry 0..4 unit bumps
    per train row, so D0, D0+1, ... D0+4 compete: in about a third of the queries the float ranking differs from the
    integer one.  The float matcher on the same values is the checker."""
    from evenvizion_amd._lib import Context
    rng = np.random.default_rng(8)
    A = (rng.integers(0, 2, 112) * 255).astype(np.int32)
    nq, nt = 300, 700
    q = np.full((nq, 128), 128, np.int32)
    for i in range(nq):
        row = A.copy(); ii = rng.choice(112, rng.integers(0, 9), replace=False); row[ii] = 255 - row[ii]; q[i, 16:] = row
    t = np.full((nt, 128), 128, np.int32); t[:, 16:] = 255 - A
    for j in range(nt):
        k = (j * 7 + 3) % 5
        pos = rng.choice(16, k, replace=False); t[j, pos] += rng.choice([-1, 1], k)
    q = q.astype(np.uint8); t = t.astype(np.uint8)
    D = ((q[:, None, :].astype(np.int64) - t[None, :, :].astype(np.int64)) ** 2).sum(-1)
    oi, od = O.knn2_f32(q.astype(np.float32), t.astype(np.float32))
    assert D.min() > (1 << 22) and (np.argsort(D, axis=1, kind="stable")[:, :2] != oi).any(axis=1).sum() > 50
    c = Context(device=0, max_w=64, max_h=64, max_features=500, max_frames=2)
    try:
        idx = torch.zeros(nq, 2, dtype=torch.int32, device="cuda")
        d2 = torch.zeros(nq, 2, dtype=torch.int32, device="cuda")
        c.knn2(dev(q), dev(t), idx, d2)
        c.synchronize()
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(np.sqrt(d2.cpu().numpy().astype(np.uint32).astype(np.float32)), od)
    finally:
        c.close()


@pytest.mark.parametrize("nq,nt,kind", [(1, 1, "rand"), (2, 1, "rand"), (1, 2, "rand"), (31, 63, "rand"), (33, 65, "edge"), (511, 700, "rand"),
                                        (513, 129, "dup"), (1500, 3000, "rand"), (700, 2050, "edge"), (129, 64, "dup")])
def test_byte_matcher_on_128_byte_rows_matrix_core_form(nq, nt, kind):
    """k_knn2_mfma128: the Gram matrix of 128-byte rows on v_mfma_i32_32x32x32_i8 with the bias trick (t - 128, 127 - q) must
    give the exact integer squared distances and the operator's ranking -- ragged sizes around the 32 / 64 / 512 tile edges,
    rows of 0s and 255s (the extremes of the signed-byte bias), duplicated train rows (equal distances: the lower index wins,
    also across the two lanes that share a query and across tiles).  Checker: exact int64 distances in numpy, stable argsort
    (all D here are below 2^22, where the float32 sqrt ranking is the integer ranking), and the float matcher of the oracle."""
    from evenvizion_amd._lib import Context
    rng = np.random.default_rng(nq * 7919 + nt)
    if kind == "edge":
        q = rng.choice(np.array([0, 255, 1, 254, 128, 127], np.uint8), (nq, 128))
        t = rng.choice(np.array([0, 255, 1, 254, 128, 127], np.uint8), (nt, 128))
        q[0] = 0; q[-1] = 255; t[0] = 255; t[-1] = 0
        # keep D below 2^22: at most 60 saturated coordinates differ
        t[:, 60:] = 100; q[:, 60:] = 100
    else:
        base = rng.integers(0, 256, (max(nt // 3, 1), 128))
        t = np.clip(base[rng.integers(0, len(base), nt)] // 3 + rng.integers(0, 30, (nt, 128)), 0, 255).astype(np.uint8)
        q = np.clip(base[rng.integers(0, len(base), nq)] // 3 + rng.integers(0, 30, (nq, 128)), 0, 255).astype(np.uint8)
        if kind == "dup":                                # every train row occurs several times, far apart
            t = t[np.arange(nt) % max(nt // 4, 1)]
    D = ((q[:, None, :].astype(np.int64) - t[None, :, :].astype(np.int64)) ** 2).sum(-1)
    assert D.max() < (1 << 22)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    c = Context(device=0, max_w=64, max_h=64, max_features=500, max_frames=2)
    try:
        idx = torch.full((nq, 2), -7, dtype=torch.int32, device="cuda")
        d2 = torch.zeros(nq, 2, dtype=torch.int32, device="cuda")
        c.knn2(dev(q), dev(t), idx, d2)
        c.synchronize()
        gi, gd = idx.cpu().numpy(), d2.cpu().numpy().astype(np.uint32)
    finally:
        c.close()
    assert np.array_equal(gi[:, 0], order[:, 0])
    assert np.array_equal(gd[:, 0], np.take_along_axis(D, order[:, :1], 1)[:, 0].astype(np.uint32))
    if nt >= 2:
        assert np.array_equal(gi[:, 1], order[:, 1])
        assert np.array_equal(gd[:, 1], np.take_along_axis(D, order[:, 1:2], 1)[:, 0].astype(np.uint32))
        oi, od = O.knn2_f32(q.astype(np.float32), t.astype(np.float32))
        assert np.array_equal(gi, oi) and np.array_equal(np.sqrt(gd.astype(np.float32)), od)
    else:
        assert (gi[:, 1] == -1).all() and (gd[:, 1] == 0xFFFFFFFF).all()


def test_detector_slot_overflow_is_a_pair_status_in_the_fused_path():
    """A frame with more SIFT key points than evh_sift_enable reserved fails ITS pairs with EVH_PAIR_CAPACITY (6) in the
    fused multi-type entries -- never a silent truncation -- and only the pairs that touch it."""
    from evenvizion_amd._lib import PAIR_CAPACITY
    w, h = 400, 224
    fr, _ = S.make_stream(53, 4, w, h)
    sparse = np.full((h, w), 100, np.uint8)                 # 65 bright squares: 260 key points, the textured frames have ~2 400
    for y in range(50, 170, 24):
        for x in range(50, 350, 24):
            sparse[y:y + 7, x:x + 7] = 220
    frames = np.stack([sparse, np.roll(sparse, 2, axis=1), fr[0], fr[1]])
    n_sparse = len(O.sift_detect(sparse)["xy"])
    assert 0 < n_sparse < 500 and len(O.sift_detect(fr[0])["xy"]) > 1024
    c = make_ctx(w, h, frames=4, sift=512)
    try:
        H = torch.zeros(3, 9, dtype=torch.float64, device="cuda")
        st = torch.full((3,), -1, dtype=torch.int32, device="cuda")
        c.stream_homography_batch_types(dev(frames), H, st, ["SIFT"])
        c.synchronize()
        got = st.cpu().numpy()
        assert got[1] == PAIR_CAPACITY and got[2] == PAIR_CAPACITY and got[0] == 0
    finally:
        c.close()


def test_type_list_with_fused_resize_at_the_reference_geometry():
    """The reference's example geometry: 1170x658 BGR frames, resize_width=400 -> 400x224 (video_processing.py:62,73), all
    three detectors.  get_homography_dict on the full-size frames (imutils.resize fused into the ingest kernel) equals the
    oracle's resize -> gray -> SURF + SIFT + ORB stream."""
    from evenvizion_amd import runtime
    from evenvizion_amd.processing import get_homography_dict
    from evenvizion_amd.processing.video_processing import resized_shape
    runtime.reset()
    g, _ = S.make_stream(59, 4, 1170, 658)
    full = np.stack([g, np.roll(g, 5, axis=2), 255 - g], axis=-1)          # three different channels
    dw, dh = resized_shape(full.shape[1:], 400)
    assert (dw, dh) == (400, 224)
    gray = np.stack([O.bgr2gray(O.resize_area(f, dw, dh)) for f in full])
    Hs, ss, rc = O.stream_gray_types(gray, ["SURF", "SIFT", "ORB"])
    d = get_homography_dict(S.SyntheticCapture(list(full)), resize_width=400, chunk_frames=3)
    got = np.array([d[k]["H"] for k in range(2, len(full) + 1)])
    assert rc == -1 and (ss == 0).all() and d["resize_info"] == {"h": 224, "w": 400}
    assert np.allclose(got, Hs, rtol=1e-9, atol=1e-12)
    runtime.reset()
