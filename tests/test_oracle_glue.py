"""Oracle (oracle/) vs the golden vectors captured from the reference's own Python glue
(tests/golden/make_glue_goldens.py) and the reference's known-answer artefact (metrics_file.txt)."""
import json
import os

import numpy as np

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_lowes_ratio_and_unique_filter(goldens):
    for c in goldens["lowes_ratio_test"]:
        q, t = O.ratio_unique(np.array(c["idx"]), np.array(c["d2"]), c["ratio"])
        got = [[int(a), int(b)] for a, b in zip(t, q)]   # reference tuples are (trainIdx, queryIdx)
        assert got == c["matches"]


def test_ratio_half_equals_integer_test():
    # d0 < 0.5*d1 on f32 square roots  <=>  4*D0 < D1 in integers
    rng = np.random.default_rng(1)
    D0 = rng.integers(0, 520200, 20000); D1 = np.maximum(D0, rng.integers(0, 2080800, 20000))
    D1[:5000] = 4 * D0[:5000] + rng.integers(-1, 2, 5000)
    D1 = np.clip(D1, D0, 2080800)
    idx = np.tile(np.array([[0, 1]]), (len(D0), 1)); idx[:, 0] = np.arange(len(D0))  # all trains distinct
    q, _ = O.ratio_unique(idx, np.stack([D0, D1], 1), 0.5)
    want = np.nonzero(4 * D0 < D1)[0]
    assert np.array_equal(q, want)


def test_remove_double_matching(goldens):
    for c in goldens["remove_double_matching"]:
        a, b = O.remove_double(c["a"], c["b"])
        assert np.array_equal(a, np.float32(c["out_a"]).reshape(-1, 2))
        assert np.array_equal(b, np.float32(c["out_b"]).reshape(-1, 2))


def test_static_filter(goldens):
    for c in goldens["static_filter"]:
        a, b = O.static_filter(c["H"], c["a"], c["b"])
        assert np.array_equal(a, np.float32(c["out_a"]).reshape(-1, 2))
        assert np.array_equal(b, np.float32(c["out_b"]).reshape(-1, 2))


def test_matrix_superposition(goldens):
    for c in goldens["matrix_superposition"]:
        assert np.array_equal(O.matrix_superposition(c["H"], c["S"], False), np.array(c["sup_false"]))
        assert np.array_equal(O.matrix_superposition(c["H"], c["S"], True), np.array(c["sup_true"]))


def test_kat_f12_metrics_file(goldens):
    """The reference's only numeric artefact: metrics_file.txt == max fixed-plane coordinate over the example
    video, recomputed here from the committed golden H JSON through the oracle's matrix_superposition."""
    want = float(open(os.path.join(GOLD, "ref_metrics_file.txt")).read().split(":")[1])
    assert want == 863.0428982580879
    d = json.load(open(os.path.join(GOLD, "ref_dict_with_homography_matrix.json")))
    ri = d.pop("resize_info")
    assert ri == {"h": 224, "w": 400} and len(d) == 120
    ys, xs = np.mgrid[0:ri["h"], 0:ri["w"]].astype(np.float64)
    sup, first, maxima = None, True, [float(max(xs.max(), ys.max()))]   # frame 1: identity
    for k in sorted(d, key=int):
        sup = O.matrix_superposition(np.array(d[k]["H"]), sup, first)
        first = False
        den = sup[2, 0] * xs + sup[2, 1] * ys + sup[2, 2]
        u = (sup[0, 0] * xs + sup[0, 1] * ys + sup[0, 2]) / den
        v = (sup[1, 0] * xs + sup[1, 1] * ys + sup[1, 2]) / den
        maxima.append(float(max(u.max(), v.max())))
    assert np.array_equal(sup, np.array(goldens["kat_f12"]["sup_last"]))
    # the reference skips the append for the last frame (processing_visualization.py:414-418)
    assert max(maxima[:-1]) == want
    assert max(maxima) == goldens["kat_f12"]["max_including_last"]


def test_compute_homography_pretransform_and_gate(goldens, monkeypatch):
    for c in goldens["compute_homography"]:
        a = np.float32(c["a"]); b = np.float32(c["b"])
        if c["Hsup"] is not None:
            Hs = np.array(c["Hsup"])
            # the oracle pre-transforms in f64 and hands f32 to findHomography (cv2 converts to CV_32F)
            pa = np.float32(c["passed_a"]); pb = np.float32(c["passed_b"])
            st, H = O.compute_homography(a, b, Hs)
            st2, H2 = O.compute_homography(pa, pb, None)
            assert st == st2 and np.array_equal(H, H2)
        else:
            assert np.array_equal(np.float32(c["passed_a"]), a)
        assert c["thr"] == 3.0 and c["method"] == 8
