#!/usr/bin/env python3
"""Capture golden vectors for the reference's Python glue by importing the reference itself.

Runs ONLY in the build container (needs /root/reference); writes tests/golden/glue_*.json which travel with the
repo.  The reference's own code is pure Python; its two third-party imports (cv2, imutils) are absent from the
image and are replaced by inert stub modules here, so only the reference's glue arithmetic executes:
    matching.lowes_ratio_test / filter_corresponding_points     (matching.py:166-239)
    utils.remove_double_matching                                (utils.py:41-68)
    utils.find_point_displacement / get_largest_group_points    (utils.py:258-325)
    utils.compute_homography's pre-transform                    (utils.py:351-358)
    utils.matrix_superposition / superposition_dict             (utils.py:118-145,184-211)
    utils.homography_transformation                             (utils.py:71-92)
    video_processing.get_homography_dict's loop                 (video_processing.py:58-107)
plus the reference's one known-answer artefact (metrics_file.txt).
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        raise RuntimeError("stubbed third-party call: " + self.__name__)


def import_reference():
    cv2 = _Stub("cv2")
    cv2.RANSAC = 8
    sys.modules["cv2"] = cv2
    sys.modules["imutils"] = _Stub("imutils")
    sys.path.insert(0, REF)
    import evenvizion.processing.matching as matching
    import evenvizion.processing.utils as utils
    import evenvizion.processing.video_processing as video_processing
    return matching, utils, video_processing, cv2


class DMatch:
    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.distance = q, t, d


def f32list(a):
    return [[float(v) for v in row] for row in np.asarray(a, np.float32)]


def main():
    matching, utils, vp, cv2 = import_reference()
    rng = np.random.default_rng(20261004)
    out = {}

    # ---- lowes_ratio_test + filter_corresponding_points ------------------------------------------------
    cases = []
    for case in range(12):
        nq, nt = int(rng.integers(1, 40)), int(rng.integers(1, 25))
        idx = np.full((nq, 2), -1, np.int64)
        d2 = np.zeros((nq, 2), np.int64)
        for i in range(nq):
            if nt >= 2:
                t0, t1 = rng.choice(nt, 2, replace=False)
                a = int(rng.integers(0, 4000))
                # force exact-boundary cases 4*D0 == D1 and near misses
                mode = rng.integers(0, 5)
                b = {0: 4 * a, 1: 4 * a + 1, 2: max(4 * a - 1, a), 3: int(rng.integers(a, 40000)), 4: a}[int(mode)]
                idx[i] = (t0, t1); d2[i] = (a, b)
            else:
                idx[i] = (0, -1); d2[i] = (int(rng.integers(0, 100)), 0)
        raw = []
        for i in range(nq):
            row = [DMatch(i, int(idx[i, k]), float(np.sqrt(np.float32(d2[i, k])))) for k in range(2) if idx[i, k] >= 0]
            raw.append(row)
        for ratio in (0.5, 0.7):
            res = matching.lowes_ratio_test(raw, ratio)
            cases.append(dict(idx=idx.tolist(), d2=d2.tolist(), ratio=ratio,
                              matches=[[int(t), int(q)] for (t, q) in res]))
    out["lowes_ratio_test"] = cases

    # ---- remove_double_matching ---------------------------------------------------------------------------
    cases = []
    for case in range(8):
        n = int(rng.integers(1, 30))
        a = rng.integers(0, 6, (n, 2)).astype(np.float32) * np.float32(1.2)
        b = rng.uniform(0, 100, (n, 2)).astype(np.float32)
        na, nb = utils.remove_double_matching(a, b)
        cases.append(dict(a=f32list(a), b=f32list(b), out_a=f32list(na), out_b=f32list(nb)))
    out["remove_double_matching"] = cases

    # ---- find_point_displacement + get_largest_group_points ----------------------------------------------------
    cases = []
    for case in range(10):
        n = int(rng.integers(1, 40))
        H = np.eye(3) + rng.normal(0, [[1e-3, 1e-3, 2.0], [1e-3, 1e-3, 2.0], [1e-6, 1e-6, 0]], (3, 3))
        a = rng.uniform(0, 400, (n, 2)).astype(np.float32)
        proj = (H @ np.c_[a.astype(np.float64), np.ones(n)].T).T
        proj = proj[:, :2] / proj[:, 2:]
        off = rng.choice([0.0, 0.5, 1.5, 2.5, 0.49, 3.0, 1.0], n)[:, None] * np.array([[1.0, 0.0]])
        b = (proj + off + (rng.normal(0, 0.2, (n, 2)) if case % 2 else 0)).astype(np.float32)
        if case == 0:  # exact .5 ties with identity H
            H = np.eye(3); a = np.zeros((5, 2), np.float32)
            b = np.float32([[.5, 0], [1.5, 0], [2.5, 0], [.49, 0], [3, 4]])
        groups = utils.find_point_displacement(H, a, b)
        sa, sb = utils.get_largest_group_points(groups, a, b)
        cases.append(dict(H=H.tolist(), a=f32list(a), b=f32list(b),
                          groups={str(int(k)): [int(i) for i in v] for k, v in groups.items()},
                          group_order=[int(k) for k in groups.keys()],
                          out_a=f32list(sa), out_b=f32list(sb)))
    out["static_filter"] = cases

    # ---- compute_homography: pre-transform + gate (cv2.findHomography replaced by a recorder) --------------------
    cases = []
    rec = {}

    def fake_find_homography(pa, pb, method, thr):
        rec["a"] = np.array(pa); rec["b"] = np.array(pb); rec["method"] = method; rec["thr"] = thr
        return rec["ret_H"], rec["ret_mask"]

    cv2.findHomography = fake_find_homography
    for case in range(8):
        n = int(rng.integers(4, 20))
        a = [np.float32(v) for v in rng.uniform(0, 400, (n, 2))]
        b = [np.float32(v) for v in rng.uniform(0, 400, (n, 2))]
        Hs = None if case % 3 == 0 else np.eye(3) + rng.normal(0, [[1e-2, 1e-2, 5.0], [1e-2, 1e-2, 5.0], [1e-5, 1e-5, 0]], (3, 3))
        k = int(rng.integers(0, n + 1))
        mask = np.zeros((n, 1), np.uint8); mask[:k] = 1
        if case == 5:  # boundary of the 0.7 gate: exactly 70 %
            n = 10; a = a[:4] * 3; b = b[:4] * 3; a = a[:10]; b = b[:10]
            mask = np.zeros((10, 1), np.uint8); mask[:7] = 1
        rec["ret_H"] = np.eye(3) * 2.0
        rec["ret_mask"] = mask
        raised = None
        try:
            utils.compute_homography(a, b, Hs)
        except utils.HomographyException as e:
            raised = str(e)
        cases.append(dict(a=f32list(a), b=f32list(b), Hsup=None if Hs is None else Hs.tolist(),
                          mask=mask.ravel().tolist(), passed_a=rec["a"].astype(np.float64).tolist(),
                          passed_b=rec["b"].astype(np.float64).tolist(), passed_dtype=str(rec["a"].dtype),
                          thr=rec["thr"], method=rec["method"], raised=raised))
    out["compute_homography"] = cases

    # ---- matrix_superposition / homography_transformation ----------------------------------------------------
    cases = []
    for case in range(6):
        H = np.eye(3) + rng.normal(0, 0.01, (3, 3)); S = np.eye(3) + rng.normal(0, 0.01, (3, 3))
        v = rng.uniform(0, 400, 2).astype(np.float32)
        cases.append(dict(H=H.tolist(), S=S.tolist(),
                          sup_false=np.asarray(utils.matrix_superposition(H, S, False)).tolist(),
                          sup_true=np.asarray(utils.matrix_superposition(H, S, True)).tolist(),
                          v=[float(v[0]), float(v[1])],
                          hv=[float(x) for x in utils.homography_transformation(v, H)]))
    out["matrix_superposition"] = cases

    # ---- driver loop (video_processing.get_homography_dict) with scripted operators ------------------------------
    class FakeFP:
        def __init__(self, frame, features_type_list=None):
            self.frame = frame

        def concatenate_all_features_types(self, other):
            return ("a", int(self.frame[0, 0, 0])), ("b", int(other.frame[0, 0, 0]))

    script = {}

    def fake_compute(pa, pb, Hsup):
        k = pa[1]
        script.setdefault("calls", []).append(dict(cur=pa[1], prev=pb[1], Hsup=None if Hsup is None else np.asarray(Hsup).tolist()))
        r = script["plan"][k]
        if r == "nomatch":
            raise matching.NoMatchesException("scripted")
        if r == "nohomo":
            raise utils.HomographyException("scripted")
        return np.array(r, np.float64)

    class FakeImutils:
        @staticmethod
        def resize(img, width=None):
            return img[:, :width]

    vp.imutils = FakeImutils
    vp.FrameProcessing = FakeFP
    vp.compute_homography = fake_compute

    class Cap:
        def __init__(self, n):
            self.n, self.i = n, 0

        def read(self):
            if self.i >= self.n:
                return False, None
            f = np.full((6, 12, 3), self.i, np.uint8)
            self.i += 1
            return True, f

    loops = []
    for plan_id, fails in enumerate([{}, {3: "nomatch"}, {2: "nohomo", 3: "nomatch"}, {5: "nohomo"}]):
        nframes = 6
        plan = {}
        for k in range(1, nframes):
            plan[k] = fails.get(k) or (np.eye(3) + rng.normal(0, [[1e-2, 1e-2, 3.0], [1e-2, 1e-2, 3.0], [1e-5, 1e-5, 0]], (3, 3))).tolist()
        script.clear(); script["plan"] = plan
        res = vp.get_homography_dict(Cap(nframes), resize_width=8)
        loops.append(dict(nframes=nframes, resize_width=8, plan={str(k): v for k, v in plan.items()},
                          result={str(k): v for k, v in res.items()}, calls=script["calls"]))
    # a failing FIRST pair: the reference crashes (None.tolist()) -- record the exception type
    script.clear(); script["plan"] = {1: "nomatch", 2: np.eye(3).tolist()}
    try:
        vp.get_homography_dict(Cap(3), resize_width=8)
        first_fail = None
    except Exception as e:  # noqa
        first_fail = type(e).__name__
    out["driver_loop"] = dict(cases=loops, first_pair_failure_exception=first_fail)

    # ---- known-answer test F12: superposition_dict + per-pixel transform of the committed golden H JSON --------------
    gpath = os.path.join(REF, "evenvizion/examples/test_video_processing/test_video/dict_with_homography_matrix.json")
    hd, resize_info = utils.read_homography_dict(gpath)
    sup = utils.superposition_dict(hd)
    h, w = resize_info["h"], resize_info["w"]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    keys = list(sup.keys())
    maxima = []
    for k in keys:
        Hk = np.asarray(sup[k], np.float64)
        d = Hk[2, 0] * xs + Hk[2, 1] * ys + Hk[2, 2]
        u = (Hk[0, 0] * xs + Hk[0, 1] * ys + Hk[0, 2]) / d
        v = (Hk[1, 0] * xs + Hk[1, 1] * ys + Hk[1, 2]) / d
        maxima.append(float(max(u.max(), v.max())))
    metric_txt = open(os.path.join(REF, "evenvizion/examples/test_video_processing/test_video/metrics_file.txt")).read()
    out["kat_f12"] = dict(metrics_file=metric_txt.strip(), n_matrices=len(hd), resize_info=resize_info,
                          max_excluding_last=float(np.max(maxima[:-1])), max_including_last=float(np.max(maxima)),
                          sup_last=np.asarray(sup[keys[-1]]).tolist(), sup_keys=[int(k) for k in keys[:3]] + [int(keys[-1])])

    # ---- N1: fixed_coordinate_system on the reference's own example files -------------------------------------------------
    import evenvizion.processing.fixed_coordinate_system as fcs
    oc = utils.read_json_with_coordinates(os.path.join(REF, "evenvizion/examples/test_video/original_coordinates.json"))
    frames = [1, 2, 60, 121]
    sub = {k: oc[k] for k in frames}
    fx = fcs.from_original_to_fix(sub, sup, [658, 1170], [224, 400])
    back = fcs.from_fix_to_original(fx, sup, [658, 1170], [224, 400])
    tofloat = lambda d: {str(k): [{kk: float(vv) for kk, vv in r.items()} for r in v] for k, v in d.items()}
    out["fixed_coordinates"] = dict(frames=frames, original=tofloat(sub), fixed=tofloat(fx), back=tofloat(back),
                                    original_shape=[658, 1170], resize_shape=[224, 400])

    with open(os.path.join(HERE, "glue_goldens.json"), "w") as f:
        json.dump(out, f)
    print("wrote glue_goldens.json", {k: (len(v) if isinstance(v, list) else "dict") for k, v in out.items()})
    print("KAT:", out["kat_f12"]["metrics_file"], out["kat_f12"]["max_excluding_last"])


if __name__ == "__main__":
    main()
