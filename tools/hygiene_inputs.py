"""Inputs of the CPU-side hygiene runs (tools/hygiene.sh): everything the parity tests feed the oracle's SIFT / SURF / ORB
statements, plus frames chosen to reach their rarely taken exits, plus the capture source on intact and damaged files.
Prints nothing but a summary line; the sanitizers / gcov do the reporting."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from evenvizion_amd import capture, synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

MP4 = os.path.join(ROOT, "tests", "golden", "ref_test_video.mp4")


def detector_inputs():
    rng = np.random.default_rng(4)
    imgs = []
    for seed, (w, h) in ((13, (400, 224)), (7, (333, 217)), (3, (97, 131)), (5, (64, 64))):
        imgs.append(S.make_pair(seed, w, h)[0])
    imgs.append(S.make_pair(2, 1280, 720)[0][:360, :640].copy())
    imgs.append(np.full((224, 400), 90, np.uint8))                                  # flat: no key points at all
    imgs.append(rng.integers(0, 256, (120, 160), dtype=np.uint8))                   # white noise: unstable extrema, many rejects
    imgs.append((rng.integers(0, 2, (120, 160)) * 255).astype(np.uint8))            # binary noise: saturated gradients
    yy, xx = np.mgrid[0:200, 0:300]
    blobs = np.zeros((200, 300), np.float64)
    for cx, cy, s in ((40, 40, 3), (100, 60, 6), (180, 90, 12), (250, 150, 25), (6, 6, 4), (295, 195, 5), (150, 196, 9)):
        blobs += 200 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))     # blobs of many scales, some at the border
    imgs.append(np.clip(blobs, 0, 255).astype(np.uint8))
    ramp = (xx * 255 // 299).astype(np.uint8)                                       # pure gradient: edge responses, no extrema
    imgs.append(ramp)
    imgs.append(np.kron(rng.integers(0, 2, (25, 38)) * 255, np.ones((8, 8))).astype(np.uint8)[:200, :300])   # checker blocks
    big = np.zeros((300, 300), np.uint8); big[60:240, 60:240] = 255                 # one huge square: SURF's largest filters
    imgs.append(big)
    # near-flat noise: DoG values of a few units, Hessians close to singular -> huge / non-converging refinement offsets
    for k in range(6):
        imgs.append((127 + rng.integers(-1, 2, (72, 88))).astype(np.uint8))
    # one large blob in a small frame: SURF key points whose orientation / descriptor window is larger than the image
    yy2, xx2 = np.mgrid[0:72, 0:72]
    for s_ in (14, 20):
        imgs.append(np.clip(220 * np.exp(-((xx2 - 36) ** 2 + (yy2 - 36) ** 2) / (2.0 * s_ * s_)), 0, 255).astype(np.uint8))
    return imgs


def main():
    n_kp = 0
    for g in detector_inputs():
        n_kp += len(O.sift_detect(g, cap=65535)["xy"]) + len(O.surf_detect(g)["xy"]) + len(O.orb_detect(g)["xy"])
    frames = capture.read_all(MP4)
    gray = np.stack([O.bgr2gray(O.resize_area(f, 400, 224)) for f in frames[:14]])
    for feats in (["SURF", "SIFT", "ORB"], ["ORB"]):
        O.stream_gray_types(gray, feats)
    O.stream_gray_types(gray[:5], ["SIFT", "ORB"], force_max_iters=True)
    fs, _ = S.make_stream(37, 5, 400, 224)
    fs[2] = 90                                                                      # a flat frame: failure statuses
    O.stream_gray_types(fs, ["SIFT", "ORB"])
    # ---- the capture source on damaged files: every outcome must be a clean error or a clean end, never a memory fault
    data = bytearray(open(MP4, "rb").read())
    rng = np.random.default_rng(9)
    outcomes = {"ok": 0, "open_error": 0, "read_error": 0}
    for trial in range(int(os.environ.get("HYGIENE_FUZZ", "40"))):
        d = bytearray(data)
        kind = trial % 4
        if kind == 0:                                  # bit flips inside the coded pictures
            for _ in range(20):
                d[int(rng.integers(48, 1310000))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                # bit flips inside the sample tables / parameter sets
            for _ in range(6):
                d[int(rng.integers(1310767, len(d)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:                                # truncation
            d = d[:int(rng.integers(100, len(d)))]
        else:                                          # a zeroed span
            a = int(rng.integers(48, 1300000)); d[a:a + int(rng.integers(1, 5000))] = bytes(1)
        cap = capture.VideoCapture(data=bytes(d))
        if not cap.isOpened():
            outcomes["open_error"] += 1
            continue
        try:
            n = 0
            while cap.read()[0] and n < 200:
                n += 1
            outcomes["ok"] += 1
        except capture.CaptureError:
            outcomes["read_error"] += 1
        cap.release()
    print("hygiene inputs done: %d key points, capture outcomes on damaged files %s" % (n_kp, outcomes))


if __name__ == "__main__":
    main()
