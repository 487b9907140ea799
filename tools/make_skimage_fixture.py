"""Independent checks of ORB's building blocks against scikit-image (two fixtures under tests/golden/).

1. skimage_fast9.npz -- the FAST-9/16 corner predicate (frame_processing.py:59-61 -> cv2.ORB -> FAST): the corner mask of
skimage.feature.corner_fast(n=9) on three small images, committed as tests/golden/skimage_fast9.npz (images + packed masks).

Run in the BUILD container only, under the interpreter that has scikit-image (0.18.3 there; no SIFT / SURF in it):
    /opt/conda/bin/python3.9 tools/make_skimage_fixture.py
skimage's corner_fast is its own Cython implementation of Rosten's segment test -- not OpenCV, not this repo's code.
Threshold: the operator's `I > centre + 20` on integers is `> 20.5 / 255` on skimage's [0, 1] floats (no equality cases).

2. skimage_orb.npz -- orientation and steered BRIEF of the oracle's own ORB key points of one 320x240 frame: stage 1
(tools/skimage_fixture_inputs.py, the repo's interpreter) writes the key points (level, x, y, angle) and the level images;
this script then asks scikit-image for (a) corner_orientations(level image, key points, OFAST_MASK) -- the intensity-
centroid angle over the same circular 31-pixel patch, exact atan2 -- and (b) orb_cy._orb_loop(blurred level image, key
points, angles) -- skimage's Cython loop over ITS copy of the published 256-test pattern (orb_descriptor_positions.txt),
rotation, rounding and comparison as in Rublee et al.  Stored: the frame, the key points as used, skimage's angles and
bits, and skimage's pattern table.  tests/test_oracle_ops.py::test_orb_orientation_and_brief_equal_skimage recomputes the
oracle on the stored frame and compares: same key points, angles within fastAtan2's error, every descriptor bit equal,
the same 1024 pattern numbers."""
import os
import numpy as np
from skimage.feature import corner_fast
import skimage

rng = np.random.RandomState(20261004)
images = []
# 1: piecewise-constant blocks of several sizes + mild noise (many true corners, plateaus, ties)
a = np.kron(rng.randint(0, 256, (20, 25)), np.ones((8, 8)))
a = 0.6 * a + 0.4 * np.kron(rng.randint(0, 256, (40, 50)), np.ones((4, 4)))
images.append(np.clip(a + rng.normal(0, 3, a.shape), 0, 255).astype(np.uint8))
# 2: smooth texture (sums of sinusoids) with salt-and-pepper points and thin lines
y, x = np.mgrid[0:160, 0:200]
b = 128 + 50 * np.sin(x / 7.0) * np.cos(y / 5.0) + 40 * np.sin((x + 2 * y) / 11.0)
b[rng.randint(0, 160, 300), rng.randint(0, 200, 300)] = 255
b[rng.randint(0, 160, 300), rng.randint(0, 200, 300)] = 0
b[::23, :] = 20; b[:, ::31] = 235
images.append(np.clip(b, 0, 255).astype(np.uint8))
# 3: pure noise at three contrasts (differences right at the threshold occur often)
c = np.concatenate([rng.randint(100, 142, (160, 70)), rng.randint(64, 192, (160, 70)), rng.randint(0, 256, (160, 60))], axis=1)
images.append(c.astype(np.uint8))

out = {}
for i, img in enumerate(images):
    resp = corner_fast(img.astype(np.float64) / 255.0, n=9, threshold=20.5 / 255.0)
    mask = resp > 0
    out["img%d" % i] = img
    out["mask%d" % i] = np.packbits(mask)
    print("image %d: %dx%d, %d corner pixels" % (i, img.shape[1], img.shape[0], int(mask.sum())))
out["skimage_version"] = np.array(skimage.__version__)
here = os.path.dirname(os.path.abspath(__file__))
dst = os.path.join(here, "..", "tests", "golden", "skimage_fast9.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")

# ---- 2. orientation + steered BRIEF ----------------------------------------------------------------------------------
import subprocess, tempfile
from skimage.feature import corner_orientations
from skimage.feature.orb import OFAST_MASK
from skimage.feature.orb_cy import _orb_loop
from skimage.feature._orb_descriptor_positions import POS
tmp = os.path.join(tempfile.mkdtemp(), "orb_in.npz")
subprocess.run(["/usr/bin/python3", os.path.join(here, "skimage_fixture_inputs.py"), tmp], check=True)
d = np.load(tmp)
order, angles, bits = [], [], []
for l in range(8):
    idx = np.nonzero(d["octave"] == l)[0]
    if not len(idx):
        continue
    kp = np.ascontiguousarray(np.c_[d["ly"][idx], d["lx"][idx]].astype(np.intp))
    ang = corner_orientations(d["level%d" % l].astype(np.float64), kp, OFAST_MASK)
    # the descriptor is steered by the OPERATOR's angle (fastAtan2, degrees, float32): that is what the oracle used
    steer = np.ascontiguousarray(np.deg2rad(d["angle"][idx].astype(np.float64)))
    b = _orb_loop(d["blur%d" % l].astype(np.float64), kp, steer)
    order.append(idx); angles.append(ang); bits.append(np.packbits(b.astype(np.uint8), axis=1, bitorder="little"))
order = np.concatenate(order)
fix = {"gray": d["gray"], "octave": d["octave"][order], "lx": d["lx"][order], "ly": d["ly"][order],
       "skimage_angle_rad": np.concatenate(angles), "skimage_desc": np.concatenate(bits),
       "skimage_pattern": POS.astype(np.int8), "skimage_version": np.array(skimage.__version__)}
dst = os.path.join(here, "..", "tests", "golden", "skimage_orb.npz")
np.savez_compressed(dst, **fix)
print("wrote", dst, os.path.getsize(dst), "bytes;", len(order), "key points")
