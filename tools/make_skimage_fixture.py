"""Independent check of the FAST-9/16 corner predicate (frame_processing.py:59-61 -> cv2.ORB -> FAST): the corner mask of
skimage.feature.corner_fast(n=9) on three small images, committed as tests/golden/skimage_fast9.npz (images + packed masks).

Run in the BUILD container only, under the interpreter that has scikit-image (0.18.3 there; no SIFT / SURF in it):
    /opt/conda/bin/python3.9 tools/make_skimage_fixture.py
skimage's corner_fast is its own Cython implementation of Rosten's segment test -- not OpenCV, not this repo's code.
Threshold: the operator's `I > centre + 20` on integers is `> 20.5 / 255` on skimage's [0, 1] floats (no equality cases)."""
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
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "skimage_fast9.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
