"""Stage 1 of tools/make_skimage_fixture.py (runs under the repo's own interpreter): the oracle's ORB key points of one
synthetic frame with the level images they were computed on -> an .npz for the scikit-image stage."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle as O
from evenvizion_amd import synthetic as S

_, gray, _ = S.make_pair(12, 320, 240)
kp = O.orb_detect(gray, 500)
pyr = O.orb_pyramid(gray)
out = {"gray": gray, "octave": kp["octave"], "lx": kp["lx"], "ly": kp["ly"], "angle": kp["angle"], "desc": kp["desc"]}
for l in range(8):
    out["level%d" % l] = pyr[l]
    out["blur%d" % l] = O.gaussian_blur7(pyr[l])
np.savez(sys.argv[1], **out)
print("oracle: %d key points" % len(kp["lx"]))
