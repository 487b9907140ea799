"""GPU probe: ORB key points of a frame set in the reference's order (k_select_cv) against the oracle, frame by frame."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context, EvhError
from oracle import oracle as O

w, h, n = 400, 224, 60
frames = S.make_pan_stream(71, 600, w, h, step=6.0)[:n]
c = Context(device=0, max_w=w, max_h=h, max_features=500, max_frames=n)
c.orb_detect_batch(torch.from_numpy(frames).cuda())
bad = 0
for f in range(n):
    o = O.orb_detect(frames[f])
    try:
        g = c.orb_download(f)
    except EvhError as e:
        print("frame", f, "error:", e, "oracle count", len(o["xy"]), np.bincount(o["octave"], minlength=8))
        bad += 1
        continue
    same = len(g["xy"]) == len(o["xy"]) and all(np.array_equal(g[k], o[k]) for k in ("octave", "lx", "ly"))
    if not same:
        print("frame", f, "differs", len(g["xy"]), len(o["xy"]))
        bad += 1
print("bad", bad, "of", n)
