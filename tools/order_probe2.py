"""GPU probe: the chunked stream driver on a pan stream, repeated, to expose run-to-run differences."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context, EvhError
from evenvizion_amd.processing import get_homography_dict
from oracle import oracle as O

w, h, n = 400, 224, int(sys.argv[1]) if len(sys.argv) > 1 else 70
frames = S.make_pan_stream(71, n, w, h, step=6.0)
if n > 310:
    frames[301] = 90; frames[302] = 90
Ho, so, rc = O.stream_gray(frames)
print('oracle status counts', np.bincount(so))
for rep in range(4):
    try:
        d = get_homography_dict(S.SyntheticCapture([S.gray_to_bgr(f) for f in frames]), resize_width=w, chunk_frames=17,
                                features_type_list=["ORB"])
        Hg = np.array([d[k]["H"] for k in range(2, n + 1)])
        print("rep", rep, "ok; max |H - oracle|", np.abs(Hg - Ho).max())
    except EvhError as e:
        print("rep", rep, "error:", str(e)[:150])
# direct detect of all frames, several times: do the flags / counts vary?
c = Context(device=0, max_w=w, max_h=h, max_features=500, max_frames=17)
bgr = torch.from_numpy(np.stack([S.gray_to_bgr(f) for f in frames[:17]])).cuda()
for rep in range(6):
    c.orb_detect_batch(bgr)
    cnt = []
    for f in range(17):
        try:
            cnt.append(len(c.orb_download(f)["xy"]))
        except EvhError as e:
            cnt.append(-1)
    print("rep", rep, cnt)
