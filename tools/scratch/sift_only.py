import os, sys, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context
w, h, nfr = 1280, 720, 33
frames, _ = S.make_stream(11, 9, w, h)
frames = np.concatenate([frames] * 4)[:nfr]
d = torch.from_numpy(S.gray_to_bgr(frames)).cuda()
ctx = Context(device=0, max_w=w, max_h=h, max_features=500, max_frames=nfr)
ctx.sift_enable(65535)
H = torch.zeros(nfr - 1, 9, dtype=torch.float64, device='cuda'); st = torch.zeros(nfr - 1, dtype=torch.int32, device='cuda')
for _ in range(3):
    ctx.stream_homography_batch_types(d, H, st, ["SIFT"]); ctx.synchronize()
