"""Multi-type stream probe: one synthetic stream through evh_stream_homography_batch_types (the reference's default
FrameProcessing list SURF, SIFT, ORB and its sub-lists), 32 pairs per call.
usage: python tools/types_probe.py [WxH ...]   (default 400x224 = the reference's default resize_width)
EVH_PROBE_SOLVER=fast: LM's systems by LDL^T (evh_set_solver_mode)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context
res = {}
for wh in (sys.argv[1:] or ["400x224"]):
    w, h = map(int, wh.split("x"))
    nfr = 33
    frames, _ = S.make_stream(11, 9, w, h)
    frames = np.concatenate([frames] * 4)[:nfr]
    d = torch.from_numpy(S.gray_to_bgr(frames)).cuda()
    ctx = Context(device=0, max_w=w, max_h=h, max_features=500, max_frames=nfr)
    ctx.sift_enable(6144 if w <= 400 else 65535); ctx.surf_enable(4096 if w <= 400 else 24576)
    ctx.set_solver_mode(1 if os.environ.get("EVH_PROBE_SOLVER") == "fast" else 0)
    for feats in (["ORB"], ["SIFT"], ["SURF"], ["SURF", "SIFT", "ORB"]):
        H = torch.zeros(nfr - 1, 9, dtype=torch.float64, device='cuda'); st = torch.zeros(nfr - 1, dtype=torch.int32, device='cuda')
        ctx.stream_homography_batch_types(d, H, st, feats); ctx.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            ctx.stream_homography_batch_types(d, H, st, feats)
        ctx.synchronize()
        dt = (time.perf_counter() - t) / 3
        res['%dx%d_%s' % (w, h, "+".join(feats))] = dict(pairs=nfr - 1, seconds=round(dt, 4), pairs_per_s=round((nfr - 1) / dt, 1),
                                                         ok=int((st == 0).sum()))
    # key points per frame of each type (frame 1 of the last call): the sizes behind the matcher / scan times
    res['%dx%d_keypoints_frame1' % (w, h)] = dict(SIFT=int(ctx.lib.evh_sift_count(ctx.h, 1)), SURF=int(ctx.lib.evh_surf_count(ctx.h, 1)))
    ctx.close()
print(json.dumps(res, indent=1))
