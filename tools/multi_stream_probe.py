"""Throughput of S concurrent 720p streams on one GPU (evh_multi_stream_homography_batch): pairs/s vs S."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context
w, h, F = 1280, 720, 17
base, _ = S.make_stream(11, F, w, h)
res = {}
for ns in (1, 8, 32, 60):
    frames = np.stack([base] * ns)                      # same content in every stream: timing only
    d = torch.from_numpy(frames).cuda()
    ctx = Context(device=0, max_w=w, max_h=h, max_features=500, max_frames=ns * F)
    ctx.set_async_solve(True)
    H = torch.zeros(ns, F - 1, 9, dtype=torch.float64, device='cuda'); st = torch.zeros(ns, F - 1, dtype=torch.int32, device='cuda')
    state = torch.zeros(ns, 18, dtype=torch.float64, device='cuda')
    ctx.multi_stream_homography_batch(d, H, st, state_out=state); ctx.synchronize()
    reps = 4
    t = time.perf_counter()
    for _ in range(reps):
        ctx.multi_stream_homography_batch(d, H, st, state_in=state, state_out=state)
    ctx.synchronize()
    dt = (time.perf_counter() - t) / reps
    res[ns] = dict(streams=ns, pairs=ns * (F - 1), ms=round(dt * 1e3, 2), pairs_per_s=round(ns * (F - 1) / dt, 1), ok=int((st == 0).sum()))
    ctx.close()
print(json.dumps(res, indent=1))
