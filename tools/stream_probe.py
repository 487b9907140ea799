import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context
res = {}
for (w, h, nfeat, nfr) in [(1280, 720, 500, 65), (1280, 720, 2000, 65)]:
    frames, _ = S.make_stream(11, 9, w, h)
    frames = np.concatenate([frames] * 8)[:nfr]          # looped content: timing only
    d = torch.from_numpy(frames).cuda()
    ctx = Context(device=0, max_w=w, max_h=h, max_features=nfeat, max_frames=nfr)
    H = torch.zeros(nfr - 1, 9, dtype=torch.float64, device='cuda'); st = torch.zeros(nfr - 1, dtype=torch.int32, device='cuda')
    for force in (False, True):
        ctx.stream_homography_batch(d, H, st, nfeatures=nfeat, force_max_iters=force); ctx.synchronize()
        ctx.profile_enable(True)
        t = time.perf_counter()
        ctx.stream_homography_batch(d, H, st, nfeatures=nfeat, force_max_iters=force); ctx.synchronize()
        dt = time.perf_counter() - t
        prof = ctx.profile_read(); ctx.profile_enable(False)
        res['%dx%d_n%d_force%d' % (w, h, nfeat, force)] = dict(pairs=nfr - 1, seconds=round(dt, 4), pairs_per_s=round((nfr - 1) / dt, 1),
            ok=int((st == 0).sum()), stage_ms={k: round(v[1], 2) for k, v in prof.items()})
    ctx.close()
print(json.dumps(res, indent=1))
