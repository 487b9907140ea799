"""Stream-mode probe: one synthetic stream through evh_stream_homography_batch, 64 pairs per call.
usage: python tools/stream_probe.py [WxH:nfeatures:force ...]   (default: the four 720p cases)
EVH_RANSAC_PROF=1 adds the in-kernel cycle accounting of the scan kernel on stderr."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context
cases = sys.argv[1:] or ["1280x720:500:0", "1280x720:500:1", "1280x720:2000:0", "1280x720:2000:1"]
res = {}
for case in cases:
    wh, nfeat, force = case.split(":")
    w, h = map(int, wh.split("x")); nfeat = int(nfeat); force = bool(int(force))
    nfr = 65 if w <= 1920 else 17
    frames, _ = S.make_stream(11, 9, w, h)
    frames = np.concatenate([frames] * 8)[:nfr]          # looped content: timing only
    d = torch.from_numpy(frames).cuda()
    ctx = Context(device=0, max_w=w, max_h=h, max_features=nfeat, max_frames=nfr)
    H = torch.zeros(nfr - 1, 9, dtype=torch.float64, device='cuda'); st = torch.zeros(nfr - 1, dtype=torch.int32, device='cuda')
    ctx.stream_homography_batch(d, H, st, nfeatures=nfeat, force_max_iters=force); ctx.synchronize()
    ctx.profile_enable(True)
    t = time.perf_counter()
    ctx.stream_homography_batch(d, H, st, nfeatures=nfeat, force_max_iters=force); ctx.synchronize()
    dt = time.perf_counter() - t
    prof = ctx.profile_read(); ctx.profile_enable(False)
    res['%dx%d_n%d_force%d' % (w, h, nfeat, force)] = dict(pairs=nfr - 1, seconds=round(dt, 4), pairs_per_s=round((nfr - 1) / dt, 1),
        ok=int((st == 0).sum()), ransac_final_ms_per_pair=round(prof['ransac_final'][1] / (nfr - 1), 4),
        stage_ms={k: round(v[1], 2) for k, v in prof.items()})
    ctx.close()
print(json.dumps(res, indent=1))
