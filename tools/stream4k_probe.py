"""BASELINE config 5 shape on one GPU: one 3840x2160 stream, ORB 4000, stream semantics (chunks of 17 frames)."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from evenvizion_amd import synthetic as S
from evenvizion_amd._lib import Context
w, h, nfeat, F = 3840, 2160, 4000, 17
a, b, _ = S.make_pair(3, w, h)
frames = np.stack([a, b] * 9)[:F]                 # alternating views of one scene: timing only
d = torch.from_numpy(frames).cuda()
ctx = Context(device=0, max_w=w, max_h=h, max_features=nfeat, max_frames=F)
H = torch.zeros(F - 1, 9, dtype=torch.float64, device='cuda'); st = torch.zeros(F - 1, dtype=torch.int32, device='cuda')
state = torch.zeros(18, dtype=torch.float64, device='cuda')
ctx.stream_homography_batch(d, H, st, state_out=state, nfeatures=nfeat); ctx.synchronize()
ctx.profile_enable(True)
t = time.perf_counter()
reps = 3
for _ in range(reps):
    ctx.stream_homography_batch(d, H, st, state_in=state, state_out=state, nfeatures=nfeat)
ctx.synchronize()
dt = (time.perf_counter() - t) / reps
prof = ctx.profile_read()
print(json.dumps(dict(pairs=F - 1, ms=round(dt * 1e3, 1), pairs_per_s=round((F - 1) / dt, 1), ok=int((st == 0).sum()),
                      stage_ms={k: round(v[1] / max(v[0], 1), 2) for k, v in prof.items()})))
