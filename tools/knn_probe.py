"""2-NN micro-probe: one call of evh_match_knn2_l2u8x128 on nq x nt rows of 128 bytes (clustered random data), ms per call.
usage: python tools/knn_probe.py [nq nt]   (EVHIP_LIBRARY / EVH_KNN_DOT4 select the build / the v_dot4 kernel)"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from evenvizion_amd._lib import Context
nq, nt = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (131072, 25600)
rng = np.random.default_rng(1)
base = rng.integers(0, 256, (4096, 128))
t = np.clip(base[rng.integers(0, 4096, nt)] // 2 + rng.integers(0, 60, (nt, 128)), 0, 255).astype(np.uint8)
q = np.clip(base[rng.integers(0, 4096, nq)] // 2 + rng.integers(0, 60, (nq, 128)), 0, 255).astype(np.uint8)
c = Context(device=0, max_w=64, max_h=64, max_features=500, max_frames=2)
dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
idx = torch.zeros(nq, 2, dtype=torch.int32, device="cuda"); d2 = torch.zeros(nq, 2, dtype=torch.int32, device="cuda")
c.knn2(dq, dt, idx, d2); c.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    c.knn2(dq, dt, idx, d2)
c.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps({"nq": nq, "nt": nt, "ms": round(ms, 3), "Tcombos_per_s": round(nq * nt / ms / 1e9, 3),
                  "cycles_per_32x32_block_per_simd": round(ms * 1e-3 * 2.4e9 * 1024 / (nq / 32 * ((nt + 31) // 32)), 1),
                  "checksum": int(idx.sum().item())}))
