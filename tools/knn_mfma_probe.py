"""Measurement for VERDICT r2 item 1: the float 2-NN with its distances in Gram form on the f32-input matrix cores
(tools/ubench/knn_mfma.hip) against the exact-order VALU matcher the library ships (k_knn2_f32), same descriptors.
Build the measurement kernel first (in the build container; the .so travels with gpurun):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o tools/ubench/knn_mfma.so tools/ubench/knn_mfma.hip
usage: python tools/knn_mfma_probe.py [N ...]     (default 749 4096 9640: SURF key points per 400x224 / - / 1280x720 frame)
Prints per N: time per pair of both forms, and how many queries the Gram form answers differently (nearest index,
second index, Lowe ratio decision at 0.5) -- it rounds differently from hal::normL2Sqr_'s order."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from evenvizion_amd._lib import Context

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "knn_mfma.so"))
lib.knn_mfma_run.restype = C.c_int
lib.knn_mfma_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int, C.POINTER(C.c_float), C.c_void_p]
res = {}
ctx = Context(device=0, max_w=640, max_h=480, max_features=500, max_frames=4)
for N in [int(a) for a in sys.argv[1:]] or [749, 4096, 9640]:
    rng = np.random.default_rng(N)
    # SURF-like rows: unit length, clustered (a train row near every query) so that nearest neighbours are meaningful
    t = rng.normal(0, 1, (N, 128)).astype(np.float32); t /= np.linalg.norm(t, axis=1, keepdims=True)
    q = t[rng.permutation(N)] + rng.normal(0, 0.05, (N, 128)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    npairs = max(1, min(32, (1 << 26) // (N * N)))                  # several pairs per launch at the small sizes, like a stream chunk
    Q = torch.from_numpy(np.ascontiguousarray(np.stack([q] * npairs))).cuda(); T = torch.from_numpy(np.ascontiguousarray(np.stack([t] * npairs))).cuda()
    idx_m = torch.zeros(npairs, N, 2, dtype=torch.int32, device="cuda"); dist_m = torch.zeros(npairs, N, 2, dtype=torch.float32, device="cuda")
    norms = torch.zeros(2 * npairs * N, dtype=torch.float32, device="cuda")
    ms = C.c_float()
    torch.cuda.synchronize()
    rc = lib.knn_mfma_run(Q.data_ptr(), T.data_ptr(), N, N, npairs, N, norms.data_ptr(), idx_m.data_ptr(), dist_m.data_ptr(), 5, C.byref(ms), None)
    assert rc == 0, rc
    # the library's exact matcher, one pair per call (its batched form is internal), timed over the same number of pairs
    idx_e = torch.zeros(N, 2, dtype=torch.int32, device="cuda"); dist_e = torch.zeros(N, 2, dtype=torch.float32, device="cuda")
    ctx.knn2_f32(Q[0], T[0], idx_e, dist_e); ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = ctx._torch_stream()
    with torch.cuda.stream(st):
        e0.record(st)
        for p in range(npairs):
            ctx.knn2_f32(Q[p], T[p], idx_e, dist_e)
        e1.record(st)
    ctx.synchronize(); torch.cuda.synchronize()
    ms_exact = e0.elapsed_time(e1) / npairs
    ie, de = idx_e.cpu().numpy(), dist_e.cpu().numpy(); im, dm = idx_m[0].cpu().numpy(), dist_m[0].cpu().numpy()
    ratio_e = de[:, 0] < 0.5 * de[:, 1]; ratio_m = dm[:, 0] < 0.5 * dm[:, 1]
    res[str(N)] = dict(pairs_per_launch=npairs, ms_per_pair_mfma_gram=round(ms.value / npairs, 4), ms_per_pair_exact_valu=round(ms_exact, 4),
                       tflops_exact=round(N * N * 128 * 3 / (ms_exact * 1e-3) / 1e12, 1),
                       tflops_gram=round(N * N * 128 * 2 / (ms.value / npairs * 1e-3) / 1e12, 1),
                       nearest_differs=int((ie[:, 0] != im[:, 0]).sum()), second_differs=int((ie[:, 1] != im[:, 1]).sum()),
                       ratio_decision_differs=int((ratio_e != ratio_m).sum()),
                       distance_bits_differ=int((de != dm).sum()), queries=N)
ctx.close()
print(json.dumps(res, indent=1))
