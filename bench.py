#!/usr/bin/env python3
"""bench.py -- frame-pair homographies/sec on MI355X (BASELINE.json metric).

Workload at N=1 = BASELINE.json configs[1]: "synthetic 720p pair batch, 1xMI355X, ORB 500 kp": one step = one
batch of `--pairs` independent 1280x720 BGR frame pairs (2 frames each), already resident in HBM, through the
whole hot path (gray -> pyramid -> FAST/NMS -> select/Harris -> orientation + rBRIEF -> 2-NN match + filters ->
RANSAC #1 -> static filter -> RANSAC #2 + LM) to one 3x3 f64 H + status per pair on the device.
N>1: one process per GPU (torch.distributed / RCCL), every rank processes its own batch (weak scaling, pairs are
independent units) and the per-pair H records are gathered with one all_gather per step (inside the timed region).

Prints ONE JSON line on rank 0 (driver contract) with two extra objects:
  roofline     -- the dominant kernel group (by device time) against the HBM roofline: algorithmic bytes per
                  launch / average launch duration measured with hipEvents on the kernels' own stream.
  cpu_baseline -- the CPU oracle (C++ restatement, oracle/) timed on a bounded sample of the same frames on this
                  host's cores (N=1, rank 0 only).  A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); measured copy peak 6290


def level_sizes(w, h):
    sizes = []
    for l in range(8):
        s = np.float32(np.float64(np.float32(1.2)) ** l)
        sizes.append((int(np.rint(np.float32(w) / s)), int(np.rint(np.float32(h) / s))))
    return sizes


def algorithmic_bytes(w, h, nfeat, channels):
    """SURVEY 8d byte model, per FRAME for the frame stages and per PAIR for the pair stages."""
    lv = level_sizes(w, h)
    areas = [a * b for a, b in lv]
    A, PA = areas[0], sum(areas)
    per_frame = {
        "gray": (channels * A + A),                       # read source, write level 0
        "pyramid": sum(areas[:-1]) + sum(areas[1:]),      # read levels 0..6, write levels 1..7
        "fast": PA,                                       # read every level once
        "select": 0,                                      # candidate lists + 9x9 patches: not streaming
        "describe": min(PA, 1369 * nfeat) + 48 * nfeat,   # 37x37 neighbourhoods + keypoint/descriptor records
    }
    per_pair = {"knn2": 72 * nfeat, "filter": 0, "ransac_static": 0, "ransac_final": 80}
    return per_frame, per_pair


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1024, help="independent frame pairs per step per GPU")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic pairs generated (tiled to --pairs)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=500)
    ap.add_argument("--channels", type=int, default=3, choices=[1, 3])
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-temporal", action="store_true", help="switch off the two exact shortcuts that lean on consecutive frames / calls looking alike (threshold sharing inside a pair, threshold hint across calls)")
    ap.add_argument("--smooth", type=int, default=0, help="3x3 box-blur passes over the synthetic frames (content with fewer, weaker corners; informational)")
    ap.add_argument("--contexts", type=int, default=1, help="independent contexts/streams the steps alternate over")
    args = ap.parse_args()

    import torch
    from evenvizion_amd import synthetic
    from evenvizion_amd._lib import Context, MODE_INDEPENDENT_PAIRS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("EVH_BENCH_FORCE_DIST") == "1")
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libevhip.so is the only compute backend")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    w, h, B = args.width, args.height, args.pairs
    # synthetic frames, a few unique pairs tiled to the batch; every copy is its own HBM region
    U = min(args.unique, B)
    gray, Htrue = synthetic.make_pair_batch(2, U, w, h)
    for _ in range(max(args.smooth, 0)):                     # informational: lower-contrast, corner-poor content
        g = np.pad(gray.astype(np.uint16), ((0, 0), (1, 1), (1, 1)), mode="edge")
        acc = sum(g[:, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
        gray = ((acc + 4) // 9).astype(np.uint8)
    host = gray if args.channels == 1 else synthetic.gray_to_bgr(gray)
    uniq = torch.from_numpy(np.ascontiguousarray(host)).to(dev)            # [2U, h, w(,3)]
    reps = -(-B // U)
    frames = uniq.repeat((reps,) + (1,) * (uniq.dim() - 1))[:2 * B].contiguous()   # physical copies in HBM
    del host, uniq

    # Steps alternate over `--contexts` independent contexts, each with its own HIP stream (+ its solve stream):
    # consecutive steps have no data dependence, so their kernels overlap and fill each other's launch tails.
    NCTX = max(1, args.contexts)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NCTX)]
    ctxs = [Context(device=local_rank, max_w=w, max_h=h, max_features=args.nfeatures, max_frames=2 * B,
                    stream=streams[i].cuda_stream) for i in range(NCTX)]
    stream, ctx = streams[0], ctxs[0]
    # The RANSAC kernels of a step run on the context's solve stream and overlap the next step's detect kernels;
    # results are double-buffered so that a step never overwrites what the previous step's gather still reads.
    for c_ in ctxs:
        c_.set_async_solve(True)
        if args.no_temporal:
            c_.set_fast_share(False); c_.set_fast_hint(False)
    NBUF = 2 * NCTX
    Hs = [torch.zeros(B, 9, dtype=torch.float64, device=dev) for _ in range(NBUF)]
    sts = [torch.full((B,), -1, dtype=torch.int32, device=dev) for _ in range(NBUF)]
    gathered = torch.zeros(world * B, 9, dtype=torch.float64, device=dev) if use_dist else None
    gstream = torch.cuda.Stream(device=dev) if use_dist else None
    gdone = [None] * NBUF
    counter = [0]

    def step():
        k = counter[0] % NBUF
        cx, sx = ctxs[counter[0] % NCTX], streams[counter[0] % NCTX]
        counter[0] += 1
        if use_dist and gdone[k] is not None:
            sx.wait_event(gdone[k])               # the gather that read this buffer NBUF steps ago has finished
        cx.pair_homography_batch(frames, B, MODE_INDEPENDENT_PAIRS, Hs[k], sts[k], nfeatures=args.nfeatures)
        if use_dist:
            # RCCL over xGMI: gather the per-pair H records of THIS step on a side stream behind the solve
            cx.solve_wait(gstream.cuda_stream)
            with torch.cuda.stream(gstream):
                dist.all_gather_into_tensor(gathered, Hs[k])
                ev = torch.cuda.Event()
                ev.record(gstream)
            gdone[k] = ev

    if True:
        for _ in range(max(args.warmup, NCTX)):
            step()
        torch.cuda.synchronize(dev)
        if saved_stdout is not None:
            if use_dist:
                dist.barrier()
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None
        for c_ in ctxs:
            c_.profile_read()         # drop warm-up spans
            c_.profile_enable(True)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    stages = {}
    for c_ in ctxs:
        c_.profile_enable(False)
        for name, (cnt_, ms_) in c_.profile_read().items():
            a_, b_ = stages.get(name, (0, 0.0))
            stages[name] = (a_ + cnt_, b_ + ms_)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    for c_ in ctxs:
        c_.synchronize()
    last = (counter[0] - 1) % NBUF
    H, status = Hs[last], sts[last]
    st = status.cpu().numpy()
    ok_frac = float((st == 0).mean())
    total_pairs = world * B * args.steps
    value = total_pairs / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel group --------------------------------------------------------------------
    per_frame, per_pair = algorithmic_bytes(w, h, args.nfeatures, args.channels)
    stage_ms = {k: (v[1] / max(v[0], 1)) for k, v in stages.items()}      # average duration of one launch group
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    by = per_frame.get(dom, 0) * 2 * B if dom in per_frame else per_pair.get(dom, 0) * B
    achieved = by / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = "%s@%dx%dx%d_n%d_c%d" % (dom, w, h, B, args.nfeatures, args.channels)
            traffic = tj.get(key)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(by), "avg_launch_ms": round(stage_ms[dom], 4),
                "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()}}
    # the dominant group is VALU-issue-bound, not HBM-bound: say so next to the HBM fraction.  Instruction
    # count from the committed PMC pass (profiles/valu.json), duration measured live; ceilings measured by
    # tools/ubench/valu_rate.hip (profiles/r01_valu_issue_rates.txt): 0.96 (3-input / packed / perm forms) .. 1.75
    # (plain 32-bit add/and) wave-instructions per clock per CU.
    vpath = os.path.join(ROOT, "profiles", "valu.json")
    if os.path.exists(vpath) and stage_ms[dom] > 0:
        try:
            vj = json.load(open(vpath)).get("%s@%dx%dx%d_n%d_c%d" % (dom, w, h, B, args.nfeatures, args.channels))
            if vj:
                props = torch.cuda.get_device_properties(dev)
                clk = 2.4e9
                per = vj["valu_wave_insts_per_launch"] / (stage_ms[dom] * 1e-3) / (props.multi_processor_count * clk)
                roofline["valu_issue"] = {"wave_insts_per_launch": vj["valu_wave_insts_per_launch"],
                                          "per_clk_per_cu_at_2.4GHz": round(per, 3), "ceiling": [0.96, 1.75],
                                          "source": vj["source"]}
        except Exception:
            pass
    # whole-pipeline view: independent pair = 2*B_frame + B_match + B_out
    pipe_bytes = 2 * sum(per_frame.values()) + sum(per_pair.values())
    roofline["pipeline_bytes_per_pair"] = int(pipe_bytes)
    roofline["pipeline_frac"] = round(pipe_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5)

    # ---- CPU baseline (oracle) on a bounded sample, rank 0 at N=1 only --------------------------------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_pairs != 0:
        from oracle import oracle as O
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, 16)   # the GPU box's CPU share for one GPU
        n_s = args.cpu_pairs if args.cpu_pairs > 0 else min(B, 64)
        idx = np.arange(2 * n_s) % (2 * U)                                   # the batch's first n_s pairs
        sample = np.ascontiguousarray(gray[idx])
        O.lib()
        t0 = time.perf_counter()
        Ho, so = O.pairs_gray_batch(sample, nfeatures=args.nfeatures, threads=cores)
        tc = time.perf_counter() - t0
        # agreement of the measured run with the checker on the sample
        Hg = H.cpu().numpy().reshape(-1, 3, 3)[:n_s]
        same = bool(np.array_equal(st[:n_s], so)) and bool(np.allclose(Hg[so == 0], Ho[so == 0], rtol=1e-9, atol=1e-12))
        cpu_baseline = {"value": round(n_s / tc, 3), "unit": "frame-pair homographies/s", "cores": cores, "kind": "port",
                        "sample": "%d of the batch's %d pairs (gray frames), oracle/ C++ restatement, %d std::threads, "
                                  "one pair per thread" % (n_s, B, cores),
                        "seconds": round(tc, 2), "gpu_matches_oracle_on_sample": same}

    if rank == 0:
        out = {
            "metric": "frame-pair homographies/sec @720p" if (w, h) == (1280, 720) else "frame-pair homographies/sec",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d %s pair batch, %d independent pairs/step/GPU, ORB %d kp, "
                                   "RANSAC max 2000 conf 0.995 (BASELINE.json configs[1])"
                                   % (w, h, "BGR" if args.channels == 3 else "gray", B, args.nfeatures),
                       "pairs_per_step_per_gpu": B, "unique_pairs": U, "parallelism": "pairs sharded, dp%d" % world, "contexts_per_gpu": NCTX,
                       "pairs_ok_fraction": ok_frac, "smooth_passes": args.smooth,
                       "fast_threshold_sharing_in_pair": not args.no_temporal, "fast_threshold_hint_across_calls": not args.no_temporal,
                       "arithmetic": "u8/i32 pixels+descriptors, f32 Harris+reprojection, f64 DLT+LM"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    for c_ in ctxs:
        c_.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
