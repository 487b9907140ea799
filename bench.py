#!/usr/bin/env python3
"""bench.py -- frame-pair homographies/sec on MI355X (BASELINE.json metric).

Default workload (what the driver runs) = BASELINE.json configs[1], "synthetic 720p pair batch, 1xMI355X, ORB 500
kp": one step = one batch of `--pairs` (1024) independent 1280x720 BGR frame pairs, already resident in HBM, through
the whole hot path (gray -> pyramid -> FAST/NMS -> select/Harris -> orientation + rBRIEF -> 2-NN match + filters ->
RANSAC #1 -> static filter -> RANSAC #2 + LM) to one 3x3 f64 H + status per pair on the device.  The batch holds 64
DISTINCT synthetic pairs and every step starts at another offset into them, so consecutive calls never see the same
frame in the same slot; the same loop with the two content-dependent FAST shortcuts switched off is timed right
after and reported as config.no_temporal_value.

--config selects the other BASELINE.json workloads (one JSON line each, committed under profiles/):
  2  SURVEY 8d "config 2": 256 independent 720p pairs per step, 40 steps
  3  configs[2]: ONE 1280x720 stream per GPU, ORB 2000 kp, stream semantics (running superposition carried on the
     device), 48 pairs per step
  4  configs[3]: ONE 1920x1080 stream over all GPUs, two-phase (phase 1 sharded, static rows all-gathered over
     RCCL, sequential scan on every rank), 48 pairs per rank per step
  5  configs[4]: one 3840x2160 stream per GPU, ORB 4000 kp, 16 pairs per step

`--gpus N` starts N ranks itself (one process per GPU, torch.distributed over RCCL) unless it is already running
under a launcher (WORLD_SIZE set, which must then equal N).  Prints ONE JSON line on rank 0 (driver contract) with
  roofline     -- the dominant kernel group (by device time) against the HBM roofline: algorithmic bytes per launch /
                  average launch duration measured with hipEvents on the kernels' own stream.
  cpu_baseline -- the CPU oracle (C++ restatement, oracle/) timed on a bounded sample of the same frames on this
                  host's cores (N=1, rank 0 only).  A reported baseline, not the target.
"""
import argparse
import json
import os
import socket
import subprocess
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4, 5], help="BASELINE.json workload (see module docstring)")
    ap.add_argument("--pairs", type=int, default=None, help="frame pairs per step per GPU")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic pairs generated (tiled to --pairs)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--nfeatures", type=int, default=None)
    ap.add_argument("--channels", type=int, default=3, choices=[1, 3])
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-temporal", action="store_true", help="switch off the two exact shortcuts that lean on consecutive frames / calls looking alike (threshold sharing inside a pair, threshold hint across calls) for the MAIN timed loop")
    ap.add_argument("--force-max-iters", action="store_true", help="stream configs: evaluate all 2000 RANSAC samples (fixed-iteration stress variant)")
    ap.add_argument("--smooth", type=int, default=0, help="3x3 box-blur passes over the synthetic frames (content with fewer, weaker corners; informational)")
    ap.add_argument("--order", choices=["reference", "canonical"], default="reference",
                    help="ORB key-point order (include/evhip.h evh_set_keypoint_order): reference = OpenCV 3.4.2 on libstdc++, what the "
                         "reference's recorded run agrees with (default); canonical = all ties, (level, y, x) order, FAST threshold lifting applies")
    ap.add_argument("--solver", choices=["exact", "fast"], default="exact",
                    help="LM's 8x8 systems: exact = the operator's Jacobi eigen-solve (H bit-identical to the oracle, default); fast = LDL^T "
                         "(include/evhip.h evh_set_solver_mode: frame corners within ~1e-3 px of the exact mode)")
    ap.add_argument("--skip-no-temporal", action="store_true", help="do not time the extra no-temporal loop (profiling runs: every launch of the run is then the same workload)")
    ap.add_argument("--gen-procs", type=int, default=0, help="host processes that generate the synthetic pairs (0 = auto; use 1 under rocprofv3: no child processes)")
    ap.add_argument("--sync-solve", action="store_true",
                    help="diagnostic: RANSAC on the detect stream (no overlap with the next step), for clean per-stage times")
    ap.add_argument("--contexts", type=int, default=1, help="independent contexts/streams the steps alternate over (pair configs)")
    args = ap.parse_args()
    cfg = {1: dict(kind="pairs", w=1280, h=720, nfeat=500, pairs=1024, steps=10),
           2: dict(kind="pairs", w=1280, h=720, nfeat=500, pairs=256, steps=40),
           3: dict(kind="stream", w=1280, h=720, nfeat=2000, pairs=48, steps=10),
           4: dict(kind="twophase", w=1920, h=1080, nfeat=2000, pairs=48, steps=6),
           5: dict(kind="stream", w=3840, h=2160, nfeat=4000, pairs=16, steps=6)}[args.config]
    args.kind = cfg["kind"]
    args.width = args.width or cfg["w"]
    args.height = args.height or cfg["h"]
    args.nfeatures = args.nfeatures or cfg["nfeat"]
    args.pairs = args.pairs or cfg["pairs"]
    args.steps = args.steps if args.steps is not None else cfg["steps"]
    return args


def spawn_ranks(n):
    """--gpus N without a launcher: start N ranks of this script, one per GPU.  The parent never touches the GPU (no
    torch import at all), so nothing is exec'ed or forked from a process that has initialised HIP."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def _make_pair_job(a):
    from evenvizion_amd import synthetic
    return synthetic.make_pair(*a)


def make_unique_pairs(config_seed, U, w, h, nproc=0):
    """U distinct synthetic pairs (SURVEY 8d seeds 1000*config + pair index), generated on the host cores in parallel."""
    import multiprocessing as mp
    # EVH_BENCH_CACHE=<dir>: keep the generated frames between runs on one box (the profiling passes of
    # tools/collect_profiles.sh run the same workload six times; under rocprofv3 the generator may not fork)
    cache = os.environ.get("EVH_BENCH_CACHE")
    cache_file = os.path.join(cache, "pairs_seed%d_%dx%d_u%d.npy" % (config_seed, w, h, U)) if cache else None
    if cache_file and os.path.exists(cache_file):
        return np.load(cache_file, allow_pickle=False)
    jobs = [(1000 * config_seed + p, w, h) for p in range(U)]
    if nproc <= 0:
        nproc = min(U, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8,
                    16 if int(os.environ.get("WORLD_SIZE", "1")) == 1 else 8)
    if nproc > 1:
        with mp.get_context("spawn").Pool(nproc) as pool:
            res = pool.map(_make_pair_job, jobs)
    else:
        res = [_make_pair_job(j) for j in jobs]
    gray = np.empty((2 * U, h, w), np.uint8)
    for p, (a, b, _) in enumerate(res):
        gray[2 * p] = a; gray[2 * p + 1] = b
    if cache_file:
        os.makedirs(cache, exist_ok=True)
        np.save(cache_file, gray)
    return gray


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch %d ranks (or drop the launcher and let --gpus start them)"
                 % (args.gpus, world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    w, h, B, nfeat = args.width, args.height, args.pairs, args.nfeatures

    # ---- synthetic frames on the host, before anything touches the GPU (the generator pool is spawned from here) -----
    from evenvizion_amd import synthetic
    if args.kind == "pairs":
        U = max(1, min(args.unique, B))
        gray = make_unique_pairs(2, U, w, h, args.gen_procs)
        for _ in range(max(args.smooth, 0)):                     # informational: lower-contrast, corner-poor content
            g = np.pad(gray.astype(np.uint16), ((0, 0), (1, 1), (1, 1)), mode="edge")
            acc = sum(g[:, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
            gray = ((acc + 4) // 9).astype(np.uint8)
    else:
        # one synthetic stream per rank; a there-and-back walk over 25 generated frames makes an endless stream whose
        # consecutive frames always differ by one small camera motion
        base, _ = synthetic.make_stream(11 + 7 * rank, 25, w, h)
        gray = np.concatenate([base, base[-2:0:-1]])             # period 48
        U = len(gray)

    import torch
    from evenvizion_amd._lib import Context, MODE_INDEPENDENT_PAIRS

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

    host = gray if args.channels == 1 else synthetic.gray_to_bgr(gray)
    uniq = torch.from_numpy(np.ascontiguousarray(host)).to(dev)            # [2U | period, h, w(,3)]
    del host
    counter = [0]
    NCTX = max(1, args.contexts) if args.kind == "pairs" else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(NCTX)]

    if args.kind == "pairs":
        # B + U pairs resident in HBM (physical copies); step s reads the B pairs starting at pair (37 * s) mod U
        reps = -(-(B + U) // U)
        frames = uniq.repeat((reps,) + (1,) * (uniq.dim() - 1))[:2 * (B + U)].contiguous()
        del uniq
        ctxs = [Context(device=local_rank, max_w=w, max_h=h, max_features=nfeat, max_frames=2 * B,
                        stream=streams[i].cuda_stream) for i in range(NCTX)]
        # The RANSAC kernels of a step run on the context's solve stream and overlap the next step's detect kernels;
        # results are double-buffered so that a step never overwrites what the previous step's gather still reads.
        for c_ in ctxs:
            c_.set_async_solve(not args.sync_solve)
        NBUF = 2 * NCTX
        Hs = [torch.zeros(B, 9, dtype=torch.float64, device=dev) for _ in range(NBUF)]
        sts = [torch.full((B,), -1, dtype=torch.int32, device=dev) for _ in range(NBUF)]
        gathered = torch.zeros(world * B, 9, dtype=torch.float64, device=dev) if use_dist else None
        gstream = torch.cuda.Stream(device=dev) if use_dist else None
        gdone = [None] * NBUF
        last_off = [0]

        def step():
            k = counter[0] % NBUF
            cx, sx = ctxs[counter[0] % NCTX], streams[counter[0] % NCTX]
            off = (37 * counter[0]) % U
            last_off[0] = off
            counter[0] += 1
            if use_dist and gdone[k] is not None:
                sx.wait_event(gdone[k])               # the gather that read this buffer NBUF steps ago has finished
            cx.pair_homography_batch(frames[2 * off:2 * (off + B)], B, MODE_INDEPENDENT_PAIRS, Hs[k], sts[k], nfeatures=nfeat)
            if use_dist:
                # RCCL over xGMI: gather the per-pair H records of THIS step on a side stream behind the solve
                cx.solve_wait(gstream.cuda_stream)
                with torch.cuda.stream(gstream):
                    dist.all_gather_into_tensor(gathered, Hs[k])
                    ev = torch.cuda.Event()
                    ev.record(gstream)
                gdone[k] = ev

        def set_temporal(on):
            for c_ in ctxs:
                c_.set_fast_share(on); c_.set_fast_hint(on)
        pairs_per_step = world * B
    else:
        # stream kinds: a step is one chunk of B pairs (B + 1 consecutive frames, chunks overlap by one frame); the
        # chunk's frames are gathered from the periodic walk into one contiguous device buffer (not timed work of
        # the path: a capture would hand over contiguous chunks)
        F = B + 1
        period = U
        ctxs = [Context(device=local_rank, max_w=w, max_h=h, max_features=nfeat, max_frames=max(F, 2), stream=streams[0].cuda_stream)]
        cx = ctxs[0]
        # the sequential scan of a chunk runs on the context's solve stream: the next chunk's detect / 2-NN kernels overlap it
        # (its filter waits for the scan: the pair buffers are shared); --sync-solve switches that off
        if args.kind == "stream":
            cx.set_async_solve(not args.sync_solve)
        chunk = torch.empty((F,) + tuple(uniq.shape[1:]), dtype=torch.uint8, device=dev)
        state = torch.zeros(18, dtype=torch.float64, device=dev)
        have_state = [False]
        npairs_glob = world * B if args.kind == "twophase" else B
        Hs = [torch.zeros(npairs_glob, 9, dtype=torch.float64, device=dev)]
        sts = [torch.full((npairs_glob,), -1, dtype=torch.int32, device=dev)]
        gathered = torch.zeros(world * B, 9, dtype=torch.float64, device=dev) if (use_dist and args.kind == "stream") else None
        first_chunk_frames = [None]
        from evenvizion_amd import sharding

        def load_chunk(first_frame):
            idx = (torch.arange(F, device=dev) + first_frame) % period
            torch.index_select(uniq, 0, idx, out=chunk)

        def step():
            s = counter[0]
            counter[0] += 1
            if args.kind == "stream":
                load_chunk(s * B)
                if s == 0:
                    first_chunk_frames[0] = chunk.clone()
                cx.stream_homography_batch(chunk, Hs[0], sts[0], state_in=state if have_state[0] else None,
                                           state_out=state, nfeatures=nfeat, force_max_iters=args.force_max_iters)
                have_state[0] = True
                if use_dist:
                    cx.order_torch_after()
                    dist.all_gather_into_tensor(gathered, Hs[0])
            else:
                # ONE stream over all ranks: this step covers world * B pairs; rank r runs phase 1 on its block of B
                # pairs (B + 1 frames), the static rows are all-gathered, every rank runs the sequential scan
                load_chunk(s * world * B + rank * B)
                if s == 0:
                    first_chunk_frames[0] = chunk.clone()
                rows, counts, st1 = cx.stream_static_batch(chunk, nfeatures=nfeat, force_max_iters=args.force_max_iters)
                rows, counts, st1 = sharding.gather_static_rows(rows, counts, st1, world * B)
                H_, st_ = cx.stream_scan(rows, counts, st1, state_in=state if have_state[0] else None, state_out=state,
                                         force_max_iters=args.force_max_iters)
                have_state[0] = True
                Hs[0], sts[0] = H_, st_

        def set_temporal(on):
            cx.set_fast_share(on); cx.set_fast_hint(on)
        pairs_per_step = world * B

    def timed(nsteps):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    set_temporal(not args.no_temporal)
    for c_ in ctxs:
        c_.set_keypoint_order(1 if args.order == "reference" else 0)
        c_.set_solver_mode(1 if args.solver == "fast" else 0)
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
    elapsed = timed(args.steps)
    stages = {}
    for c_ in ctxs:
        c_.profile_enable(False)
        for name, (cnt_, ms_) in c_.profile_read().items():
            a_, b_ = stages.get(name, (0, 0.0))
            stages[name] = (a_ + cnt_, b_ + ms_)
    for c_ in ctxs:
        c_.synchronize()
    if args.kind == "pairs":
        last = (counter[0] - 1) % len(Hs)
        H, status, off_last = Hs[last].clone(), sts[last].clone(), last_off[0]
    else:
        H, status = Hs[0].clone(), sts[0].clone()
    st = status.cpu().numpy()
    ok_frac = float((st == 0).mean())
    value = pairs_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # the same loop without the two content-dependent (exact) FAST shortcuts, reported next to the headline
    no_temporal_value = None
    if not args.no_temporal and args.kind == "pairs" and not args.skip_no_temporal:
        set_temporal(False)
        for _ in range(2):
            step()
        nt_steps = max(2, min(args.steps, 10))
        no_temporal_value = round(pairs_per_step * nt_steps / timed(nt_steps), 2)
        set_temporal(True)

    # the same loop in the canonical key-point order (rounds 1-3: all ties, (level, y, x) order; FAST threshold lifting applies):
    # faster, but H then agrees with the reference's OpenCV run only where RANSAC's consensus does not depend on the draw
    canonical_order_value = None
    if args.order == "reference" and args.kind == "pairs" and not args.skip_no_temporal:
        for c_ in ctxs:
            c_.set_keypoint_order(0)
        for _ in range(2):
            step()
        co_steps = max(2, min(args.steps, 10))
        canonical_order_value = round(pairs_per_step * co_steps / timed(co_steps), 2)
        for c_ in ctxs:
            c_.set_keypoint_order(1)

    # the same loop with the tolerance-mode solver (LDL^T for LM's 8x8 systems): what a caller who accepts frame corners within
    # ~1e-3 px of the operator's own arithmetic gets; reported next to the headline, which stays on the exact solver
    fast_solver_value = None
    if args.solver == "exact" and not args.skip_no_temporal:
        for c_ in ctxs:
            c_.set_solver_mode(1)
        for _ in range(2):
            step()
        fs_steps = max(2, min(args.steps, 10))
        fast_solver_value = round(pairs_per_step * fs_steps / timed(fs_steps), 2)
        for c_ in ctxs:
            c_.set_solver_mode(0)

    # ---- roofline of the dominant kernel group --------------------------------------------------------------------
    per_frame, per_pair = algorithmic_bytes(w, h, nfeat, args.channels)
    stage_ms = {k: (v[1] / max(v[0], 1)) for k, v in stages.items() if v[0] > 0}      # average duration of one launch group
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    frames_per_launch = 2 * B if args.kind == "pairs" else B + 1
    pairs_per_launch = world * B if (args.kind == "twophase" and dom == "ransac_final") else B
    by = per_frame[dom] * frames_per_launch if dom in per_frame else per_pair.get(dom, 0) * pairs_per_launch
    achieved = by / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes_per_launch": int(by), "avg_launch_ms": round(stage_ms[dom], 4),
                "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()}}
    # A stream config's dominant kernel is the sequential scan (one wave per SIMD, ~80 bytes per pair): an HBM fraction says
    # nothing about it.  What bounds it is a chain of dependent f64 operations; the kernel can account its own cycles by phase
    # (EVH_RANSAC_PROF), so one more step of the same workload is run in a child process with the accounting on and its
    # per-pair cycle model is attached as roofline.scan_cycles (clock 2.4 GHz: cycles / 2.4e6 = ms per pair).
    if dom == "ransac_final" and args.kind != "pairs" and rank == 0 and not args.skip_no_temporal and not os.environ.get("EVH_RANSAC_PROF"):
        try:
            import re
            env = dict(os.environ, EVH_RANSAC_PROF="1")
            for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "EVH_BENCH_FORCE_DIST"):
                env.pop(k_, None)
            argv = [sys.executable, os.path.abspath(__file__), "--config", str(args.config), "--gpus", "1", "--steps", "1", "--warmup", "1",
                    "--cpu-pairs", "0", "--skip-no-temporal", "--gen-procs", "1", "--solver", args.solver, "--order", args.order]
            if args.force_max_iters:
                argv.append("--force-max-iters")
            pr = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=600)
            lines = [l for l in pr.stderr.splitlines() if l.startswith("[evh ransac_final prof]")]
            if lines:
                t = lines[-1]
                def num(pat):
                    m = re.search(pat, t)
                    return float(m.group(1)) if m else None
                roofline["scan_cycles"] = {
                    "per_pair_total": num(r"total (\d+)"), "hypotheses": num(r"hyp (\d+)"), "compact": num(r"compact (\d+)"),
                    "refit": num(r"refit (\d+)"), "lm": num(r" lm (\d+)"), "lm_iterations": num(r"iters ([\d.]+)"),
                    "lm_solves": num(r"solve8 (\d+)"), "lm_evaluations": num(r"eval (\d+)"),
                    "jacobi_rotations_9x9": num(r"9x9 ([\d.]+)"), "jacobi_rotations_8x8": num(r"8x8 ([\d.]+)"),
                    "source": "EVH_RANSAC_PROF=1 on one more step of this workload in a child process (in-kernel s_memtime accounting, thread 0 of each scan)"}
                tot = roofline["scan_cycles"]["per_pair_total"]
                if tot:
                    roofline["scan_cycles"]["ms_per_pair_at_2.4GHz"] = round(tot / 2.4e6, 4)
        except Exception as e:      # the model is informational
            roofline["scan_cycles"] = {"error": str(e)[:200]}
    # HBM bytes (PMC) and VALU instruction counts are NOT measured in this run: they come from the committed rocprofv3
    # --pmc passes of exactly this workload (profiles/traffic.json, profiles/valu.json, keyed by kernel group and
    # geometry) and are labelled as replayed; a workload without a committed pass reports null.
    key = "%s@%dx%dx%d_n%d_c%d" % (dom, w, h, B, nfeat, args.channels)
    for fname, field in (("traffic.json", "traffic"), ("valu.json", "valu_issue")):
        path = os.path.join(ROOT, "profiles", fname)
        try:
            entry = json.load(open(path)).get(key) if (args.kind in ("pairs", "stream") and os.path.exists(path)) else None
        except Exception:
            entry = None
        if entry is None:
            continue
        if field == "traffic":
            if isinstance(entry, dict):
                roofline["traffic"] = entry["hbm_bytes"]
                roofline["traffic_correction"] = {
                    "fetch_factor": entry["fetch_correction"], "fetch_counter_bytes": entry["fetch_counter_bytes"],
                    "write_bytes": entry["write_bytes"],
                    "why": ("FETCH_SIZE tallies 128-byte lines at 64 bytes on gfx950; calibrated on this kernel's own staging "
                            "pattern: profiles/r03_fetch_calibration.txt") if dom == "fast" else
                           ("FETCH_SIZE tallies 128-byte lines at 64 bytes on gfx950 (factor calibrated on the FAST kernel's line-sized "
                            "reads, profiles/r03_fetch_calibration.txt; an upper bound for this kernel's shorter reads)")}
            else:                     # a round-2 record: raw counter sum
                roofline["traffic"] = entry
            roofline["traffic_over_algorithmic"] = round(roofline["traffic"] / max(by, 1), 3)
            roofline["traffic_source"] = "replayed: profiles/%s[%s] (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not this run)" % (fname, key)
        elif stage_ms[dom] > 0:
            props = torch.cuda.get_device_properties(dev)
            per = entry["valu_wave_insts_per_launch"] / (stage_ms[dom] * 1e-3) / (props.multi_processor_count * 2.4e9)
            roofline["valu_issue"] = {"wave_insts_per_launch": entry["valu_wave_insts_per_launch"],
                                      "per_clk_per_cu_at_2.4GHz": round(per, 3), "ceiling": [0.96, 1.75],
                                      "valu_source": "replayed: profiles/%s[%s] (%s); duration measured in this run" % (fname, key, entry.get("source", ""))}
    if dom == "fast":
        # the FAST tile grid covers only the pixels ORB can emit (31-px border rule) plus one ring: the kernel reads this
        # fraction of the P*A bytes the SURVEY 8d model charges it with (`achieved` keeps the model's bytes)
        lvs = level_sizes(w, h)
        roofline["fast_tiled_fraction"] = round(sum(max(1, -(-(a - 55) // 128)) * 128 * max(1, -(-(b - 62) // 28)) * 28 for a, b in lvs)
                                                / float(sum(a * b for a, b in lvs)), 4)
    # whole-pipeline view: independent pair = 2*B_frame + B_match + B_out; a stream pair = 1 frame
    pipe_bytes = (2 if args.kind == "pairs" else 1) * sum(per_frame.values()) + sum(per_pair.values())
    roofline["pipeline_bytes_per_pair"] = int(pipe_bytes)
    roofline["pipeline_frac"] = round(pipe_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5)
    # the same with SURVEY 8d's own byte model: B_frame = A*(1 + (P-1) + P) + min(P*A, 1369*N) + 48*N with the gray frame
    # resident (+ 3*A when the BGR entry is timed), B_match = 72*N, B_out = 80 -- source read once, levels 1..7 written
    # once, every level read once; no re-read by the unfused pyramid, no write of level 0
    A8 = areas_sum = None
    lv8 = level_sizes(w, h)
    A8 = lv8[0][0] * lv8[0][1]; areas_sum = sum(a * b for a, b in lv8)
    b_frame = A8 + (areas_sum - A8) + areas_sum + min(areas_sum, 1369 * nfeat) + 48 * nfeat + (3 * A8 if args.channels == 3 else 0)
    pipe8d = (2 if args.kind == "pairs" else 1) * b_frame + 72 * nfeat + 80
    roofline["pipeline_bytes_per_pair_survey8d"] = int(pipe8d)
    roofline["pipeline_frac_survey8d"] = round(pipe8d * value / world / 1e9 / HBM_PEAK_GBS, 5)

    # rocprof-measured HBM bytes of a whole step (corrected FETCH_SIZE + WRITE_SIZE over every kernel, committed PMC passes of
    # this workload) against this run's step time: what "rocprof HBM GB/s against the chip's peak" reads for the path
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        ent = tj.get("step@%dx%dx%d_n%d_c%d" % (w, h, B, nfeat, args.channels)) if args.kind in ("pairs", "stream") else None
    except Exception:
        ent = None
    if ent:
        gbs = ent["hbm_bytes_per_step"] / (ms_per_step * 1e-3) / 1e9
        roofline["hbm_traffic"] = {"bytes_per_step": ent["hbm_bytes_per_step"], "GBps": round(gbs, 1),
                                   "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                                   "source": "replayed: " + ent["source"] + "; step time of this run"}

    # ---- CPU baseline (oracle) on a bounded sample, rank 0 at N=1 only --------------------------------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_pairs != 0:
        from oracle import oracle as O
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, 16)   # the GPU box's CPU share for one GPU
        O.lib()
        Hg_all = H.cpu().numpy().reshape(-1, 3, 3)
        if args.kind == "pairs":
            n_s = args.cpu_pairs if args.cpu_pairs > 0 else min(B, 64)
            idx = (np.arange(2 * n_s) + 2 * off_last) % (2 * U)                  # the last step's first n_s pairs
            sample = np.ascontiguousarray(gray[idx])
            t0 = time.perf_counter()
            Ho, so = O.pairs_gray_batch(sample, nfeatures=nfeat, threads=cores)
            tc = time.perf_counter() - t0
            Hg = Hg_all[:n_s]
            same = bool(np.array_equal(st[:n_s], so)) and bool(np.allclose(Hg[so == 0], Ho[so == 0], rtol=1e-9, atol=1e-12))
            cpu_baseline = {"value": round(n_s / tc, 3), "unit": "frame-pair homographies/s", "cores": cores, "kind": "port",
                            "sample": "%d of the batch's %d pairs (gray frames), oracle/ C++ restatement, %d std::threads, "
                                      "one pair per thread" % (n_s, B, cores),
                            "seconds": round(tc, 2), "gpu_matches_oracle_on_sample": same}
        else:
            # the stream's first pairs (no carried state) through the oracle's stream loop, single thread (the scan
            # is sequential); the GPU result of the same frames is recomputed for the comparison
            n_s = args.cpu_pairs if args.cpu_pairs > 0 else (3 if w > 1920 else 6)
            n_s = min(n_s, B)
            sample = np.ascontiguousarray(gray[np.arange(n_s + 1) % U])
            t0 = time.perf_counter()
            Ho, so, _ = O.stream_gray(sample, nfeatures=nfeat, force_max_iters=args.force_max_iters)
            tc = time.perf_counter() - t0
            Hc = torch.zeros(B, 9, dtype=torch.float64, device=dev); sc = torch.full((B,), -1, dtype=torch.int32, device=dev)
            ctxs[0].stream_homography_batch(first_chunk_frames[0], Hc, sc, nfeatures=nfeat, force_max_iters=args.force_max_iters)
            ctxs[0].synchronize()
            Hg = Hc.cpu().numpy().reshape(-1, 3, 3)[:n_s]
            same = bool(np.array_equal(sc.cpu().numpy()[:n_s], so)) and bool(np.allclose(Hg[so == 0], Ho[so == 0], rtol=1e-9, atol=1e-12))
            cpu_baseline = {"value": round(n_s / tc, 3), "unit": "frame-pair homographies/s", "cores": 1, "kind": "port",
                            "sample": "the stream's first %d pairs (gray frames) through the oracle's sequential stream loop, "
                                      "one thread" % n_s,
                            "seconds": round(tc, 2), "gpu_matches_oracle_on_sample": same}

    if rank == 0:
        chan = "BGR" if args.channels == 3 else "gray"
        if args.kind == "pairs":
            workload = ("synthetic %dx%d %s pair batch, %d independent pairs/step/GPU (%d distinct pairs, offset rotated "
                        "every step), ORB %d kp, RANSAC max 2000 conf 0.995 (BASELINE.json configs[1]%s)"
                        % (w, h, chan, B, U, nfeat, "; SURVEY 8d config 2: 256 pairs x 40 steps" if args.config == 2 else
                           "; throughput record at 1024 pairs/step, see profiles/r02_bench_cfg2.json for 256 x 40"))
            par = "pairs sharded, dp%d" % world
        elif args.kind == "stream":
            workload = ("synthetic %dx%d %s STREAM, one stream per GPU, %d pairs (%d frames) per step, running superposition "
                        "carried on the device, ORB %d kp, RANSAC max 2000%s (BASELINE.json configs[%d])"
                        % (w, h, chan, B, B + 1, nfeat, " forced (all 2000 samples)" if args.force_max_iters else " adaptive, conf 0.995",
                           2 if args.config == 3 else 4))
            par = "one stream per GPU, dp%d" % world
        else:
            workload = ("synthetic %dx%d %s STREAM sharded over %d GPU(s), two-phase: phase 1 on %d pairs per rank, static rows "
                        "all-gathered (RCCL), sequential scan over all %d pairs on every rank, ORB %d kp (BASELINE.json configs[3])"
                        % (w, h, chan, world, B, world * B, nfeat))
            par = "one stream, phase 1 sharded dp%d, scan replicated" % world
        out = {
            "metric": "frame-pair homographies/sec @720p" if (w, h) == (1280, 720) else "frame-pair homographies/sec",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "bench_config": args.config,
                       "pairs_per_step_per_gpu": B, "unique_pairs": U if args.kind == "pairs" else None,
                       "parallelism": par, "contexts_per_gpu": NCTX, "async_solve": not args.sync_solve,
                       "pairs_ok_fraction": ok_frac, "smooth_passes": args.smooth,
                       "fast_threshold_sharing_in_pair": not args.no_temporal, "fast_threshold_hint_across_calls": not args.no_temporal,
                       "no_temporal_value": no_temporal_value,
                       "keypoint_order": "OpenCV 3.4.2 retainBest on libstdc++ nth_element/partition (the reference's; every FAST corner at threshold 20 is scored)" if args.order == "reference" else "canonical (all ties, level/y/x)",
                       "canonical_order_value": canonical_order_value,
                       "lm_solver": args.solver, "fast_solver_value": fast_solver_value,
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
