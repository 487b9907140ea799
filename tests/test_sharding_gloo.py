"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): pairs are sharded in contiguous blocks, each rank
fills its block, one all_gather of fixed-size records rebuilds the global (H, status) arrays on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from evenvizion_amd.sharding import gather_pair_records, shard_range, sharded_stream_homographies, stream_block


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 256, 1001):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    g = torch.Generator().manual_seed(1234)
    H_all = torch.randn(n_total, 9, dtype=torch.float64, generator=g)
    st_all = torch.randint(0, 6, (n_total,), dtype=torch.int32, generator=g)
    H, st = gather_pair_records(H_all[lo:hi].clone(), st_all[lo:hi].clone(), n_total)
    ok = bool(torch.equal(H, H_all) and torch.equal(st, st_all))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pair_records_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_single_process_passthrough():
    H = torch.zeros(3, 9, dtype=torch.float64); st = torch.zeros(3, dtype=torch.int32)
    H2, st2 = gather_pair_records(H, st, 3)
    assert H2.shape == (3, 9) and torch.equal(st2, st)


# ---- one stream over several ranks: phase 1 sharded with a one-frame overlap, rows gathered, phase 2 redundant ------
CAP = 5


def _fake_static(f_lo, f_hi):
    """stand-in for Context.stream_static_batch: rows that encode the pair's two frame indices"""
    n = max(f_hi - f_lo - 1, 0)
    rows = torch.zeros(n, CAP, 4, dtype=torch.float32)
    counts = torch.zeros(n, dtype=torch.int32)
    status = torch.zeros(n, dtype=torch.int32)
    for i in range(n):
        p = f_lo + i
        rows[i, :, 0] = p; rows[i, :, 1] = p + 1; rows[i, :, 2] = torch.arange(CAP); rows[i, :, 3] = 7 * p
        counts[i] = 1 + p % CAP
        status[i] = p % 3
    return rows, counts, status


def _fake_scan(rows, counts, status):
    """stand-in for Context.stream_scan: an order-dependent running quantity"""
    acc = torch.cumsum(rows[:, 0, 0].double() * 3 + counts.double() + status.double(), 0)
    return acc, status


def _stream_worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc, st = sharded_stream_homographies(_fake_static, _fake_scan, n_frames)
    want_acc, want_st = _fake_scan(*_fake_static(0, n_frames))
    q.put((rank, bool(torch.equal(acc, want_acc) and torch.equal(st, want_st))))
    dist.barrier()
    dist.destroy_process_group()


def test_stream_block_covers_every_pair_once():
    for n_frames in (2, 3, 9, 10, 101):
        for world in (1, 2, 3, 8):
            pairs = []
            for r in range(world):
                f_lo, f_hi, p_lo, p_hi = stream_block(n_frames, r, world)
                assert (p_hi == p_lo and f_hi == f_lo) or (f_lo == p_lo and f_hi == p_hi + 1)
                pairs += list(range(p_lo, p_hi))
            assert pairs == list(range(n_frames - 1))


def test_sharded_stream_world2():
    for n_frames in (10, 2):          # 2 frames = 1 pair: rank 1 owns an empty block
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(60)
        assert res == [(0, True), (1, True)]


# ---- bench.py's gather (one all_gather_into_tensor of the per-pair f64 H records per step) ---------------------------
def _bench_gather_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H = torch.full((B, 9), float(rank + 1), dtype=torch.float64) + torch.arange(B, dtype=torch.float64)[:, None]
    gathered = torch.zeros(world * B, 9, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, H)
    want = torch.cat([torch.full((B, 9), float(r + 1), dtype=torch.float64) + torch.arange(B, dtype=torch.float64)[:, None]
                      for r in range(world)])
    q.put((rank, bool(torch.equal(gathered, want))))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gather_path_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_gather_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


# ---- BASELINE configs[4] shape: one stream PER RANK, chunked, each rank carrying its own {H_sup, H_prev}; per step one
# all_gather of the ranks' H records (bench.py --config 5) ------------------------------------------------------------
def _per_rank_stream_worker(rank, world, port, B, nsteps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def chunk_result(r, step, state):
        """stand-in for Context.stream_homography_batch on rank r's own stream: an order-dependent running quantity
        carried in `state` between chunks (what the device keeps in state_in / state_out)"""
        H = torch.zeros(B, 9, dtype=torch.float64)
        for k in range(B):
            state = state * 1.0001 + (r + 1) * 0.5 + step * B + k
            H[k] = state
        return H, state

    ok = True
    state = torch.zeros((), dtype=torch.float64)
    want_states = [torch.zeros((), dtype=torch.float64) for _ in range(world)]
    gathered = torch.zeros(world * B, 9, dtype=torch.float64)
    for step in range(nsteps):
        H, state = chunk_result(rank, step, state)
        dist.all_gather_into_tensor(gathered, H)
        for r in range(world):                       # every rank holds every stream's chunk, in rank order
            Hw, want_states[r] = chunk_result(r, step, want_states[r])
            ok = ok and bool(torch.equal(gathered[r * B:(r + 1) * B], Hw))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_one_stream_per_rank_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_per_rank_stream_worker, args=(r, 2, port, 16, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


# ---- the same two paths over RCCL on real devices: needs two GPUs (the driver's 8-GPU node; skipped on a 1-GPU box) ----
def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      LOCAL_RANK=str(rank), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    from evenvizion_amd import synthetic as S
    from evenvizion_amd._lib import Context
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    frames, _ = S.make_stream(5, 7, 400, 224)
    n = len(frames)
    c = Context(device=rank, max_w=400, max_h=224, max_features=500, max_frames=n)
    d = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)

    def static_fn(f_lo, f_hi):
        return c.stream_static_batch(d[f_lo:f_hi])

    def scan_fn(rows, counts, st1):
        return c.stream_scan(rows, counts, st1)

    H, st = sharded_stream_homographies(static_fn, scan_fn, n)
    Hw = torch.zeros(n - 1, 9, dtype=torch.float64, device=dev); sw = torch.zeros(n - 1, dtype=torch.int32, device=dev)
    c.stream_homography_batch(d, Hw, sw)
    c.synchronize(); torch.cuda.synchronize(dev)
    ok = bool(torch.equal(H, Hw) and torch.equal(st, sw))
    gathered = torch.zeros(world * (n - 1), 9, dtype=torch.float64, device=dev)     # bench.py's gather of the H records
    dist.all_gather_into_tensor(gathered, Hw)
    ok = ok and bool(torch.equal(gathered[:n - 1], Hw) and torch.equal(gathered[n - 1:], Hw))
    q.put((rank, ok))
    dist.barrier()
    c.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_stream_and_gather_nccl_world2():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL world_size 2); this box exposes %d" %
                    (torch.cuda.device_count() if torch.cuda.is_available() else 0))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
    assert res == [(0, True), (1, True)]
