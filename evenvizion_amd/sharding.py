"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI;
"gloo" in CPU tests).  Frame pairs (or whole streams) are independent units, so they are split in contiguous
blocks with no data-path collective; the only exchange is one all_gather of fixed-size per-pair records
{f64 H[9], f64 status} at the end of a batch.
"""
import torch


def shard_range(n_units, rank, world):
    """Contiguous block [lo, hi) of `n_units` owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_units), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_pair_records(H_local, status_local, n_total, group=None):
    """All ranks receive (H f64[n_total,9], status i32[n_total]) in global pair order.

    H_local f64[n_local,9], status_local i32[n_local] hold this rank's block (shard_range order)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return H_local.reshape(-1, 9), status_local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cap = -(-int(n_total) // world)                           # ceil: every rank sends a block of equal size
    rec = torch.zeros(cap, 10, dtype=torch.float64, device=H_local.device)
    n_local = H_local.shape[0]
    rec[:n_local, :9] = H_local.reshape(-1, 9)
    rec[:n_local, 9] = status_local.to(torch.float64)
    parts = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(parts, rec, group=group)
    H_all = torch.empty(n_total, 9, dtype=torch.float64, device=H_local.device)
    st_all = torch.empty(n_total, dtype=torch.int32, device=H_local.device)
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        H_all[lo:hi] = parts[r][:hi - lo, :9]
        st_all[lo:hi] = parts[r][:hi - lo, 9].to(torch.int32)
    return H_all, st_all


def stream_block(n_frames, rank, world):
    """Frames [lo, hi] (inclusive) of an n_frames stream that `rank` needs for ITS pairs: the n_frames-1 pairs are
    split with shard_range, pair p is (frame p, frame p+1), so consecutive blocks overlap by one frame.
    Returns (frame_lo, frame_hi_exclusive, pair_lo, pair_hi); an empty block has pair_lo == pair_hi."""
    p_lo, p_hi = shard_range(max(int(n_frames) - 1, 0), rank, world)
    if p_hi <= p_lo:
        return p_lo, p_lo, p_lo, p_hi
    return p_lo, p_hi + 1, p_lo, p_hi


def gather_static_rows(rows_local, counts_local, status_local, n_pairs_total, group=None):
    """Phase-1 results of every rank in global pair order on every rank: (rows f32[n_pairs_total,cap,4],
    counts i32[n_pairs_total], status1 i32[n_pairs_total]).  One all_gather per array (RCCL on GPUs); blocks are
    padded to the same number of pairs.  rows_local f32[n_local,cap,4] in shard_range order of the pairs."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows_local, counts_local, status_local
    world = dist.get_world_size(group)
    blk = -(-int(n_pairs_total) // world)
    cap = rows_local.shape[1]
    dev = rows_local.device
    n_local = rows_local.shape[0]
    rows = torch.zeros(blk, cap, 4, dtype=torch.float32, device=dev)
    meta = torch.zeros(blk, 2, dtype=torch.int32, device=dev)
    rows[:n_local] = rows_local
    meta[:n_local, 0] = counts_local
    meta[:n_local, 1] = status_local
    rows_parts = [torch.empty_like(rows) for _ in range(world)]
    meta_parts = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(rows_parts, rows, group=group)
    dist.all_gather(meta_parts, meta, group=group)
    rows_all = torch.empty(n_pairs_total, cap, 4, dtype=torch.float32, device=dev)
    counts_all = torch.empty(n_pairs_total, dtype=torch.int32, device=dev)
    status_all = torch.empty(n_pairs_total, dtype=torch.int32, device=dev)
    for r in range(world):
        lo, hi = shard_range(n_pairs_total, r, world)
        rows_all[lo:hi] = rows_parts[r][:hi - lo]
        counts_all[lo:hi] = meta_parts[r][:hi - lo, 0]
        status_all[lo:hi] = meta_parts[r][:hi - lo, 1]
    return rows_all, counts_all, status_all


def sharded_stream_homographies(static_fn, scan_fn, n_frames, group=None):
    """One stream over all ranks with the reference's semantics (two-phase).

    static_fn(frame_lo, frame_hi) -> (rows, counts, status1) for the pairs of frames [frame_lo, frame_hi) -- phase 1
    on this rank's GPU (Context.stream_static_batch on the rank's frames); scan_fn(rows, counts, status1) -> (H, status)
    -- phase 2, the sequential H_sup scan (Context.stream_scan), run redundantly on every rank over ALL pairs so that
    every rank ends with the full, identical result and no second exchange is needed."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    f_lo, f_hi, p_lo, p_hi = stream_block(n_frames, rank, world)
    rows, counts, status1 = static_fn(f_lo, f_hi)
    rows, counts, status1 = gather_static_rows(rows, counts, status1, n_frames - 1, group)
    return scan_fn(rows, counts, status1)
