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
