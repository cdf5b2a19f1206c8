"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend 'nccl' is RCCL on ROCm).

The reference's only multi-GPU mechanism is single-process nn.DataParallel (model/networks.py:113-115).
Here sampling shards the image list across ranks -- every reverse chain is independent
(GroupNorm is per sample, attention per image), so there is NO collective in the data path; the
only collectives are an optional all_gather of the finished images and the barrier / MAX-reduce
around a timed region.  Each rank keeps a full weight replica (391 MB for SR3 16->128).
"""
import time

import torch


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of n_items for this rank (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def timed_region(fn, dist=None, device=None):
    """barrier + synchronize, run fn(), synchronize + barrier; returns the MAX elapsed seconds over ranks."""
    def sync():
        if device is not None and device.type == 'cuda':
            torch.cuda.synchronize(device)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def sample_sharded(sample_fn, cond_all, dist=None, gather=True):
    """Super-resolve a list of conditioning images across ranks.

    sample_fn(cond_shard) -> images of the shard (same leading dim); cond_all: (N, C, H, W) on every
    rank.  Returns the (N, ...) result on every rank when gather, else this rank's shard."""
    if dist is None:
        return sample_fn(cond_all)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = cond_all.shape[0]
    lo, hi = shard_range(n, rank, world)
    out = sample_fn(cond_all[lo:hi]) if hi > lo else cond_all.new_zeros((0,) + tuple(cond_all.shape[1:]))
    if not gather:
        return out
    # ragged all_gather: pad every shard to the largest one
    sizes = [shard_range(n, r, world) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = out.new_zeros((mx,) + tuple(out.shape[1:]))
    pad[:hi - lo] = out
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][:h - l] for r, (l, h) in enumerate(sizes)], dim=0)
