"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend 'nccl' is RCCL on ROCm).

The reference's only multi-GPU mechanism is single-process nn.DataParallel (model/networks.py:113-115).
Here sampling shards the image list across ranks -- every reverse chain is independent
(GroupNorm is per sample, attention per image), so there is NO collective in the data path; the
only collectives are an optional all_gather of the finished images and the barrier / MAX-reduce
around a timed region.  Each rank keeps a full weight replica (391 MB for SR3 16->128).
"""
import time

import torch


def dp_world_size(unet=None):
    """Ranks that share one training step (1 outside torch.distributed); `unet.force_dp` keeps the collective path
    on with a single rank (tests)."""
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_world_size()
    return 1


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


# ---- data-parallel training: bucketed gradient all-reduce overlapped with the backward ----------------
def bucket_ranges(n_floats, bucket_bytes=32 << 20):
    """Split the flat gradient arena [0, n) into contiguous buckets, returned from the TAIL (the backward
    finishes the last layers first): [(lo, hi), ...] with lo descending.  32 MB default: large enough to run
    a ring at link rate over xGMI (7 x ~153 GB/s point-to-point links), small enough to start early."""
    per = max(1, bucket_bytes // 4)
    out = []
    hi = n_floats
    while hi > 0:
        lo = max(0, hi - per)
        out.append((lo, hi))
        hi = lo
    return out


class GradReducer(object):
    """Sum-all-reduce of the gradient arena across ranks in tail-first buckets.  Each bucket waits only for
    its own gradient-ready event (recorded by sr3_train_step on the compute stream) and runs on a side
    stream, so communication overlaps the remaining backward kernels."""

    def __init__(self, n_floats, device, dist, bucket_bytes=32 << 20):
        self.dist = dist
        self.device = device
        self.buckets = bucket_ranges(n_floats, bucket_bytes)
        self.cuda = device.type == 'cuda'
        if self.cuda:
            self.stream = torch.cuda.Stream(device)
            self.events = [torch.cuda.Event() for _ in self.buckets]
            for ev in self.events:                 # materialise the hipEvent_t handles
                ev.record(torch.cuda.current_stream(device))

    def mark_args(self):
        """(n, offsets array, event-handle array) for sr3_train_step."""
        import ctypes as C
        n = len(self.buckets)
        offs = (C.c_size_t * n)(*[lo for lo, _ in self.buckets])
        evs = (C.c_void_p * n)(*[ev.cuda_event for ev in self.events])
        return n, offs, evs

    def reduce(self, grad_arena, extra=None):
        """Launch the bucket all-reduces (after sr3_train_step has been enqueued); the caller's stream then
        waits for them.  `extra`: small tensors (e.g. the loss scalar) reduced with the last bucket."""
        d = self.dist
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.stream):
                for (lo, hi), ev in zip(self.buckets, self.events):
                    self.stream.wait_event(ev)
                    d.all_reduce(grad_arena[lo:hi], op=d.ReduceOp.SUM)
                if extra is not None:
                    self.stream.wait_stream(cur)
                    for t in extra:
                        d.all_reduce(t, op=d.ReduceOp.SUM)
            cur.wait_stream(self.stream)
        else:                                       # CPU tensors (gloo tests): same bucket walk, synchronous
            for lo, hi in self.buckets:
                d.all_reduce(grad_arena[lo:hi], op=d.ReduceOp.SUM)
            if extra is not None:
                for t in extra:
                    d.all_reduce(t, op=d.ReduceOp.SUM)
