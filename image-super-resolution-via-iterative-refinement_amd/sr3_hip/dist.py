"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend 'nccl' is RCCL on ROCm).

The reference's only multi-GPU mechanism is single-process nn.DataParallel (model/networks.py:113-115, switched on
by core/logger.py:49-59 when more than one gpu id is listed).  Here the same callers (sr.py / infer.py / sample.py,
unchanged) are started once per GPU by `python -m torch.distributed.run`, and the drop-in packages join the job by
themselves: `bootstrap()` is called by `model.create_model` and `data.create_dataloader` (whichever the script reaches
first), picks `cuda:LOCAL_RANK`, creates the RCCL process group and gives every rank its own RNG stream.

* Training: the loader shards the GLOBAL batch over the ranks, `sr3_train_step` marks gradient buckets ready as the
  backward produces them and `GradReducer` all-reduces them on a side stream; every rank applies the same Adam step.
* Sampling / validation: every reverse chain is independent (GroupNorm is per sample, attention per image), so there
  is NO collective in the reverse loop; `ValWave` / `SampleWave` deal consecutive validation items (or `sample()`
  calls) round-robin over the ranks and all-gather only the finished images, so every rank's caller sees the same
  sequence of results a single process would produce, N images at a time.
"""
import os
import time

import torch

_BOOT = None          # (rank, world, local_rank) once bootstrap() has run
force_collectives = False     # SR3_DP=force: keep the gradient all-reduce on even with one rank (tests on a 1-GPU box)
resume_epoch = 0      # set by DDPM.load_network: the training loader (built before the model, sr.py:52-66) continues
                      # its per-epoch reshuffle sequence from here instead of replaying epoch 0's order after a resume


def _env_int(name, default):
    v = os.environ.get(name)
    return default if v in (None, '') else int(v)


def bootstrap():
    """Join the data-parallel job the launcher's environment describes; idempotent; returns (rank, world, local_rank).

    No WORLD_SIZE > 1 in the environment and no process group => (0, 1, 0) and nothing is touched.  Otherwise (torchrun /
    torch.distributed.run env: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT): the current device becomes
    `cuda:LOCAL_RANK` and a process group is created -- 'nccl' (= RCCL) bound to that device when a GPU is visible, 'gloo'
    otherwise (CPU tests) -- unless the caller already made one.  The torch RNG of every rank r > 0 is re-seeded to
    `initial_seed + r * 1000003` so ranks draw different z / dropout seeds; rank 0's generator is left exactly where the
    caller's own seeding and draws put it (re-seeding it would replay draws already made).  A script that seeds must do so
    BEFORE its first call into the drop-in packages: a `torch.manual_seed(s)` after this point gives every rank the same
    stream again.  (numpy's global RNG, which draws the SR3 training timestep, is entropy-seeded per process.)  On ranks
    > 0 the loggers the reference's scripts set up (`core/logger.py:setup_logger`: the root logger and 'val', both
    writing experiments/<name>_<time>/logs/*.log) are raised to WARNING, so the INFO stream -- option dump, l_pix lines,
    PSNR -- is written once, by rank 0, as checkpoints and images are.  `SR3_DP=0` opts out: the processes stay independent replicas;
    `SR3_DP=force` creates the group even for WORLD_SIZE=1 and keeps the collectives on (tests)."""
    global _BOOT
    import torch.distributed as tdist
    if _BOOT is not None and (_BOOT[1] == 1 or tdist.is_initialized()):
        return _BOOT
    if not tdist.is_available():
        _BOOT = (0, 1, 0)
        return _BOOT
    local = _env_int('LOCAL_RANK', 0)
    if tdist.is_initialized():
        _BOOT = (tdist.get_rank(), tdist.get_world_size(), local)
        return _BOOT
    world = _env_int('WORLD_SIZE', 1)
    mode = os.environ.get('SR3_DP', '1')
    force = mode == 'force' and 'WORLD_SIZE' in os.environ      # tests: a 1-rank job that still takes the collective path
    if (world <= 1 and not force) or mode == '0':
        _BOOT = (0, 1, 0)
        return _BOOT
    global force_collectives
    force_collectives = force
    rank = _env_int('RANK', 0)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # dmabuf IPC: RCCL needs it on this driver.  Only effective if HIP is not initialised yet in this process (the image
    # and the launcher export it anyway; this covers a bare environment)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if 'MASTER_PORT' not in os.environ:
        raise RuntimeError('WORLD_SIZE=%d but MASTER_PORT is not set: start the script with '
                           '`python -m torch.distributed.run --nproc-per-node N ...`' % world)
    if torch.cuda.is_available():
        n_dev = torch.cuda.device_count()
        if local >= n_dev:
            raise RuntimeError('rank %d wants cuda:%d but only %d device(s) are visible: list one gpu id per rank '
                               '(e.g. `-gpu %s`; core/logger.py exports it as CUDA_VISIBLE_DEVICES)'
                               % (rank, local, n_dev, ','.join(str(i) for i in range(world))))
        torch.cuda.set_device(local)
        tdist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
        tdist.init_process_group('gloo', rank=rank, world_size=world)
    if rank != 0:
        torch.manual_seed(torch.initial_seed() + rank * 1000003)
        import logging
        for name in (None, 'base', 'val'):
            logging.getLogger(name).setLevel(logging.WARNING)
    _BOOT = (rank, world, local)
    return _BOOT


def dp_info():
    """(rank, world, local_rank) without creating anything: the bootstrap result, else the live process group."""
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_rank(), tdist.get_world_size(), _env_int('LOCAL_RANK', 0)
    return 0, 1, 0


def dp_active():
    """True when the collective paths are on: more than one rank, or a forced 1-rank job (SR3_DP=force)."""
    import torch.distributed as tdist
    rank, world, _ = dp_info()
    return world > 1 or (force_collectives and tdist.is_available() and tdist.is_initialized())


def is_primary():
    """True on the rank that writes checkpoints / images / the network description (rank 0, or any single process)."""
    return dp_info()[0] == 0


def _coll_device():
    """Where a small collective's tensor has to live: the current GPU under RCCL, the host under gloo."""
    import torch.distributed as tdist
    if tdist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def broadcast_int(value, src=0):
    """One int64 from rank `src` to every rank (sampler seeds, begin_epoch)."""
    import torch.distributed as tdist
    if dp_info()[1] == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_coll_device())
    tdist.broadcast(t, src)
    return int(t.item())


def sync_replicas(unet, src=0):
    """Make every rank's parameters rank `src`'s: one broadcast of the packed arena (+ the DDPM frequency table).  The
    reference has one copy of the weights that DataParallel re-broadcasts every step (model/networks.py:113-115); here
    replicas persist, so they are equalised once -- after init / checkpoint load -- and stay equal because every rank
    applies the same Adam update to the same all-reduced gradient."""
    import torch.distributed as tdist
    if not dp_active():
        return
    arena = unet.arena.data
    if tdist.get_backend() == 'nccl' and not arena.is_cuda:
        raise RuntimeError('sync_replicas: move the model to its GPU first')
    tdist.broadcast(arena, src)
    tdist.broadcast(unet.freq, src)
    unet.weights_changed()


def barrier():
    import torch.distributed as tdist
    if dp_info()[1] > 1:
        tdist.barrier()


def dp_world_size():
    """Ranks that share one training step (1 outside torch.distributed)."""
    return dp_info()[1]


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
        self.measure_exposed = False       # bench.py: event pair on the compute stream around its wait for the side stream
        self._exposed = []

    def mark_args(self):
        """(n, offsets array, event-handle array) for sr3_train_step."""
        import ctypes as C
        if not self.cuda:
            return 0, None, None
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
            if self.measure_exposed:               # e0: the backward is done; e1: the last collective too -> exposed communication
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.stream)
                e1.record(cur)
                self._exposed.append((e0, e1))
            else:
                cur.wait_stream(self.stream)
        else:                                       # CPU tensors (gloo tests): same bucket walk, synchronous
            for lo, hi in self.buckets:
                d.all_reduce(grad_arena[lo:hi], op=d.ReduceOp.SUM)
            if extra is not None:
                for t in extra:
                    d.all_reduce(t, op=d.ReduceOp.SUM)


    def exposed_ms(self):
        """Mean time the compute stream waited for the side stream per step since measure_exposed was set (synchronises)."""
        if not self._exposed:
            return None
        torch.cuda.synchronize(self.device)
        v = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return sum(v) / len(v)


# ---- validation / unconditional sampling: consecutive items dealt round-robin over the ranks ----------------
def _gather_ragged(own, world, tdist, dev):
    """all_gather of one fp32 tensor per rank whose shapes may differ (or be absent: `own` None): the shapes travel
    first, the payloads are padded to the largest one.  Returns a list of `world` tensors (None for absent ones)."""
    meta = torch.zeros(8, dtype=torch.int64, device=_coll_device())
    if own is not None:
        if own.dim() > 7:
            raise ValueError('gather: tensors of more than 7 dimensions are not supported')
        meta[0] = own.dim()
        meta[1:1 + own.dim()] = torch.tensor(list(own.shape), dtype=torch.int64)
    metas = [torch.empty_like(meta) for _ in range(world)]
    tdist.all_gather(metas, meta)
    shapes = []
    for m in metas:
        m = m.tolist()
        shapes.append(None if m[0] == 0 else [int(v) for v in m[1:1 + int(m[0])]])
    numel = [0 if sh is None else int(torch.Size(sh).numel()) for sh in shapes]
    flat = torch.zeros(max(max(numel), 1), dtype=torch.float32, device=dev)
    if own is not None:
        flat[:own.numel()].copy_(own.reshape(-1))
    parts = [torch.empty_like(flat) for _ in range(world)]
    tdist.all_gather(parts, flat)
    return [None if sh is None else p[:n].view(sh) for p, sh, n in zip(parts, shapes, numel)]


def val_chain_batch():
    """Validation items one reverse chain carries (ValWave): the reference's validation loader hands `infer.py` / `sr.py` ONE image per
    call (data/__init__.py:18, batch_size = 1), and a batch-1 chain is launch-bound (bench.py: other_configs.sr3_16_128_b1) -- so
    consecutive items are run as one batch and handed back one by one.  SR3_VAL_CHAIN_BATCH overrides (1 = the reference's one
    chain per image)."""
    v = os.environ.get('SR3_VAL_CHAIN_BATCH')
    return max(1, int(v)) if v not in (None, '') else 16


def val_item_streams():
    """Per-item noise streams of the batched validation chains (default on; SR3_VAL_ITEM_STREAMS=0: one draw per batch from the
    default torch generator).  Item k of the validation sequence (its j-th image, if a loader batch holds several) draws x_T and
    every step's z from its own generator, seeded by `val_item_seed(k, j)`: the image it gets does not depend on SR3_VAL_CHAIN_BATCH,
    on the number of ranks or on which items share its chain -- up to the engine's rounding (kernels differ with the batch size),
    a batched infer.py run reproduces the one-chain-per-image run the reference makes (infer.py:67-71)."""
    return os.environ.get('SR3_VAL_ITEM_STREAMS', '1') not in ('0', '')


_val_base = [None]


def val_base_seed():
    """Base of the per-item seeds: SR3_VAL_SEED, else the default generator's seed (`torch.manual_seed(s)` in the calling script
    fixes it; unseeded it is torch's own random seed) -- rank 0's value on every rank."""
    if _val_base[0] is None:
        v = os.environ.get('SR3_VAL_SEED')
        base = int(v) if v not in (None, '') else int(torch.initial_seed())
        _val_base[0] = broadcast_int(base & 0x7FFFFFFFFFFFFFFF, 0) if dp_active() else base & 0x7FFFFFFFFFFFFFFF
    return _val_base[0]


def val_item_seed(item, image=0, base=None):
    """Seed of image `image` of validation item `item`: a 63-bit mix of (base, item, image) -- Philox keys need not be far apart,
    but neighbouring runs (base, base + 1) must not share streams item for item."""
    x = ((val_base_seed() if base is None else int(base)) + 0x9E3779B97F4A7C15 * (int(item) + 1) + 0xC2B2AE3D27D4EB4F * int(image)) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    return x & 0x7FFFFFFFFFFFFFFF


class ValWave(object):
    """Up to `world * chain` consecutive validation batches.  Every rank iterates ALL of them in order (so the caller's idx,
    file names and PSNR average are those of a single process -- sr.py:112-141, infer.py:64-90), but the reverse chains run
    once, when the first item of the wave is tested: the items are dealt to the ranks in contiguous runs, a rank runs its items
    of equal shape as ONE chain batch (round 6), the finished images are all-gathered and `DDPM.test` hands item k's to the
    caller when it gets there.  Rides in the batch dict under '_dp_wave'.  With one process the same object batches consecutive
    items into one chain.  Noise: every image draws from its own stream, seeded from its position in the validation sequence
    (`val_item_streams`, default on) -- the image does not depend on the batch it rides in; with SR3_VAL_ITEM_STREAMS=0 a batched
    chain takes one draw per batch from the torch RNG (each image a sample of the same distribution as its batch-1 chain's, not the
    same realisation; the reference does not seed)."""

    def __init__(self, conds, first_item=0, streams=None):
        self.conds = conds            # list of (B, 3, H, W) conditioning tensors ('SR' entries), one per item
        self.first_item = int(first_item)        # index of conds[0] in the validation sequence (per-item noise streams)
        self.streams = val_item_streams() if streams is None else bool(streams)
        self.results = None
        self.continous = None

    def to(self, *args, **kwargs):    # feed_data moves every dict entry with .to(device)
        return self

    @staticmethod
    def _run_items(netG, conds, continous, items=None):
        """The chains of `conds` (a list), equal shapes batched: returns one result per item, in the form
        `super_resolution(cond, continous)` has for that item alone.  `items`: the items' indices in the validation sequence
        (per-item noise streams), or None (one draw per batch)."""
        out = [None] * len(conds)
        groups = {}
        for i, c in enumerate(conds):
            groups.setdefault(tuple(c.shape), []).append(i)

        def seeds(idx):
            if items is None:
                return {}
            return dict(item_seeds=[val_item_seed(items[i], j) for i in idx for j in range(conds[i].shape[0])])
        for shape, idx in groups.items():
            if len(idx) == 1:
                out[idx[0]] = netG.super_resolution(conds[idx[0]], continous, **seeds(idx))
                continue
            b = [conds[i].shape[0] for i in idx]
            ret = netG.super_resolution(torch.cat([conds[i] for i in idx], dim=0), True, **seeds(idx))     # (snapshots * sum(b), C, H, W)
            n = sum(b)
            snaps = ret.view(ret.shape[0] // n, n, *ret.shape[1:])
            lo = 0
            for i, bi in zip(idx, b):
                own = snaps[:, lo:lo + bi]
                lo += bi
                # continous: the snapshots of this item stacked along dim 0; else the reference's `ret_img[-1]`: the LAST image
                out[i] = own.reshape(-1, *ret.shape[1:]).clone() if continous else own[-1, -1].clone()
        return out

    def run(self, netG, continous):
        rank, world, _ = dp_info()
        n = len(self.conds)
        items = [self.first_item + i for i in range(n)] if self.streams else None
        if items:
            val_base_seed()        # (a broadcast the first time: every rank takes part, also one without items in a ragged wave)
        if not dp_active():
            self.results = self._run_items(netG, self.conds, continous, items)
            self.continous = continous
            return
        import torch.distributed as tdist
        per = -(-n // world)                       # contiguous runs: rank r owns items [r * per, (r + 1) * per)
        mine = list(range(rank * per, min((rank + 1) * per, n)))
        own = self._run_items(netG, [self.conds[i] for i in mine], continous, [items[i] for i in mine] if items else None) if mine else []
        # items may differ in shape (an inference set with mixed resolutions) and a ragged last wave leaves ranks with fewer
        # items: per slot, shapes are gathered first, payloads padded to the largest
        dev = own[0].device if own else self.conds[0].device
        results = [None] * n
        for j in range(per):
            parts = _gather_ragged(own[j] if j < len(own) else None, world, tdist, dev)
            for r in range(world):
                if r * per + j < n:
                    results[r * per + j] = parts[r]
        self.results = results
        self.continous = continous

    def result(self, netG, pos, continous):
        if self.results is None or self.continous != continous:
            self.run(netG, continous)
        return self.results[pos]


class SampleWave(object):
    """The same dealing for `DDPM.sample()` (sample.py:104,140 call it `data_len` times): call k of a wave of `world`
    calls returns rank k's draw; the wave is sampled when its first call arrives."""

    def __init__(self):
        self.results = []
        self.key = None

    def next(self, netG, batch_size, continous):
        import torch.distributed as tdist
        rank, world, _ = dp_info()
        key = (int(batch_size), bool(continous))
        if not self.results or self.key != key:
            own = netG.sample(batch_size, continous)
            parts = [torch.empty_like(own) for _ in range(world)]
            tdist.all_gather(parts, own.contiguous())
            self.results = parts
            self.key = key
        return self.results.pop(0)
