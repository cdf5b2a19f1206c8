"""N > 1 path on CPU: 2 processes over gloo exercise the sharding / gather / MAX-timing helpers that
bench.py and multi-GPU sampling use (no collective sits in the data path itself)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from helpers import ROOT

PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, ret):
    sys.path.insert(0, PKG)
    import torch.distributed as dist
    from sr3_hip import dist as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    cond = torch.arange(n_items * 3 * 2 * 2, dtype=torch.float32).view(n_items, 3, 2, 2)
    seen = []

    def fake_sampler(c):            # stands in for netG.super_resolution on this rank's shard
        seen.append(c.shape[0])
        return c * 2 + 1
    out = D.sample_sharded(fake_sampler, cond, dist=dist, gather=True)
    ok = torch.equal(out, cond * 2 + 1)
    lo, hi = D.shard_range(n_items, rank, world)
    import time
    dt = D.timed_region(lambda: time.sleep(0.05 * (rank + 1)), dist=dist)
    ret[rank] = (bool(ok), seen, (lo, hi), dt)
    dist.destroy_process_group()


@pytest.mark.parametrize('n_items', [5, 8, 1])
def test_sharded_sampling_two_ranks(n_items):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_items, ret), nprocs=world, join=True)
    assert len(ret) == world
    covered = []
    for r in range(world):
        ok, seen, (lo, hi), dt = ret[r]
        assert ok
        covered += list(range(lo, hi))
        assert seen == ([hi - lo] if hi > lo else [])
        assert dt >= 0.09            # MAX over ranks: the slower rank slept 0.10 s
    assert covered == list(range(n_items))
    assert abs(ret[0][3] - ret[1][3]) < 1e-9     # both ranks report the same (max) time


def test_shard_range_balanced():
    sys.path.insert(0, PKG)
    from sr3_hip import dist as D
    for n in (0, 1, 7, 16, 17):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1


def _dp_worker(rank, world, port, n, ret):
    sys.path.insert(0, PKG)
    import torch.distributed as dist
    from sr3_hip import dist as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(n, generator=g)
    loss = torch.tensor([float(rank + 1)])
    red = D.GradReducer(n, torch.device('cpu'), dist, bucket_bytes=4096)
    red.reduce(grad, extra=[loss])
    ret[rank] = (grad.clone(), float(loss))
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_two_ranks():
    """The DP gradient exchange (tail-first buckets over the flat arena) == a plain sum over ranks."""
    sys.path.insert(0, PKG)
    from sr3_hip import dist as D
    n = 10007
    b = D.bucket_ranges(n, 4096)
    assert b[0][1] == n and b[-1][0] == 0 and all(b[i][0] == b[i + 1][1] for i in range(len(b) - 1))
    assert all(hi - lo <= 1024 for lo, hi in b)
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, n, ret), nprocs=world, join=True)
    expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    for r in range(world):
        g, l = ret[r]
        assert torch.allclose(g, expect, atol=1e-6)
        assert l == 3.0


class _IndexDataset(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {'Index': i}


def _loader_worker(rank, world, port, n, bs, ret):
    sys.path.insert(0, PKG)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import data as Data
    from sr3_hip.dist import dp_world_size
    dl = Data.create_dataloader(_IndexDataset(n), dict(batch_size=bs, use_shuffle=True, num_workers=0), 'train')
    epochs = []
    for _ in range(2):
        seen, sizes = [], []
        for batch in dl:
            seen += batch['Index'].tolist()
            sizes.append(len(batch['Index']))
        epochs.append((seen, sizes))
    ret[rank] = (epochs, dp_world_size(), dl.batch_size)
    dist.destroy_process_group()


def test_rank_sharded_training_loader_two_ranks():
    """create_dataloader under torch.distributed: `batch_size` stays the global batch (DataParallel semantics), each rank
    draws batch_size / world samples from a disjoint shard, shards reshuffle together every epoch."""
    world, n, bs = 2, 24, 8
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_loader_worker, args=(world, port, n, bs, ret), nprocs=world, join=True)
    for e in range(2):
        a, b = ret[0][0][e][0], ret[1][0][e][0]
        assert not (set(a) & set(b)) and sorted(a + b) == list(range(n))
        assert all(s == bs // world for s in ret[0][0][e][1])
    assert ret[0][0][0][0] != ret[0][0][1][0]              # set_epoch: a new permutation each pass
    assert ret[0][1] == ret[1][1] == 2 and ret[0][2] == bs // world
    # an indivisible global batch is refused rather than silently changed
    sys.path.insert(0, PKG)


def test_single_process_loader_unchanged():
    sys.path.insert(0, PKG)
    import data as Data
    dl = Data.create_dataloader(_IndexDataset(10), dict(batch_size=4, use_shuffle=False, num_workers=0), 'train')
    assert [b['Index'].tolist() for b in dl] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_bench_refuses_inconsistent_rank_counts():
    """bench.py --gpus N: it launches the ranks itself only when N devices are visible, and under a launcher it refuses a
    WORLD_SIZE that differs from --gpus -- it never prints a line whose n_gpus is not the number of ranks that ran."""
    import subprocess
    bench = os.path.join(ROOT, 'bench.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '1'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and b'refusing' in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr[-300:])
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '1'], env=dict(env, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and b'WORLD_SIZE=4' in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr[-300:])


def test_bench_two_ranks_end_to_end_stub_engine():
    """`python bench.py --gpus 2` end to end on CPU (SR3_BENCH_STUB=1: gloo instead of RCCL, the engine replaced by stand-ins;
    the launcher, the rank checks, the barrier / MAX-over-ranks timing, the record and the training leg's tail-first bucket
    walk over the REAL parameter count are the code the driver's 8-GPU run takes): one JSON line with n_gpus 2, global batch
    32, and a training leg whose gradient arena really was summed over both ranks, 13 buckets per step."""
    import json
    import subprocess
    bench = os.path.join(ROOT, 'bench.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['SR3_BENCH_STUB'] = '1'
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '5', '--warmup', '2', '--train-steps', '2', '--train-batch', '2',
                        '--no-roofline', '--no-exact-leg', '--no-cpu-baseline', '--no-torch-baseline', '--no-other-configs'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd='/tmp')
    assert r.returncode == 0, r.stderr[-600:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-600:]                      # rank 0 prints, rank 1 does not
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 5 and rec['warmup'] == 2 and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == 32 and rec['config']['batch_per_gpu'] == 16
    assert rec['metric'].startswith('SR3 16->128 images/sec') and rec['unit'] == 'images/s' and rec['higher_is_better'] is True
    assert abs(rec['value'] - 2 * 16 / (2000 * rec['ms_per_step'] * 1e-3)) < 1e-6 * rec['value']      # whole-job aggregate
    assert rec['data'].startswith('STUB')                       # a stub line can never pass for a measurement
    tr = rec['train']
    assert tr['stub'] and tr['global_batch'] == 4 and tr['batch_per_gpu'] == 2 and tr['steps'] == 2
    assert tr['gradient_buckets_per_step'] == 12 and tr['gradient_buckets_walked'] == 12 * (2 + 2)     # 391 MB in 32 MB buckets
    assert tr['l_pix_last'] == 2.0                               # every gradient is 1.0 on each rank: the all-reduce summed them
    assert abs(tr['value'] - 2 * 2 / (tr['ms_per_step'] * 1e-3)) < 1e-6 * tr['value']


def _valwave_worker(rank, world, port, shapes, ret):
    sys.path.insert(0, PKG)
    import torch.distributed as dist
    from sr3_hip import dist as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)

    class Net(object):                  # stands in for the reverse loop: result depends on the item and on `continous`
        calls = []
        seeds = []

        def super_resolution(self, cond, continous, item_seeds=None):
            Net.calls.append(tuple(cond.shape))
            Net.seeds.append(None if item_seeds is None else list(item_seeds))
            img = cond * 3 + 1
            return torch.cat([cond, img], 0) if continous else img[-1]
    conds = [torch.arange(1 * 3 * h * w, dtype=torch.float32).view(1, 3, h, w) + 100 * k for k, (h, w) in enumerate(shapes)]
    out = {}
    torch.manual_seed(77 + rank)        # (the ranks' own default seeds differ: the per-item seeds follow rank 0's)
    for continous in (True, False):
        wave = D.ValWave(conds, first_item=10)
        out[continous] = [wave.result(Net(), pos, continous).clone() for pos in range(len(conds))]
    ret[rank] = (out, Net.calls, Net.seeds)
    dist.destroy_process_group()


@pytest.mark.parametrize('shapes', [[(4, 4), (8, 6)], [(8, 8)], [(2, 2), (2, 2)]])
def test_validation_wave_mixed_shapes_and_ragged_wave_two_ranks(shapes):
    """ValWave (validation / inference items dealt over the ranks): items of DIFFERENT resolutions in one wave and a last
    wave with fewer items than ranks both come back complete on every rank, each chain run exactly once, on one rank."""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_valwave_worker, args=(world, port, shapes, ret), nprocs=world, join=True)
    sys.path.insert(0, PKG)
    from sr3_hip import dist as D
    for r in range(world):
        out, calls, seeds = ret[r]
        mine = [(1, 3, h, w) for k, (h, w) in enumerate(shapes) if k == r]
        assert calls == mine * 2, (r, calls)                 # once per `continous` flavour, own item only
        # per-item noise streams: the item's seed comes from its place in the validation sequence and RANK 0's base, not from the rank it ran on
        assert seeds == [[D.val_item_seed(10 + k, 0, base=77)] for k in range(len(shapes)) if k == r] * 2, (r, seeds)
        for k, (h, w) in enumerate(shapes):
            cond = torch.arange(3 * h * w, dtype=torch.float32).view(1, 3, h, w) + 100 * k
            assert torch.equal(out[True][k], torch.cat([cond, cond * 3 + 1], 0))
            assert torch.equal(out[False][k], (cond * 3 + 1)[-1])


def test_validation_wave_batches_equal_shapes_single_process():
    """ValWave without a process group (plain `infer.py`): consecutive items of equal shape run as ONE chain batch, odd shapes on
    their own, and every item gets back exactly what its own batch-1 call would return -- for both `continous` flavours (the
    reference's continous = False returns `ret_img[-1]`: the last image only)."""
    sys.path.insert(0, PKG)
    from sr3_hip import dist as D

    class Net(object):
        calls = []
        seeds = {}

        def super_resolution(self, cond, continous, item_seeds=None):                 # two "snapshots": the conditioning, then the result
            Net.calls.append(tuple(cond.shape))
            Net.seeds[tuple(cond.shape)] = item_seeds
            img = cond * 3 + 1
            return torch.cat([cond, img], 0) if continous else img[-1]
    shapes = [(4, 4), (4, 4), (8, 6), (4, 4), (2, 2)]
    conds = [torch.arange(3 * h * w, dtype=torch.float32).view(1, 3, h, w) + 100 * k for k, (h, w) in enumerate(shapes)]
    D._val_base[0] = None
    torch.manual_seed(4242)
    for continous in (True, False):
        Net.calls = []
        wave = D.ValWave(conds, first_item=32)
        got = [wave.result(Net(), pos, continous) for pos in range(len(conds))]
        assert sorted(Net.calls) == sorted([(3, 3, 4, 4), (1, 3, 8, 6), (1, 3, 2, 2)]), Net.calls
        # per-item noise streams (default on): image k of the validation sequence gets seed (base, k) whatever batch it rides in
        assert Net.seeds[(3, 3, 4, 4)] == [D.val_item_seed(32 + k, 0, base=4242) for k in (0, 1, 3)]
        assert Net.seeds[(1, 3, 8, 6)] == [D.val_item_seed(34, 0, base=4242)] and Net.seeds[(1, 3, 2, 2)] == [D.val_item_seed(36, 0, base=4242)]
        assert len({D.val_item_seed(k, j, base=b) for k in range(64) for j in range(2) for b in (4242, 4243)}) == 256
        for k, c in enumerate(conds):
            want = torch.cat([c, c * 3 + 1], 0) if continous else (c * 3 + 1)[-1]
            assert torch.equal(got[k], want), (continous, k)
    # SR3_VAL_ITEM_STREAMS=0 (or streams=False): one draw per batch from the default generator -- no seeds are passed
    Net.seeds = {}
    wave = D.ValWave(conds, streams=False)
    wave.result(Net(), 0, False)
    assert all(v is None for v in Net.seeds.values()) and len(Net.seeds) == 3
    D._val_base[0] = None
