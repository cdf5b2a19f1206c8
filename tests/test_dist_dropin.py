"""The data-parallel path reached THROUGH THE DROP-IN, the way a user reaches it: an unchanged caller script started
once per GPU by `python -m torch.distributed.run`.  tests/dp_dropin_worker.py replays sr.py's call sequence and never
touches torch.distributed itself; the launcher's environment variables are all the packages get.  (CPU, gloo, 2 ranks:
the engine calls are stand-ins, the distributed plumbing is the shipped code.  The RCCL counterpart is
tests/test_gpu_dist.py.)  Reference behaviour matched: nn.DataParallel's global-batch semantics
(model/networks.py:113-115, model/model.py:48-58), one writer for checkpoints (model/model.py:124-143)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, out_dir, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['SR3_VAL_CHAIN_BATCH'] = '1'         # one chain per validation image, as the reference: these tests count chains per rank
    env.update(extra_env or {})
    worker = os.path.join(ROOT, 'tests', 'dp_dropin_worker.py')
    if world == 1:
        cmd = [sys.executable, worker, str(out_dir)]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), worker, str(out_dir)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    return [torch.load(os.path.join(out_dir, 'rank%d.pt' % k), weights_only=False) for k in range(world)]


@pytest.mark.timeout(900)
def test_sr_py_call_sequence_under_torchrun_is_data_parallel(tmp_path):
    world = 2
    recs = _launch(world, tmp_path)
    a, b = recs
    assert (a['rank'], b['rank']) == (0, 1) and a['world'] == b['world'] == 2
    # per-rank RNG streams
    assert a['torch_seed'] != b['torch_seed']
    # replicas equalised at construction although every rank drew its own initial weights ...
    assert torch.equal(a['w_init'], b['w_init'])
    # ... and identical after every rank applied the same update to the all-reduced gradient
    assert torch.equal(a['w_final'], b['w_final'])
    assert not torch.equal(a['w_init'], a['w_final'])
    # the loader: batch_size 4 is the GLOBAL batch -> 2 per rank, disjoint shards that cover the set, reshuffled per epoch
    steps = len(a['seen'])
    assert steps == len(b['seen']) == 2 * (16 // 4)
    for e in range(2):
        ea = sum(a['seen'][e * 4:(e + 1) * 4], [])
        eb = sum(b['seen'][e * 4:(e + 1) * 4], [])
        assert all(len(s) == 2 for s in a['seen'])
        assert not (set(ea) & set(eb)) and sorted(ea + eb) == list(range(16))
    assert a['seen'][:4] != a['seen'][4:]
    # l_pix is the global-batch value on every rank
    assert a['l_pix'] == b['l_pix']
    # validation: both ranks walked all 5 items in order and hold the SAME images; item k was produced by rank k % 2;
    # each rank ran only its share of the reverse chains (3 + 2 instead of 5 + 5)
    assert [v[0] for v in a['val']] == [v[0] for v in b['val']] == [0, 1, 2, 3, 4]
    for va, vb in zip(a['val'], b['val']):
        assert torch.equal(va[1], vb[1]) and torch.equal(va[2], vb[2])
    assert len(a['sr_calls']) == 3 and len(b['sr_calls']) == 2
    # the stand-in image of item k is 0.5 * cond + (producing rank): a single process (rank 0 everywhere) gives the tag
    (tmp_path / 'single').mkdir()
    s = _launch(1, tmp_path / 'single')[0]
    assert s['world'] == 1 and len(s['seen'][0]) == 4 and len(s['sr_calls']) == 5
    for k in range(5):
        tag = (a['val'][k][1] - s['val'][k][1]).mean().item()
        assert abs(tag - (k % 2)) < 1e-6, (k, tag)
    # one writer: checkpoint present, images only from rank 0
    assert a['ckpt_exists_after_save'] and b['ckpt_exists_after_save']
    names = sorted(os.listdir(tmp_path))
    assert [n for n in names if n.endswith('rank1.png')] == []
    assert len([n for n in names if n.endswith('rank0.png')]) == 5
    sd = torch.load(os.path.join(tmp_path, 'I8_E1_gen.pth'), map_location='cpu')
    assert any(k.startswith('denoise_fn.') for k in sd)
    # global-batch semantics, against what the reference's single process computes on the union of the two shards
    # (model/model.py:52-53): l_pix = sum |.| over all 4 samples / (4*c*h*w); the stand-in gradient is
    # direction * sum(HR) * grad_scale, so the weights tell whether grad_scale was 1 / (GLOBAL b*c*h*w) and summed once
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from dp_dropin_worker import _Pairs, _cpu_u8_to_f32
    ds = _Pairs(16)

    def f32(i, key):
        return _cpu_u8_to_f32(ds[i][key][None], None, (-1, 1), None)[0]
    n = a['w_init'].numel()
    direction = torch.linspace(1.0, 2.0, n)
    w = a['w_init'].clone()
    for t in range(steps):
        idx = a['seen'][t] + b['seen'][t]
        numel = 4 * 3 * 16 * 16
        lp = sum(float((f32(i, 'HR') - f32(i, 'SR')).abs().double().sum()) for i in idx) / numel
        assert abs(a['l_pix'][t] - lp) <= 1e-5 * abs(lp), (t, a['l_pix'][t], lp)
        gsum = sum(float(f32(i, 'HR').double().sum()) for i in idx) / numel
        w -= 1e-4 * direction * gsum
    assert torch.allclose(w, a['w_final'], rtol=1e-5, atol=1e-7)


def test_opt_out_keeps_processes_independent(tmp_path):
    """SR3_DP=0: the launcher's environment is ignored, every process is a plain single-GPU run."""
    env = dict(os.environ, WORLD_SIZE='2', RANK='1', LOCAL_RANK='1', MASTER_PORT='1', SR3_DP='0')
    code = ("import sys; sys.path.insert(0, %r); from sr3_hip import dist as D; import torch.distributed as t; "
            "print(D.bootstrap(), t.is_initialized())" % os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'))
    r = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and r.stdout.decode().strip() == '(0, 1, 0) False', (r.stdout, r.stderr[-500:])
