"""Data-parallel training step over RCCL, one process per GPU, against the CPU oracle on the CONCATENATED batch:
rank-summed gradients (all-reduced in tail-first buckets on a side stream) == d(sum-reduced loss / GLOBAL b*c*h*w) and
the logged l_pix == the reference's value over the global batch (model/model.py:52-53 under nn.DataParallel,
model/networks.py:113-115).  world = 1 runs on any GPU box (collective path forced on); world = 2 needs two GPUs and is
skipped otherwise.  Tolerances: loss rel 1e-5, gradients normwise rel 1e-4."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from helpers import DESCS, ROOT, load_golden, opt_for      # noqa: E402

PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')
NAME = 'sr3_tiny'
PER_RANK = 2


def _inputs(world):
    g = torch.Generator().manual_seed(321)
    n = world * PER_RANK
    hr = torch.rand(n, 3, 16, 16, generator=g) * 2 - 1
    sr = torch.rand(n, 3, 16, 16, generator=g) * 2 - 1
    z = torch.randn(n, 3, 16, 16, generator=g)
    gamma = torch.rand(n, generator=g) * 0.8 + 0.1
    return hr, sr, z, gamma


def _rank_main(rank, world, port, ret):
    for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        import model as Model
        from sr3_hip.dist import GradReducer
        opt = opt_for(NAME, phase='train', gpu=True)
        m = Model.create_model(opt)
        m.device = dev
        m.netG = m.netG.to(dev)
        m.netG.set_new_noise_schedule(opt['model']['beta_schedule']['train'], dev)
        _, sd = load_golden(NAME)
        m.netG.load_state_dict(sd, strict=True)
        un = m.netG.denoise_fn
        un.force_dp = True                                  # world 1: keep the collective path on
        un._reducer = GradReducer(un.arena.numel(), dev, dist, bucket_bytes=16 << 10)    # several buckets
        hr, sr, z, gamma = _inputs(world)
        sl = slice(rank * PER_RANK, (rank + 1) * PER_RANK)
        orig = m.netG.p_losses
        m.netG.p_losses = lambda x_in, noise=None: orig(x_in, noise=z[sl].to(dev), gamma=gamma[sl])
        m.feed_data({'HR': hr[sl].clone(), 'SR': sr[sl].clone()})
        before = un.arena.data.clone()
        m.optimize_parameters()
        torch.cuda.synchronize(dev)
        grads = {k: v.cpu().clone() for k, v in un.named_gradients()}
        ret[rank] = dict(l_pix=m.get_current_log()['l_pix'], grads=grads, buckets=len(un._reducer.buckets),
                         moved=float((un.arena.data - before).abs().max()), weights=un.arena.data.cpu().clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)          # a wedged collective must fail this test, not hang the suite
@pytest.mark.parametrize('world', [1, 2])
def test_dp_training_step_matches_oracle_on_the_global_batch(world):
    if torch.cuda.device_count() < world:
        pytest.skip('needs %d GPUs (have %d)' % (world, torch.cuda.device_count()))
    from oracle import sr3_oracle as O
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_rank_main, args=(world, port, ret), nprocs=world, join=True)
    hr, sr, z, gamma = _inputs(world)
    _, sd = load_golden(NAME)
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and k.startswith('denoise_fn.')) for k, v in sd.items()}
    ref_loss = O.p_losses_sr3(sdr, DESCS[NAME], hr, sr, gamma, z, conditional=True)
    ref_lpix = ref_loss / hr.numel()                        # GLOBAL b*c*h*w
    ref_lpix.backward()
    for r in range(world):
        out = ret[r]
        assert out['buckets'] >= 4
        assert abs(out['l_pix'] - float(ref_lpix)) <= 1e-5 * abs(float(ref_lpix)), (r, out['l_pix'], float(ref_lpix))
        bad = []
        for key, grad in out['grads'].items():
            ref = sdr['denoise_fn.' + key].grad
            num, den = (grad - ref).norm().item(), max(ref.norm().item(), 1e-7)
            if num / den > 1e-4 and den > 1e-6:
                bad.append((num / den, key))
        assert not bad, (r, sorted(bad, reverse=True)[:6])
        assert out['moved'] > 0                             # Adam ran after the reduction
    if world > 1:                                           # identical replicas after the step
        assert torch.equal(ret[0]['weights'], ret[1]['weights'])
