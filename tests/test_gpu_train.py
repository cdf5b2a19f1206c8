"""Training step parity: engine forward+backward+Adam vs the reference's autograd step recorded in
tests/golden (loss, every parameter gradient, weights after one Adam step; dropout 0, injected t / gamma / z).
Tolerances: loss rel 1e-5, gradients normwise rel 1e-4, post-Adam weights see below."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import DESCS, SCHEDS, CONDITIONAL, load_golden, opt_for      # noqa: E402
import gpu_util as G                                             # noqa: E402


def build_train(name):
    import model as Model
    opt = opt_for(name, phase='train', gpu=True)
    m = Model.create_model(opt)
    g, sd = load_golden(name)
    m.netG.load_state_dict(sd, strict=True)
    return m, g, sd


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_uncond'])
def test_loss_and_gradients_match_reference_autograd(name):
    m, g, sd = build_train(name)
    d = G.dev()
    data = {'HR': torch.from_numpy(g['loop/hr']).to(d), 'SR': torch.from_numpy(g['loop/sr']).to(d)}
    z = torch.from_numpy(g['train/z']).to(d)
    if DESCS[name]['variant'] == 'sr3':
        loss = m.netG.p_losses(data, noise=z, gamma=torch.from_numpy(g['train/gamma']))
    else:
        loss = m.netG.p_losses(data, noise=z, t=torch.from_numpy(g['train/t']).to(d))
    torch.cuda.synchronize()
    ref_loss = float(g['train/loss_sum'])
    assert abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss), (float(loss), ref_loss)
    worst = []
    for key, grad in m.netG.denoise_fn.named_gradients():
        ref = torch.from_numpy(g['grad/denoise_fn.' + key])
        got = grad.cpu()
        assert got.shape == ref.shape, key
        num = (got - ref).norm().item()
        den = max(ref.norm().item(), 1e-7)
        worst.append((num / den, key, den))
    worst.sort(reverse=True)
    bad = [w for w in worst if w[0] > 1e-4 and w[2] > 1e-6]
    assert not bad, 'gradient mismatch (rel err, key, |ref|): %s' % bad[:8]


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny'])
def test_training_step_is_bitwise_reproducible(name):
    """Two evaluations of the same training step (same weights, batch, noise, dropout seed) give the same loss and the same
    gradient arena BIT for bit: every reduction on the path has a fixed order -- since round 6 the attention backward's dK / dV too
    (per-query-block slabs summed in order instead of fp32 atomics, attention_bwd.hip).  Both tiny networks carry attention blocks."""
    m, g, sd = build_train(name)
    d = G.dev()
    un = m.netG.denoise_fn
    assert any(o['kind'] == 60 for o in un.plan.op_list(2)), 'no attention op in this network: nothing to test'
    data = {'HR': torch.from_numpy(g['loop/hr']).to(d), 'SR': torch.from_numpy(g['loop/sr']).to(d)}
    z = torch.from_numpy(g['train/z']).to(d)
    m.netG.train()
    outs = []
    for _ in range(3):
        if DESCS[name]['variant'] == 'sr3':
            loss = m.netG.p_losses(data, noise=z, gamma=torch.from_numpy(g['train/gamma']), drop_seed=1234)
        else:
            loss = m.netG.p_losses(data, noise=z, t=torch.from_numpy(g['train/t']).to(d), drop_seed=1234)
        torch.cuda.synchronize()
        outs.append((float(loss), un.grad_arena.clone()))
    for l, ga in outs[1:]:
        assert l == outs[0][0]
        assert torch.equal(ga, outs[0][1]), 'gradient arenas differ between two runs of the same step: max |diff| %.3g' % float((ga - outs[0][1]).abs().max())


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_uncond'])
def test_optimize_parameters_one_adam_step(name):
    """feed_data -> optimize_parameters (RNG draws patched to the recorded ones) -> weights after Adam."""
    m, g, sd = build_train(name)
    d = G.dev()
    netG = m.netG
    z = torch.from_numpy(g['train/z']).to(d)
    orig = netG.p_losses
    if DESCS[name]['variant'] == 'sr3':
        netG.p_losses = lambda x_in, noise=None: orig(x_in, noise=z, gamma=torch.from_numpy(g['train/gamma']))
    else:
        netG.p_losses = lambda x_in, noise=None: orig(x_in, noise=z, t=torch.from_numpy(g['train/t']).to(d))
    m.feed_data({'HR': torch.from_numpy(g['loop/hr']), 'SR': torch.from_numpy(g['loop/sr'])})
    m.optimize_parameters()
    assert abs(m.get_current_log()['l_pix'] - float(g['train/l_pix'])) <= 1e-5 * abs(float(g['train/l_pix']))
    out = netG.state_dict()
    # Adam's first step moves every weight by ~lr * sign(g): compare the *update*; entries whose
    # reference gradient is ~0 have an ill-conditioned sign and are excluded
    tot = bad = 0
    for k, ref_new in g.items():
        if not k.startswith('adam1/'):
            continue
        key = k[len('adam1/'):]
        ref_new = torch.from_numpy(ref_new)
        old = sd[key]
        grad = torch.from_numpy(g['grad/' + key])
        mask = grad.abs() > 1e-6 * max(grad.abs().max().item(), 1e-12) + 1e-9
        upd = (out[key].cpu() - old)[mask]
        ref_upd = (ref_new - old)[mask]
        tot += mask.sum().item()
        bad += ((upd - ref_upd).abs() > 2e-6).sum().item()
    assert tot > 1000 and bad <= 1e-4 * tot, (bad, tot)


def test_dropout_training_step_matches_oracle_autograd():
    """Train-mode dropout (p = 0.2): the engine's counter-based mask is restated in the oracle, whose torch
    autograd then gives reference loss and gradients for the same mask."""
    from oracle import sr3_oracle as O
    import model as Model
    name = 'sr3_tiny'
    opt = opt_for(name, phase='train', gpu=True)
    opt['model']['unet']['dropout'] = 0.2
    m = Model.create_model(opt)
    g, sd = load_golden(name)
    m.netG.load_state_dict(sd, strict=True)
    m.netG.train()
    d = G.dev()
    hr, sr = torch.from_numpy(g['loop/hr']), torch.from_numpy(g['loop/sr'])
    z, gamma = torch.from_numpy(g['train/z']), torch.from_numpy(g['train/gamma'])
    seed = 987654321
    loss = m.netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma, drop_seed=seed)
    torch.cuda.synchronize()
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and k.startswith('denoise_fn.')) for k, v in sd.items()}
    ref_loss = O.p_losses_sr3(sdr, DESCS[name], hr, sr, gamma, z, conditional=True, dropout=(0.2, seed))
    (ref_loss / hr.numel()).backward()
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    nodrop = float(g['train/loss_sum'])
    assert abs(float(loss) - nodrop) > 1e-3 * nodrop          # the mask really changed the forward
    bad = []
    for key, grad in m.netG.denoise_fn.named_gradients():
        ref = sdr['denoise_fn.' + key].grad
        num = (grad.cpu() - ref).norm().item()
        den = max(ref.norm().item(), 1e-7)
        if num / den > 1e-4 and den > 1e-6:
            bad.append((num / den, key))
    assert not bad, sorted(bad, reverse=True)[:8]
    # eval mode ignores dropout: sampling path unchanged
    m.netG.eval()


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny'])
def test_l2_loss_training_step_matches_oracle_autograd(name):
    """GaussianDiffusion(loss_type='l2') -> nn.MSELoss(reduction='sum') (set_loss, diffusion.py:84-90): loss and every
    gradient against the oracle's autograd.  (define_G hard-codes 'l1', so the reference's callers never take this path.)"""
    from oracle import sr3_oracle as O
    import model as Model
    m = Model.create_model(opt_for(name, phase='train', gpu=True))
    g, sd = load_golden(name)
    m.netG.load_state_dict(sd, strict=True)
    m.netG.loss_type = 'l2'
    m.netG.set_loss(G.dev())
    d = G.dev()
    hr, sr = torch.from_numpy(g['loop/hr']), torch.from_numpy(g['loop/sr'])
    z = torch.from_numpy(g['train/z'])
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and k.startswith('denoise_fn.')) for k, v in sd.items()}
    if name == 'sr3_tiny':
        gamma = torch.from_numpy(g['train/gamma'])
        loss = m.netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma)
        ref_loss = O.p_losses_sr3(sdr, DESCS[name], hr, sr, gamma, z, conditional=True, loss_type='l2')
    else:
        t = torch.from_numpy(g['train/t'])
        loss = m.netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), t=t)
        ref_loss = O.p_losses_ddpm(sdr, DESCS[name], O.schedule_tables(SCHEDS[name]), hr, sr, t, z, conditional=False, loss_type='l2')
    torch.cuda.synchronize()
    (ref_loss / hr.numel()).backward()
    assert abs(float(loss) - float(ref_loss.detach())) <= 1e-5 * abs(float(ref_loss.detach()))
    assert abs(float(loss) - float(g['train/loss_sum'])) > 1e-3 * float(g['train/loss_sum'])      # not the L1 value
    bad = []
    for key, grad in m.netG.denoise_fn.named_gradients():
        ref = sdr['denoise_fn.' + key].grad
        num, den = (grad.cpu() - ref).norm().item(), max(ref.norm().item(), 1e-7)
        if num / den > 1e-4 and den > 1e-6:
            bad.append((num / den, key))
    assert not bad, sorted(bad, reverse=True)[:8]
    m.netG.loss_type = 'l1'
    m.netG.set_loss(d)
    assert m.netG.denoise_fn.plan.options['loss_l2'] == 0


def test_dropout_mask_statistics():
    from oracle import sr3_oracle as O
    mk = O.dropout_mask((4, 32, 16, 16), 0.2, 12345, 7)
    keep = (mk > 0).float().mean().item()
    assert abs(keep - 0.8) < 0.01 and abs(mk.max().item() - 1.25) < 1e-6
    mk2 = O.dropout_mask((4, 32, 16, 16), 0.2, 12346, 7)
    assert (mk != mk2).float().mean().item() > 0.2


def test_data_parallel_path_world1_equals_plain_step():
    """The DP machinery (gradient-ready events, side-stream bucket all-reduce over RCCL) with one rank must
    reproduce the plain step bit for bit."""
    import os
    import torch.distributed as dist
    m, g, sd = build_train('sr3_tiny')
    d = G.dev()
    data = {'HR': torch.from_numpy(g['loop/hr']).to(d), 'SR': torch.from_numpy(g['loop/sr']).to(d)}
    z = torch.from_numpy(g['train/z']).to(d)
    gamma = torch.from_numpy(g['train/gamma'])
    un = m.netG.denoise_fn
    l0 = float(m.netG.p_losses(data, noise=z, gamma=gamma))
    g0 = un.grad_arena.clone()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        un.force_dp = True
        from sr3_hip.dist import GradReducer
        un._reducer = GradReducer(un.arena.numel(), d, dist, bucket_bytes=64 << 10)     # several buckets
        un.grad_arena.zero_()
        l1 = float(m.netG.p_losses(data, noise=z, gamma=gamma))
        torch.cuda.synchronize()
        assert l1 == l0 and torch.equal(un.grad_arena, g0)
        assert len(un._reducer.buckets) >= 4
    finally:
        un.force_dp = False
        dist.destroy_process_group()
