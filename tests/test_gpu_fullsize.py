"""Full-size BASELINE.json configurations on the GPU vs the CPU oracle (random-init weights drawn
in the reference's order), batch 1 so the oracle finishes in seconds:
  C2 SR3 16->128 (config/sr_sr3_16_128.json), C5 DDPM-128 (config/sample_ddpm_128.json),
  C4 SR3 64->512 (config/sr_sr3_64_512.json: norm_groups 16, N = 1024 / d = 1024 mid attention).
Tolerance: 2e-5 * max(1, |ref|_inf) per forward."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import gpu_util as G                                # noqa: E402

CONFIGS = {
    'sr3_16_128': dict(which='sr3', in_channel=6, inner=64, groups=None, mults=[1, 2, 4, 8, 8], attn=[16], rb=2, size=128,
                       cond=True),
    'ddpm_128': dict(which='ddpm', in_channel=3, inner=64, groups=None, mults=[1, 1, 2, 2, 4, 4], attn=[16], rb=2, size=128,
                     cond=False),
    'sr3_64_512': dict(which='sr3', in_channel=6, inner=64, groups=16, mults=[1, 2, 4, 8, 16], attn=[], rb=1, size=512,
                       cond=True),
}


def make_opt(c, T=2000):
    sched = dict(schedule='linear', n_timestep=T, linear_start=1e-6, linear_end=1e-2)
    unet = dict(in_channel=c['in_channel'], out_channel=3, inner_channel=c['inner'], channel_multiplier=c['mults'],
                attn_res=c['attn'], res_blocks=c['rb'], dropout=0)
    if c['groups']:
        unet['norm_groups'] = c['groups']
    return {'phase': 'train', 'gpu_ids': [0], 'distributed': False, 'path': {'checkpoint': '/tmp', 'resume_state': None},
            'train': {'optimizer': {'type': 'adam', 'lr': 1e-4}},
            'model': {'which_model_G': c['which'], 'finetune_norm': False, 'unet': unet,
                      'beta_schedule': {'train': dict(sched), 'val': dict(sched)},
                      'diffusion': dict(image_size=c['size'], channels=3, conditional=c['cond'])}}


@pytest.mark.parametrize('name', ['sr3_16_128', 'ddpm_128', 'sr3_64_512'])
def test_fullsize_forward_and_step_vs_oracle(name):
    from oracle import sr3_oracle as O
    import model.networks as networks
    c = CONFIGS[name]
    opt = make_opt(c)
    torch.manual_seed(11)
    netG = networks.define_G(opt)                       # orthogonal init, reference draw order
    sd = {k: v.clone() for k, v in netG.state_dict().items()}
    d = G.dev()
    netG = netG.to(d)
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['val'], d)
    netG.show_progress = False
    desc = O.desc_from_opt(opt)
    S = c['size']
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, c['in_channel'], S, S, generator=g)
    if c['which'] == 'sr3':
        t = torch.tensor([[0.7312]])
    else:
        t = torch.tensor([1234], dtype=torch.long)
    with torch.no_grad():
        ref = O.unet_forward(sd, desc, x, t)
    got = netG.denoise_fn(x.to(d), t.to(d)).cpu()
    err = G.assert_close(got, ref, what=name + ' eps')
    from helpers import experiments_built
    if name == 'sr3_16_128' and experiments_built():       # opt-in 3 x bf16 split MFMA path (experiment build): same stated tolerance
        netG.denoise_fn.plan.set_option('split_bf16', 1)
        got_s = netG.denoise_fn(x.to(d), t.to(d)).cpu()
        err_s = G.assert_close(got_s, ref, what=name + ' eps (split_bf16)')
        print('%s: split_bf16 eps max abs err %.2e (exact-fp32 path %.2e)' % (name, err_s, err))
        netG.denoise_fn.plan.set_option('split_bf16', 0)
    # one reverse step with injected noise through the public p_sample
    tab = O.schedule_tables(opt['model']['beta_schedule']['val'])
    xs = torch.randn(1, 3, S, S, generator=g)
    z = torch.randn(1, 3, S, S, generator=g)
    cond = x[:, :3] if c['cond'] else None
    tt = 1500
    with torch.no_grad():
        ref_step = O.p_sample(sd, desc, tab, xs, tt, z, condition_x=cond)
    if c['which'] == 'sr3':
        got_step = netG.p_sample(xs.to(d), tt, condition_x=None if cond is None else cond.to(d), noise=z.to(d))
    else:
        got_step = netG.p_sample(xs.to(d), torch.full((1,), tt, dtype=torch.long, device=d),
                                 condition_x=None if cond is None else cond.to(d), noise=z.to(d))
    G.assert_close(got_step.cpu(), ref_step, what=name + ' p_sample')
    print('%s: eps max abs err %.2e (|ref|max %.2f)' % (name, err, ref.abs().max().item()))


def test_config1_ten_step_loop_all_frames():
    """BASELINE.json configs[0]: config/sr_sr3_16_128.json, batch 1, 10 reverse steps (the `-debug` size):
    feed_data -> test(continous=True) through the drop-in DDPM wrapper, all 11 returned frames vs the oracle
    loop on the same x_T / z sequence."""
    from oracle import sr3_oracle as O
    import model as Model
    c = CONFIGS['sr3_16_128']
    opt = make_opt(c, T=10)
    opt['phase'] = 'val'
    torch.manual_seed(21)
    m = Model.create_model(opt)
    sd = {k: v.clone().cpu() for k, v in m.netG.state_dict().items()}
    m.netG.show_progress = False
    desc = O.desc_from_opt(opt)
    tab = O.schedule_tables(opt['model']['beta_schedule']['val'])
    g = torch.Generator().manual_seed(3)
    sr = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1
    hr = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1
    x_T = torch.randn(1, 3, 128, 128, generator=g)
    zs = torch.randn(10, 1, 3, 128, 128, generator=g)
    d = G.dev()
    m.feed_data({'HR': hr, 'SR': sr})
    out = m.netG.p_sample_loop(m.data['SR'], continous=True, x_T=x_T.to(d), noise_seq=zs.to(d)).cpu()
    with torch.no_grad():
        ref = O.p_sample_loop(sd, desc, tab, sr, x_T, zs, conditional=True, continous=True)
    assert tuple(out.shape) == tuple(ref.shape) == (11, 3, 128, 128)
    G.assert_close(out, ref, tol=1e-4, what='config 1 loop')
    # and the seeded production path (graph replay) returns the same shapes through test()
    m.test(continous=True)
    assert tuple(m.SR.shape) == (11, 3, 128, 128) and bool(torch.isfinite(m.SR).all())
    vis = m.get_current_visuals(need_LR=False)
    assert tuple(vis['SR'].shape) == (11, 3, 128, 128) and tuple(vis['INF'].shape) == (1, 3, 128, 128)


@pytest.mark.parametrize('name', ['sr3_16_128', 'ddpm_128'])
def test_fullsize_training_step_vs_oracle_autograd(name):
    """BASELINE.json configs[2] (SR3 16->128) and C5 (DDPM-128) at batch 1, dropout 0: loss and EVERY parameter gradient
    of the engine's forward + backward against torch autograd over the CPU oracle with the same (gamma | t, z).
    Tolerances: loss rel 1e-5, gradients normwise rel 1e-4."""
    from oracle import sr3_oracle as O
    import model.networks as networks
    c = CONFIGS[name]
    opt = make_opt(c)
    torch.manual_seed(17)
    netG = networks.define_G(opt)                       # phase 'train': orthogonal init in the reference's order
    sd = {k: v.clone() for k, v in netG.state_dict().items()}
    d = G.dev()
    netG = netG.to(d)
    netG.set_loss(d)
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['train'], d)
    netG.train()
    desc = O.desc_from_opt(opt)
    S = c['size']
    g = torch.Generator().manual_seed(8)
    hr = torch.rand(1, 3, S, S, generator=g) * 2 - 1
    sr = torch.rand(1, 3, S, S, generator=g) * 2 - 1
    z = torch.randn(1, 3, S, S, generator=g)
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and k.startswith('denoise_fn.')) for k, v in sd.items()}
    if c['which'] == 'sr3':
        gamma = torch.tensor([0.6180339])
        loss = netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma)
        ref_loss = O.p_losses_sr3(sdr, desc, hr, sr, gamma, z, conditional=True)
    else:
        t = torch.tensor([1234], dtype=torch.long)
        loss = netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), t=t)
        tab = O.schedule_tables(opt['model']['beta_schedule']['train'])
        ref_loss = O.p_losses_ddpm(sdr, desc, tab, hr, sr, t, z, conditional=False)
    torch.cuda.synchronize()
    (ref_loss / hr.numel()).backward()
    rl = float(ref_loss.detach())
    assert abs(float(loss) - rl) <= 1e-5 * abs(rl), (float(loss), rl)
    worst, n = [], 0
    for key, grad in netG.denoise_fn.named_gradients():
        ref = sdr['denoise_fn.' + key].grad
        num, den = (grad.cpu() - ref).norm().item(), max(ref.norm().item(), 1e-12)
        worst.append((num / den, key, den))
        n += 1
    worst.sort(reverse=True)
    bad = [w for w in worst if w[0] > 1e-4 and w[2] > 1e-7]
    assert n > 150 and not bad, bad[:8]
    print('%s: training step loss rel err %.1e, worst gradient rel err %.1e (%s) over %d tensors'
          % (name, abs(float(loss) - rl) / abs(rl), worst[0][0], worst[0][1], n))
