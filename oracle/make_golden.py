#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from
/root/reference, CPU, fp32) on seeded inputs.  Run in the build container only:

    python oracle/make_golden.py            # writes tests/golden/{sr3_tiny,ddpm_tiny,sr3_seam}.npz

The reference cannot travel to the GPU box, so the vectors are committed; this script is
the provenance.  Nothing in the reference is modified: RNG draws are made reproducible by
temporarily swapping ``torch.randn`` / ``torch.randn_like`` for functions that replay a
pre-drawn sequence (harness-side monkeypatch) and by seeding numpy's global RNG.
"""
import os
import sys
import contextlib

import numpy as np
import torch

REF = os.environ.get('SR3_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import model.networks as networks            # noqa: E402  (reference package)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def make_opt(which, in_ch, inner, groups, mults, attn_res, res_blocks, image, T, conditional, phase,
             lin=(1e-6, 1e-2)):
    sched = dict(schedule='linear', n_timestep=T, linear_start=lin[0], linear_end=lin[1])
    return {
        'phase': phase, 'gpu_ids': None, 'distributed': False,
        'model': {
            'which_model_G': which, 'finetune_norm': False,
            'unet': dict(in_channel=in_ch, out_channel=3, inner_channel=inner, norm_groups=groups,
                         channel_multiplier=mults, attn_res=attn_res, res_blocks=res_blocks, dropout=0),
            'beta_schedule': {'train': dict(sched), 'val': dict(sched)},
            'diffusion': dict(image_size=image, channels=3, conditional=conditional),
        },
    }


@contextlib.contextmanager
def replay_randn(seq):
    """Make torch.randn / randn_like return the tensors in seq, in order."""
    it = iter(seq)
    o_randn, o_like = torch.randn, torch.randn_like

    def randn(*a, **k):
        return next(it).clone()

    def randn_like(x, **k):
        z = next(it).clone()
        assert z.shape == x.shape
        return z
    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = o_randn, o_like


def build(name, opt, batch, seed, with_train=True, scale_weights=None):
    torch.manual_seed(seed)
    np.random.seed(seed)
    torch.set_num_threads(1)
    mo = opt['model']
    which = mo['which_model_G']
    cond = mo['diffusion']['conditional']
    S = mo['diffusion']['image_size']
    T = mo['beta_schedule']['train']['n_timestep']
    netG = networks.define_G(opt)                 # phase 'train' => orthogonal init
    if scale_weights is not None:                 # make GN affine / biases non-trivial
        with torch.no_grad():
            for k, p in netG.named_parameters():
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * scale_weights)
    netG.set_loss('cpu')
    netG.set_new_noise_schedule(mo['beta_schedule']['train'], 'cpu')
    netG.eval()
    out = {}
    sd = netG.state_dict()
    for k, v in sd.items():
        out['sd/' + k] = v.detach().numpy().copy()
    out['meta/T'] = np.int64(T)
    if which == 'sr3':
        out['meta/host_sqrt_alphas_cumprod_prev'] = np.asarray(netG.sqrt_alphas_cumprod_prev, dtype=np.float64)

    # ---- one UNet forward with per-layer taps -------------------------------------------
    in_ch = mo['unet']['in_channel']
    x = torch.randn(batch, in_ch, S, S).clamp(-3, 3)
    if which == 'sr3':
        tm = torch.rand(batch, 1) * 0.98 + 0.01
    else:
        tm = torch.randint(0, T, (batch,)).long()
    taps = {}
    hooks = []
    un = netG.denoise_fn
    for grp in ('downs', 'mid', 'ups'):
        for i, m in enumerate(getattr(un, grp)):
            hooks.append(m.register_forward_hook(
                lambda mod, inp, o, key='%s.%d' % (grp, i): taps.__setitem__(key, o.detach().numpy().copy())))
    with torch.no_grad():
        eps = un(x, tm)
    for h in hooks:
        h.remove()
    out['unet/x'] = x.numpy()
    out['unet/time'] = tm.numpy()
    out['unet/eps'] = eps.numpy()
    for k, v in taps.items():
        out['unet/tap/' + k] = v

    # ---- reverse loop with replayed noise -----------------------------------------------
    sr = torch.rand(batch, 3, S, S) * 2 - 1
    hr = torch.rand(batch, 3, S, S) * 2 - 1
    x_T = torch.randn(batch, 3, S, S)
    zs = [torch.randn(batch, 3, S, S) for _ in range(T)]        # zs[i] consumed at step i
    # order of draws in the reference loop: x_T, then step T-1 ... (ddpm draws at t == 0 too)
    order = [x_T] + [zs[i] for i in reversed(range(T)) if (i > 0 or which == 'ddpm')]
    for cont in (True, False):
        with replay_randn(order), torch.no_grad():
            if cond:
                r = netG.super_resolution(sr, continous=cont)
            else:
                r = netG.sample(batch_size=batch, continous=cont)
        out['loop/ret_continous' if cont else 'loop/ret_last'] = r.numpy()
    # single steps (first, middle, last) from a fixed state
    xs = torch.randn(batch, 3, S, S)
    for t in sorted({T - 1, T // 2, 0}):
        with replay_randn([zs[t]]), torch.no_grad():
            if which == 'sr3':
                r = netG.p_sample(xs, t, condition_x=sr if cond else None)
            else:
                r = netG.p_sample(xs, torch.full((batch,), t, dtype=torch.long),
                                  condition_x=sr if cond else None)
        out['step/%d' % t] = r.numpy()
    out['loop/sr'] = sr.numpy()
    out['loop/hr'] = hr.numpy()
    out['loop/x_T'] = x_T.numpy()
    out['loop/zs'] = torch.stack(zs).numpy()
    out['step/x'] = xs.numpy()

    # ---- training step: p_losses + backward + Adam (dropout 0) --------------------------
    if with_train:
        netG.train()
        z = torch.randn(batch, 3, S, S)
        data = {'HR': hr, 'SR': sr}
        if which == 'sr3':
            np.random.seed(seed + 1)
            t_draw = np.random.randint(1, T + 1)
            gam = np.random.uniform(netG.sqrt_alphas_cumprod_prev[t_draw - 1],
                                    netG.sqrt_alphas_cumprod_prev[t_draw], size=batch)
            out['train/t'] = np.int64(t_draw)
            out['train/gamma'] = torch.FloatTensor(gam).numpy()
            np.random.seed(seed + 1)
            with replay_randn([z]):
                loss = netG(data)
        else:
            torch.manual_seed(seed + 1)
            t_draw = torch.randint(0, T, (batch,)).long()
            out['train/t'] = t_draw.numpy()
            torch.manual_seed(seed + 1)
            with replay_randn([z]):
                loss = netG(data)
        optim = torch.optim.Adam(list(netG.parameters()), lr=1e-4)
        optim.zero_grad()
        l_pix = loss.sum() / int(hr.numel())
        l_pix.backward()
        out['train/z'] = z.numpy()
        out['train/loss_sum'] = loss.detach().numpy()
        out['train/l_pix'] = l_pix.detach().numpy()
        for k, p in netG.named_parameters():
            out['grad/' + k] = p.grad.detach().numpy().copy()
        optim.step()
        for k, p in netG.named_parameters():
            out['adam1/' + k] = p.detach().numpy().copy()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'params', sum(p.numel() for p in netG.parameters()), '->', path,
          '%.1f KB' % (os.path.getsize(path) / 1024))


def build_all(only=None):
    want = lambda n: only is None or n in only
    # sr3_tiny: conditional SR3, 3 levels, attention at 8x8 (+ mid), concat seams 16+16, 16+8 ...
    if want('sr3_tiny'):
      build('sr3_tiny', make_opt('sr3', 6, 8, 4, [1, 2, 2], [8], 1, 16, 8, True, 'train'),
            batch=2, seed=1234, scale_weights=0.1)
    # ddpm_tiny: unconditional DDPM variant (timestep embedding, per-sample t)
    if want('ddpm_tiny'):
      build('ddpm_tiny', make_opt('ddpm', 3, 8, 4, [1, 2], [8], 2, 16, 6, False, 'train', lin=(1e-4, 2e-2)),
            batch=2, seed=4321, scale_weights=0.1)
    # sr3_seam: default 32 groups, inner 32 -> GroupNorm groups straddle the concat seam
    # (64+32 = 96 channels / 32 groups = 3 per group), attention at 8x8.
    if want('sr3_seam'):
      build('sr3_seam', make_opt('sr3', 6, 32, 32, [1, 2], [8], 1, 16, 4, True, 'train'),
            batch=2, seed=99, with_train=False, scale_weights=0.1)
    # sr3_uncond: UNCONDITIONAL SR3 (config/sample_sr3_128.json's shape: which_model_G sr3, in_channel 3, conditional
    # false): sample() starts from noise, snapshots accumulate from x_T, continous=False returns ret_img[-1]
    # (sr3 diffusion.py:180-187); p_losses feeds x_noisy alone (:238-239)
    if want('sr3_uncond'):
      build('sr3_uncond', make_opt('sr3', 3, 8, 4, [1, 2, 2], [8], 1, 16, 8, False, 'train'),
            batch=2, seed=777, scale_weights=0.1)


if __name__ == '__main__':
    build_all(set(sys.argv[1:]) or None)
