#!/usr/bin/env python3
"""Diagnostic (GPU box): who is closer to float64 on the batch-B training step -- the engine or the fp32 CPU oracle?
Runs the engine's train step (dropout p), then torch autograd over the oracle in fp32 and in fp64 (chunks of 8), and
prints normwise relative errors of the worst parameter gradients against the fp64 result.
  python tools/train_error_probe.py --batch 64 --dropout 0.2"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch                                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dropout', type=float, default=0.2)
    ap.add_argument('--config', default='sr3_16_128')
    ap.add_argument('--chunk', type=int, default=8)
    ap.add_argument('--top', type=int, default=12)
    ap.add_argument('--offset', type=int, default=0, help='first sample of the seeded 64-sample set to use')
    ap.add_argument('--skip-f32', action='store_true', help='skip the fp32 CPU oracle (its column then repeats the f64 one)')
    ap.add_argument('--variant', action='append', default=[], help='extra engine runs with plan options, e.g. ksplit=1,fuse_stats=0')
    a = ap.parse_args()
    from oracle import sr3_oracle as O
    if a.config in ('sr3_tiny', 'sr3_seam'):
        from helpers import DESCS, opt_for
        import model.networks as networks
        opt = opt_for(a.config, phase='train', gpu=True)
        opt['model']['unet']['dropout'] = a.dropout
        torch.manual_seed(17)
        netG = networks.define_G(opt)
        sd = {k: v.clone() for k, v in netG.state_dict().items()}
        netG = netG.to('cuda:0')
        netG.set_loss('cuda:0')
        netG.set_new_noise_schedule(opt['model']['beta_schedule']['train'], torch.device('cuda:0'))
        desc, c = DESCS[a.config], dict(size=16)
    else:
        from test_gpu_bench_configs import _build
        netG, sd, desc, opt, c = _build(a.config, phase='train', seed=17, dropout=a.dropout)
    netG.train()
    d = torch.device('cuda:0')
    B, S, seed = a.batch, c['size'], 20240607
    g = torch.Generator().manual_seed(8)
    NB = max(64, B + a.offset)
    hr = (torch.rand(NB, 3, S, S, generator=g) * 2 - 1)[a.offset:a.offset + B]
    sr = (torch.rand(NB, 3, S, S, generator=g) * 2 - 1)[a.offset:a.offset + B]
    z = torch.randn(NB, 3, S, S, generator=g)[a.offset:a.offset + B]
    gamma = (torch.rand(NB, generator=g) * 0.9 + 0.05)[a.offset:a.offset + B]
    print('gamma', [round(float(x), 3) for x in gamma])
    loss = netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma, drop_seed=seed)
    torch.cuda.synchronize()
    got = {k: v.cpu().double() for k, v in netG.denoise_fn.named_gradients()}
    variants = {}
    for var in a.variant:
        for kv in var.split(','):
            k, v = kv.split('=')
            netG.denoise_fn.plan.set_option(k, int(v))
        netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma, drop_seed=seed)
        torch.cuda.synchronize()
        variants[var] = {k: v.cpu().double() for k, v in netG.denoise_fn.named_gradients()}
        for kv in var.split(','):
            netG.denoise_fn.plan.set_option(kv.split('=')[0], 0 if kv.split('=')[0] != 'fuse_stats' else 1)
    res = {}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        if tag == 'f32' and a.skip_f32:
            continue
        t0 = time.time()
        sdr = {k: (v.to(dt) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and k.startswith('denoise_fn.'))
               for k, v in sd.items()}
        tot = 0.0
        for lo in range(0, B, a.chunk):
            sl = slice(lo, lo + a.chunk)
            drop = (a.dropout, seed, lo) if a.dropout > 0 else None
            l = O.p_losses_sr3(sdr, desc, hr[sl].to(dt), sr[sl].to(dt), gamma[sl].to(dt), z[sl].to(dt), conditional=True, dropout=drop)
            (l / hr.numel()).backward()
            tot += float(l.detach())
        res[tag] = ({k[len('denoise_fn.'):]: v.grad.double() for k, v in sdr.items() if v.grad is not None}, tot)
        print('%s oracle: loss %.6f  (%.0f s)' % (tag, tot, time.time() - t0), flush=True)
    print('engine loss %.6f' % float(loss))
    ref = res['f64'][0]
    if a.skip_f32:
        res['f32'] = res['f64']
    rows = []
    for k, r in ref.items():
        den = max(r.norm().item(), 1e-30)
        rows.append(((got[k] - r).norm().item() / den, (res['f32'][0][k] - r).norm().item() / den,
                     (got[k] - res['f32'][0][k]).norm().item() / den, k))
    rows.sort(reverse=True)
    print('worst by engine-vs-f64:  engine_vs_f64  oracle32_vs_f64  engine_vs_oracle32  key')
    for r in rows[:a.top]:
        print('  %.2e  %.2e  %.2e  %s' % r)
    rows.sort(key=lambda r: -r[1])
    print('worst by oracle32-vs-f64:')
    for r in rows[:8]:
        print('  %.2e  %.2e  %.2e  %s' % r)
    for var, gv in variants.items():
        vr = sorted((((gv[k] - r).norm().item() / max(r.norm().item(), 1e-30)), k) for k, r in ref.items())
        print('variant %s: worst engine_vs_f64 %.2e (%s), median %.2e' % (var, vr[-1][0], vr[-1][1], vr[len(vr) // 2][0]))
    import statistics
    print('median engine_vs_f64 %.2e, median oracle32_vs_f64 %.2e' % (statistics.median(r[0] for r in rows),
                                                                      statistics.median(r[1] for r in rows)))


if __name__ == '__main__':
    main()
