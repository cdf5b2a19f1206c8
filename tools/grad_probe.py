#!/usr/bin/env python3
"""Training-gradient error of the engine against FLOAT64 autograd over the oracle, evaluated on the GPU box.

The float64 reference is the oracle's own torch ops run on `cuda` in double (PyTorch-ROCm has no MIOpen path for
double, so F.conv2d falls back to its native im2col + dgemm: exact enough to be the yardstick and ~10x faster
than the 256-core CPU run tools/train_error_probe.py does).  For every engine variant (plan options) the table is the
normwise relative error of each parameter gradient against that reference; `--f32` adds the same oracle in fp32 on
cuda (stock PyTorch-ROCm / MIOpen) and `--f32-cpu` on the host (oneDNN) as the "reference's own fp32 noise" columns.

  python tools/grad_probe.py --batch 64 --gamma uniform --data-seed 8 \
      --variant default --variant winograd=0 --kink-margin 1e-4
"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch                                      # noqa: E402

DEFAULTS = dict(winograd=1, wino_split=1, ksplit=0, fuse_stats=1, fuse_res=1, tile_cfg=0)


def gammas(mode, B, g):
    u = torch.rand(B, generator=g)
    if mode == 'uniform':
        return u * 0.9 + 0.05                      # the seeded draw of tests/test_gpu_bench_configs.py
    if mode == 'low':
        return 0.05 + 0.01 * u                     # sqrt(alpha_bar) ~ 0.05: almost pure noise
    if mode == 'high':
        return 0.999 + 0.0009 * u                  # ~ 0.999: almost clean images
    if mode == 'mixed':                            # both ends and the middle in one batch
        out = u * 0.9 + 0.05
        out[0::4] = 0.05 + 0.01 * u[0::4]
        out[1::4] = 0.999 + 0.0009 * u[1::4]
        return out
    raise SystemExit('unknown gamma mode ' + mode)


def oracle_grads(O, sd, desc, hr, sr, gamma, z, p_drop, seed, chunk, dtype, device):
    import grad_ref as R
    return R.oracle_grads(O, sd, desc, 'sr3', hr, sr, z, dict(gamma=gamma, conditional=True), p_drop, seed, chunk, dtype, device)


def rel_errors(got, ref):
    import grad_ref as R
    return [(e, k) for e, k, _ in R.rel_errors(got, ref)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dropout', type=float, default=0.2)
    ap.add_argument('--chunk', type=int, default=8)
    ap.add_argument('--data-seed', type=int, default=8)
    ap.add_argument('--gamma', default='uniform')
    ap.add_argument('--top', type=int, default=8)
    ap.add_argument('--f32', action='store_true')
    ap.add_argument('--f32-cpu', action='store_true')
    ap.add_argument('--ref-device', default='cuda')
    ap.add_argument('--variant', action='append', default=[])
    ap.add_argument('--kink-margin', type=float, default=0.0,
                    help='> 0: nudge z away from the kinks of the L1 loss first (tests/grad_ref.py:dekink)')
    ap.add_argument('--ref-cache', default=None, help='file to keep the float64 reference gradients in between runs')
    a = ap.parse_args()
    from oracle import sr3_oracle as O
    from test_gpu_bench_configs import _build
    netG, sd, desc, opt, c = _build('sr3_16_128', phase='train', seed=17, dropout=a.dropout)
    netG.train()
    d = torch.device('cuda:0')
    B, S, seed = a.batch, c['size'], 20240607
    g = torch.Generator().manual_seed(a.data_seed)
    hr = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    sr = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    z = torch.randn(B, 3, S, S, generator=g)
    gamma = gammas(a.gamma, B, g)
    print('batch %d, data seed %d, gamma mode %s: min %.4f max %.4f' % (B, a.data_seed, a.gamma, gamma.min(), gamma.max()), flush=True)
    if a.kink_margin > 0:
        import grad_ref as R
        t0 = time.time()
        z, moved, rmin = R.dekink(O, sd, desc, 'sr3', hr, sr, z, dict(gamma=gamma, conditional=True), a.dropout, seed, a.chunk,
                                  a.kink_margin)
        print('de-kinked z: %d of %d elements moved by %.0e, min |eps - z| now %.2e (%.0f s)'
              % (moved, z.numel(), 4 * a.kink_margin, rmin, time.time() - t0), flush=True)
    plan = netG.denoise_fn.plan
    eng = {}
    for var in (a.variant or ['default']):
        opts = dict(DEFAULTS)
        if var != 'default':
            for kv in var.split(','):
                k, v = kv.split('=')
                opts[k] = int(v)
        for k, v in opts.items():
            plan.set_option(k, v)
        t0 = time.time()
        try:
            loss = netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma, drop_seed=seed)
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(2):
                netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma, drop_seed=seed)
            torch.cuda.synchronize()
            ms = (time.time() - t1) / 2 * 1e3
            loss = netG.p_losses({'HR': hr.to(d), 'SR': sr.to(d)}, noise=z.to(d), gamma=gamma, drop_seed=seed)
            torch.cuda.synchronize()
        except Exception as e:                      # an option combination the plan refuses: report and go on
            print('engine %-40s FAILED: %s' % (var, e), flush=True)
            continue
        eng[var] = ({k: v.detach().clone() for k, v in netG.denoise_fn.named_gradients()}, float(loss), ms)
        print('engine %-40s loss %.6f  step %.1f ms (first call %.1f s)' % (var, float(loss), ms, t1 - t0), flush=True)
    if a.ref_cache and os.path.exists(a.ref_cache):
        t0 = time.time()
        ref, ref_loss = torch.load(a.ref_cache)
        ref = {k: v.cuda() for k, v in ref.items()}
        dt = time.time() - t0
    else:
        ref, ref_loss, dt = oracle_grads(O, sd, desc, hr, sr, gamma, z, a.dropout, seed, a.chunk, torch.float64, a.ref_device)
        if a.ref_cache:
            torch.save(({k: v.cpu() for k, v in ref.items()}, ref_loss), a.ref_cache)
    print('f64 oracle on %s: loss %.6f (%.0f s)' % (a.ref_device, ref_loss, dt), flush=True)
    others = {}
    if a.f32:
        g32, l32, dt = oracle_grads(O, sd, desc, hr, sr, gamma, z, a.dropout, seed, a.chunk, torch.float32, 'cuda')
        others['oracle fp32 on cuda (stock PyTorch-ROCm)'] = (g32, l32)
        print('f32 oracle on cuda: loss %.6f (%.0f s)' % (l32, dt), flush=True)
    if a.f32_cpu:
        g32, l32, dt = oracle_grads(O, sd, desc, hr, sr, gamma, z, a.dropout, seed, a.chunk, torch.float32, 'cpu')
        others['oracle fp32 on the host (oneDNN)'] = (g32, l32)
        print('f32 oracle on cpu: loss %.6f (%.0f s)' % (l32, dt), flush=True)
    for name, (gr, ls) in list(others.items()) + [('engine ' + k, (v[0], v[1])) for k, v in eng.items()]:
        rows = rel_errors(gr, ref)
        errs = [r[0] for r in rows]
        print('%-60s loss rel %.1e | worst %.2e  p90 %.2e  median %.2e' % (
            name, abs(ls - ref_loss) / abs(ref_loss), errs[0], errs[len(errs) // 10], statistics.median(errs)))
        for e, k in rows[:a.top]:
            print('      %.2e  %s' % (e, k))
    if others and eng:      # the engine against the fp32 oracle directly (what the pytest case compares)
        oname, (og, _) = next(iter(others.items()))
        for k, v in eng.items():
            rows = rel_errors(v[0], og)
            print('engine %s vs %s: worst %.2e (%s)' % (k, oname, rows[0][0], rows[0][1]))


if __name__ == '__main__':
    main()
