#!/usr/bin/env python3
"""HIP-event time of EVERY launch of one UNet forward of the plan the bench runs (batch of the BASELINE.json config),
one row per plan op (+ one per split-K reduce), and the sums per op kind.  Usage on the GPU box:
    python tools/op_table.py [--config sr3_16_128] [--reps 5] [--opt key=value ...] > gpurun_out/op_table.txt"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')
KIND = {10: 'embed+FiLM', 20: 'conv_in', 30: 'GN stats', 40: 'GN fold', 50: 'conv', 60: 'attention', 70: 'conv_out'}
TILE = {1: 'im2col 128x128', 2: 'im2col 128x64', 3: 'im2col 64x64', 4: 'im2col 64x128', 5: 'halo 128x128', 6: 'halo 256x64',
        9: 'halo 256x128', 11: 'winograd', 12: 'winograd 3xbf16', 13: 'winograd 3xbf16 4w',
        14: 'im2col 3xbf16 128x128', 15: 'im2col 3xbf16 128x64', 16: 'im2col 3xbf16 64x64', 17: 'im2col 3xbf16 64x128',
        18: 'im2col 3xbf16 128x128 presplit-w', 19: 'im2col 3xbf16 128x64 presplit-w', 20: 'im2col 3xbf16 64x64 presplit-w',
        21: 'im2col 3xbf16 64x128 presplit-w', 22: 'gemm1x1 3xbf16 64x128', 23: 'gemm1x1 3xbf16 128x128'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='sr3_16_128')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--opt', action='append', default=[])
    a = ap.parse_args()
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import model.networks as networks
    from sr3_hip import lib as L
    cfg = bench.CONFIGS[a.config]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    netG = networks.define_G(bench.config_opt(a.config)).to(dev)
    un = netG.denoise_fn
    plan = un.plan
    for kv in a.opt:
        k, v = kv.split('=')
        plan.set_option(k, int(v))
    un.ensure_derived()
    B, S = (a.batch or cfg['batch']), cfg['size']
    x = torch.randn(B, 3, S, S, device=dev)
    cond = (torch.rand(B, 3, S, S, device=dev) * 2 - 1) if cfg['conditional'] else None
    lib = L.load()
    if plan.variant == 'sr3':
        level, tstep = torch.full((B,), 0.5, device=dev), None
    else:
        level, tstep = None, torch.full((B,), 1000, dtype=torch.long, device=dev)
    wsbuf, need = un._ws.get(plan, B, dev)
    out = torch.empty(B, 3, S, S, device=dev)
    max_ops = 4096
    ms = (C.c_float * max_ops)()
    kind = (C.c_int * max_ops)()
    fl = (C.c_double * max_ops)()
    n = C.c_int()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    acc = None
    for r in range(a.reps + 1):
        L.check(lib.sr3_unet_forward_profile(plan.handle, L.ptr(x), L.ptr(cond), 0 if cond is None else 3, L.ptr(level),
                                             L.ptr(tstep), L.ptr(un.freq), L.ptr(un.arena.data), L.ptr(wsbuf), need, L.ptr(out),
                                             B, stream, max_ops, ms, kind, fl, C.byref(n)))
        if r == 0:
            continue
        if acc is None:
            acc = [0.0] * n.value
        for i in range(n.value):
            acc[i] += ms[i] / a.reps
    ops = plan.op_list(B)
    j = -1
    sums = {}
    print('# %s batch %d: %d timed launches, %.3f ms summed (HIP events around every launch)' % (a.config, B, n.value, sum(acc)))
    for i in range(n.value):
        k = int(kind[i])
        if k != 59:
            j += 1
        o = ops[j]
        if k == 59:
            label = '   split-K reduce'
        elif o['kind'] == 50:
            label = '%dx%d%s %4d->%4d @%3dx%-3d %-16s ks%d%s' % (o['ksize'], o['ksize'], ' s2' if o['stride'] == 2 else (' up' if o['upsample'] else '   '),
                                                               o['cin'], o['cout'], o['h_out'], o['w_out'], TILE.get(o['tile_cfg'], str(o['tile_cfg'])),
                                                               o['ksplit'], ' +stats' if o['fused_output_stats'] else '')
        elif o['kind'] == 60:
            label = 'attention N=%d d=%d' % (o['h_out'] * o['w_out'], o['cin'])
        else:
            label = KIND.get(o['kind'], str(o['kind']))
        tf = fl[i] / (acc[i] * 1e-3) / 1e12 if acc[i] > 0 and fl[i] > 0 else 0.0
        print('%4d %4d  %8.1f us  %7.2f GF %6.1f TF  %s' % (j, k, acc[i] * 1e3, fl[i] / 1e9, tf, label))
        s = sums.setdefault(k, [0.0, 0, 0.0])
        s[0] += acc[i]
        s[1] += 1
        s[2] += fl[i]
    print('# per kind: kind  ms  launches  TF')
    for k, s in sorted(sums.items()):
        print('# %4d  %7.3f  %3d  %6.1f' % (k, s[0], s[1], s[2] / (s[0] * 1e-3) / 1e12 if s[0] > 0 else 0.0))


if __name__ == '__main__':
    main()
