#!/usr/bin/env python3
"""1x1 GEMM kernel (csrc/gemm1x1.hip, ABI tiles 22 / 23) against the im2col split tile (16) on the 1x1 layer shapes of the C2
forward at batch 16: error against float64 and time per launch (HIP events over `--reps` launches; `--cold` touches a 512 MB buffer
between launches so the operands come from HBM as they do inside a forward).  GPU only.
  python tools/gemm1x1_probe.py [--reps 20] [--cold] [--tiles 16,22,23]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gpu_util as G  # noqa: E402
from sr3_hip import lib as L  # noqa: E402

# name, B, C0, C1, H, W, Cout, act, res
SHAPES = [
    ('qkv_16', 16, 512, 0, 16, 16, 1536, 1, False),
    ('out_16', 16, 512, 0, 16, 16, 512, 0, True),
    ('rc_256_512_16', 16, 256, 0, 16, 16, 512, 0, False),
    ('rc_1024_512_16', 16, 512, 512, 16, 16, 512, 0, False),
    ('rc_768_512_16', 16, 512, 256, 16, 16, 512, 0, False),
    ('rc_128_256_32', 16, 128, 0, 32, 32, 256, 0, False),
    ('rc_768_256_32', 16, 512, 256, 32, 32, 256, 0, False),
    ('rc_512_256_32', 16, 256, 256, 32, 32, 256, 0, False),
    ('rc_384_256_32', 16, 256, 128, 32, 32, 256, 0, False),
    ('rc_64_128_64', 16, 64, 0, 64, 64, 128, 0, False),
    ('rc_384_128_64', 16, 256, 128, 64, 64, 128, 0, False),
    ('rc_256_128_64', 16, 128, 128, 64, 64, 128, 0, False),
    ('rc_192_128_64', 16, 128, 64, 64, 64, 128, 0, False),
    ('qkv_8', 16, 512, 0, 8, 8, 1536, 1, False),
    ('out_8', 16, 512, 0, 8, 8, 512, 0, True),
    ('rc_1024_512_8', 16, 512, 512, 8, 8, 512, 0, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--cold', action='store_true')
    ap.add_argument('--tiles', default='16,22,23')
    ap.add_argument('--only', default='')
    ap.add_argument('--ksplit', type=int, default=0)
    ap.add_argument('--no-check', action='store_true')
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(',')]
    lib = L.load()
    d = G.dev()
    flush = torch.empty(128 << 20, dtype=torch.float32, device=d) if a.cold else None
    print('%-18s %s' % ('shape', '  '.join('t%-2d us (TF)   maxerr ' % t for t in tiles)))
    for name, B, C0, C1, H, W, Cout, act, res in SHAPES:
        if a.only and a.only not in name:
            continue
        g = torch.Generator().manual_seed(5)
        Cin = C0 + C1
        x0 = torch.randn(B, H, W, C0, generator=g)
        x1 = torch.randn(B, H, W, C1, generator=g) if C1 else None
        w = torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5        # OHWI
        bias = torch.randn(Cout, generator=g)
        ss = torch.stack([1.0 + 0.3 * torch.randn(B, Cin, generator=g), 0.5 * torch.randn(B, Cin, generator=g)], 2).contiguous() if act else None
        r0 = torch.randn(B, H, W, Cout, generator=g) if res else None
        to = lambda t: None if t is None else t.to(d)
        x0d, x1d, wd, bd, ssd, r0d = to(x0), to(x1), to(w), to(bias), to(ss), to(r0)
        ref = None
        if not a.no_check:
            xx = (x0d if x1d is None else torch.cat([x0d, x1d], 3)).double()
            if act:
                xx = xx * ssd[:, None, None, :, 0].double() + ssd[:, None, None, :, 1].double()
            ref = xx.reshape(-1, Cin) @ wd.reshape(Cout, Cin).double().t() + bd.double()
            if res:
                ref = ref + r0d.reshape(-1, Cout).double()
        out = torch.empty(B, H, W, Cout, device=d)
        flops = 2.0 * B * H * W * Cin * Cout
        cells = []
        for t in tiles:
            nb = int(lib.sr3_conv_scratch_bytes(B, H, W, Cin, Cout, 1, t, a.ksplit))
            scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device=d)

            def call():
                return lib.sr3_conv_f32(L.ptr(x0d), C0, L.ptr(x1d), C1, B, H, W, 0, 1, 1, Cout, L.ptr(wd), L.ptr(bd), L.ptr(ssd), act,
                                        None, 0, L.ptr(r0d), Cout if res else 0, None, 0, L.ptr(out), None, t, a.ksplit,
                                        L.ptr(scratch), nb, G.stream())
            out.fill_(float('nan'))
            rc = call()
            if rc != 0:
                cells.append('  refused             ')
                continue
            torch.cuda.synchronize()
            err = float('nan') if ref is None else (out.reshape(-1, Cout).double() - ref).abs().max().item()
            # the per-op entry re-derives the pre-split weights on every call (k_split_weights): time the GEMM kernel alone
            # through a start event placed by the profiler would need rocprof; here: events around the call, minus the
            # derive-only time measured the same way on a Cout = 128 slice is not exact -- so report the whole call too
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * a.reps)]
            for r in range(a.reps):
                if flush is not None:
                    flush.add_(1.0)
                ev[2 * r].record()
                call()
                ev[2 * r + 1].record()
            torch.cuda.synchronize()
            ts = sorted(ev[2 * r].elapsed_time(ev[2 * r + 1]) * 1e3 for r in range(a.reps))
            med = ts[len(ts) // 2]
            cells.append('%7.1f (%5.1f) %8.1e' % (med, flops / med * 1e-6, err))
        print('%-18s %s' % (name, '  '.join(cells)), flush=True)


if __name__ == '__main__':
    main()
