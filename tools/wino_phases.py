#!/usr/bin/env python3
"""Phase timeline of the persistent Winograd kernel (conv3x3_wino.hip, template DBG = 64; library built with
-DSR3_WINO_ABLATIONS): thread 0 of every workgroup stamps the shader clock at the phase boundaries of every tile.
    SR3_WINO_DBG=64 SR3_LIBRARY=.../libsr3_ablate.so python tools/wino_phases.py
Prints, per layer shape, the median duration of each phase and the gap between consecutive tiles of a workgroup."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'))
from sr3_hip import lib as L      # noqa: E402

PH = ['prologue: stage chunks 0/1, issue chunk 2, barrier', 'main loop', 'epi: issue bias/FiLM/residual loads (round 0)',
      'epi: barrier (all waves out of the main loop)', 'epi: round 0 LDS writes', 'epi: next tile part a + round 1 residual loads',
      'epi: barrier', 'epi: round 0 reads + combine + stores', 'epi: barrier + round 1 LDS writes', 'epi: next tile part b (chunk 1, ss, U)',
      'epi: barrier', 'epi: round 1 reads + combine + stores', 'epi: barrier', 'epi: statistics butterfly + park + barrier',
      'epi: statistics final + barrier']


def run(name, C0, C1, H, Cout, ups, act, B=16, res=True, stats=True):
    lib = L.load()
    d = torch.device('cuda:0')
    Cin = C0 + C1
    Ho = H << ups
    s0 = torch.randn(B, H, H, C0, device=d)
    s1 = torch.randn(B, H, H, C1, device=d) if C1 else None
    w = torch.randn(Cout, 9, Cin, device=d) * 0.02
    bias = torch.randn(Cout, device=d)
    ss = torch.randn(B, Cin, 2, device=d) if act else None
    r0 = torch.randn(B, Ho, Ho, Cout, device=d) if res else None
    out = torch.empty(B, Ho, Ho, Cout, device=d)
    T = int(lib.sr3_conv_stats_slices(B, H, H, ups, Cin, Cout, 11, 1))
    st_buf = torch.empty(B, max(T, 1), Cout, 2, dtype=torch.float64, device=d) if stats else None
    nb = int(lib.sr3_conv_scratch_bytes(B, Ho, Ho, Cin, Cout, 3, 11, 1))
    extra = 8 << 20
    scratch = torch.zeros(nb + extra, dtype=torch.uint8, device=d)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call():
        L.check(lib.sr3_conv_f32(L.ptr(s0), C0, L.ptr(s1), C1, B, H, H, ups, 1, 3, Cout, L.ptr(w), L.ptr(bias), L.ptr(ss), act,
                                 None, 0, L.ptr(r0), Cout if res else 0, None, 0, L.ptr(out), L.ptr(st_buf), 11, 1,
                                 L.ptr(scratch), nb + extra, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ntiles = B * (Ho // 16) * (Ho // 16) * ((Cout + 63) // 64)
    ts = scratch[nb:nb + ntiles * 8 * 16 * 8].view(torch.int64).cpu().numpy().reshape(ntiles, 8, 16).astype(np.float64)
    dur = np.diff(ts, axis=2)                                   # 15 phases per (tile, wave)
    G = min(ntiles, 256)
    gaps = []
    for b in range(G):                                          # tile v + G follows tile v in the same workgroup
        seq = ts[b::G, 0]
        gaps += list(seq[1:, 0] - seq[:-1, 15])
    print('%s: %.1f us (event, incl. the filter transform); %d tiles, %.1f per workgroup' % (name, ms * 1e3, ntiles, ntiles / G))
    print('   %-52s %s' % ('median ticks per wave ->', ' '.join('%7d' % w for w in range(8))))
    for i in range(15):
        print('   %-52s %s' % (PH[i], ' '.join('%7.0f' % np.median(dur[:, w, i]) for w in range(8))))
    print('   %-52s %s' % ('whole tile', ' '.join('%7.0f' % np.median(ts[:, w, 15] - ts[:, w, 0]) for w in range(8))))
    if gaps:
        print('   gap to the next tile of the workgroup (wave 0): median %.0f' % np.median(gaps))
    sys.stdout.flush()


if __name__ == '__main__':
    if os.environ.get('SR3_WINO_DBG') != '64':
        sys.exit('run with SR3_WINO_DBG=64 and an ablation build (SR3_LIBRARY)')
    run('128x128 64->64 (block2: GN+SiLU, bias, residual, stats)', 64, 0, 128, 64, 0, 2)
    run('128x128 64->64 (no residual, no stats)', 64, 0, 128, 64, 0, 2, res=False, stats=False)
    run('32x32 256->256', 256, 0, 32, 256, 0, 2)
