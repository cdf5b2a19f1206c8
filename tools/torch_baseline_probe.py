#!/usr/bin/env python3
"""Diagnostic (GPU box): what stock PyTorch-ROCm (MIOpen / rocBLAS, eager) does with the same network -- run under
rocprofv3 --kernel-trace --stats to see which library kernels it picks and how long they take.
  python tools/torch_baseline_probe.py --config sr3_16_128 --batch 16 --steps 5 [--train-batch 64]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')):
    sys.path.insert(0, p)
import torch                                      # noqa: E402
import bench                                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='sr3_16_128')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--train-batch', type=int, default=0)
    a = ap.parse_args()
    from oracle import sr3_oracle as O
    import model.networks as networks
    c = bench.CONFIGS[a.config]
    opt = bench.config_opt(a.config)
    torch.manual_seed(0)
    netG = networks.define_G(opt)
    sd = {k: v.detach().clone() for k, v in netG.state_dict().items()}
    desc = O.desc_from_opt(opt)
    tab = O.schedule_tables(opt['model']['beta_schedule']['val'])
    dev = torch.device('cuda:0')
    sdd = {k: v.to(dev) for k, v in sd.items()}
    B, S = a.batch, c['size']
    x = torch.randn(B, 3, S, S, device=dev)
    z = torch.randn(B, 3, S, S, device=dev)
    cond = (torch.rand(B, 3, S, S, device=dev) * 2 - 1) if c['conditional'] else None
    with torch.no_grad():
        t0 = time.time()
        x1 = O.p_sample(sdd, desc, tab, x, 1000, z, condition_x=cond)
        torch.cuda.synchronize()
        print('warm-up step %.1f s' % (time.time() - t0), flush=True)
        t0 = time.time()
        for i in range(a.steps):
            x1 = O.p_sample(sdd, desc, tab, x1, 999 - i, z, condition_x=cond)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / a.steps * 1e3
    fl = 1.0
    print('torch_baseline sampling: %s batch %d: %.2f ms/step' % (a.config, B, ms), flush=True)
    del sdd, x1
    torch.cuda.empty_cache()
    if a.train_batch > 0:
        r = bench.oracle_train_baseline(a.config, sd, a.train_batch, 'cuda:0', 0, steps=3)
        print('torch_baseline training: batch %d: %.1f ms/step, %.1f images/s' % (a.train_batch, r['s_per_step'] * 1e3, r['value']),
              flush=True)


if __name__ == '__main__':
    main()
