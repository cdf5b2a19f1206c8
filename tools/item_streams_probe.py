#!/usr/bin/env python3
"""What the per-item noise streams of a batched validation chain cost (sr3_hip.dist.val_item_streams): the SAME 2000-step chain of the
BASELINE config at batch 16 through `super_resolution`, once with one draw per step for the batch (default generator) and once with one
generator per image (item_seeds: 16 `normal_` nodes per step in the captured graph instead of one), timed with HIP events after the
graphs exist.  Also checks the property the streams are for: image k of the batched chain against its own batch-1 chain.
    python tools/item_streams_probe.py [--config sr3_16_128] [--steps 2000] [--reps 2]      (GPU box)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='sr3_16_128')
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--single', type=int, default=1, help='how many images to re-run as batch-1 chains')
    a = ap.parse_args()
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import model.networks as networks
    from sr3_hip import dist as D
    cfg = bench.CONFIGS[a.config]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    opt = bench.config_opt(a.config, n_timestep=a.steps)
    netG = networks.define_G(opt).to(dev)
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['val'], dev)
    netG.show_progress = False
    B, S = cfg['batch'], cfg['size']
    cond = torch.rand(B, 3, S, S, device=dev) * 2 - 1
    seeds = [D.val_item_seed(k, 0, base=1234) for k in range(B)]

    def timed(**kw):
        netG.super_resolution(cond, False, **kw)          # capture + one chain (warm)
        best = None
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            netG.super_resolution(cond, False, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            best = ms if best is None else min(best, ms)
        return best
    rec = {'what': 'per-item noise streams', 'config': a.config, 'batch': B, 'steps': a.steps}
    rec['ms_per_step_one_draw_per_batch'] = timed()
    rec['ms_per_step_item_streams'] = timed(item_seeds=seeds)
    rec['extra_us_per_step'] = 1e3 * (rec['ms_per_step_item_streams'] - rec['ms_per_step_one_draw_per_batch'])
    batched = netG.super_resolution(cond, True, item_seeds=seeds)
    n_snap = batched.shape[0] // B
    batched = batched.view(n_snap, B, 3, S, S)
    worst = 0.0
    for k in range(min(a.single, B)):
        one = netG.super_resolution(cond[k:k + 1], True, item_seeds=[seeds[k]]).view(n_snap, 3, S, S)
        worst = max(worst, float((one - batched[:, k]).abs().max()))
    rec['images_rerun_as_batch_1'] = min(a.single, B)
    rec['max_abs_batched_vs_batch_1'] = worst
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
