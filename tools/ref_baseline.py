#!/usr/bin/env python3
"""CPU baseline with the REFERENCE ITSELF: imports `model` from --ref (the unmodified reference tree,
never this repo's drop-in -- hence a separate process), loads the weights bench.py hands over, and times
  * sampling: `netG.p_sample(x, t, condition_x=cond)` (model/sr3_modules/diffusion.py:169-174) at the config's batch,
    1 warm-up + up to 8 timed steps within --budget seconds;
  * training: `DDPM.optimize_parameters()` (model/model.py:48-58) at --train-batch, 1 warm-up + 3 timed steps.
Prints one JSON object (the `cpu_baseline` record of bench.py, kind "reference") on the last stdout line.
Only bench.py's cpu_baseline leg (and a CPU test) run this; nothing in the product path does."""
import argparse
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', required=True)
    ap.add_argument('--config', required=True)
    ap.add_argument('--state', required=True, help='torch file {sd, x, cond, t} written by bench.py')
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--budget', type=float, default=25.0)
    ap.add_argument('--train-batch', type=int, default=4)
    ap.add_argument('--max-steps', type=int, default=8)
    ap.add_argument('--train-steps', type=int, default=3)
    a = ap.parse_args()

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, here)
    from bench import CONFIGS, config_opt            # plain dict helpers; imports nothing of the engine
    # the reference tree must win every `import model` / `import core`
    sys.path = [a.ref] + [p for p in sys.path if 'image-super-resolution-via-iterative-refinement_amd' not in p]
    import logging
    logging.disable(logging.CRITICAL)
    import torch
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    import model.networks as networks
    assert os.path.realpath(networks.__file__).startswith(os.path.realpath(a.ref)), networks.__file__

    c = CONFIGS[a.config]
    blob = torch.load(a.state, map_location='cpu')
    opt = config_opt(a.config)
    opt['gpu_ids'] = None
    netG = networks.define_G(opt)
    netG.set_loss('cpu')
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['val'], 'cpu')
    missing = netG.load_state_dict(blob['sd'], strict=False)
    assert not [k for k in missing.missing_keys if k.startswith('denoise_fn.')], missing.missing_keys[:4]
    netG.eval()
    x, cond, t = blob['x'], blob['cond'], int(blob['t'])
    B = x.shape[0]
    def step(x, ti):
        # SR3 takes a Python int (sr3 diffusion.py:169), DDPM a (B,) int64 tensor (ddpm diffusion.py:184)
        tt = ti if c['which'] == 'sr3' else torch.full((B,), ti, dtype=torch.long)
        return netG.p_sample(x, tt, condition_x=cond) if cond is not None else netG.p_sample(x, tt)

    with torch.no_grad():
        t0 = time.time()
        x = step(x, t)
        warm = time.time() - t0
        times = []
        start = time.time()
        n = 0
        while n < a.max_steps and (n == 0 or (time.time() - start) * (n + 1) / n < a.budget):
            t1 = time.time()
            x = step(x, t - 1 - n)
            times.append(time.time() - t1)
            n += 1
    t_step = sum(times) / len(times)
    rec = dict(value=B / (2000.0 * t_step), unit='images/s', cores=int(torch.get_num_threads()), kind='reference',
               sample='%d reverse steps of the reference GaussianDiffusion.p_sample (imported from %s) at batch %d after 1 warm-up '
                      'step at the same batch (%.2f s), %.2f s/step, extrapolated x2000' % (len(times), a.ref, B, warm, t_step))
    # ---- training: the reference's DDPM wrapper, CPU ----
    try:
        del netG
        import model as Model
        topt = config_opt(a.config, phase='train')
        topt['gpu_ids'] = None
        m = Model.create_model(topt)
        S = c['size']
        g = torch.Generator().manual_seed(9)
        tb = a.train_batch
        data = {'HR': torch.rand(tb, 3, S, S, generator=g) * 2 - 1, 'SR': torch.rand(tb, 3, S, S, generator=g) * 2 - 1}
        m.feed_data(data)
        tt = []
        for it in range(a.train_steps + 1):
            t1 = time.time()
            m.optimize_parameters()
            if it > 0:
                tt.append(time.time() - t1)
        ts = sum(tt) / len(tt)
        rec['train'] = dict(value=tb / ts, unit='images/s', s_per_step=ts, batch=tb, cores=int(torch.get_num_threads()),
                            sample='%d DDPM.optimize_parameters() steps of the reference at batch %d after 1 warm-up '
                                   '(dropout %.1f as configured)' % (len(tt), tb, c['unet']['dropout']))
    except Exception as e:                       # the sampling record must survive
        rec['train'] = {'error': '%s: %s' % (type(e).__name__, e)}
    print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
