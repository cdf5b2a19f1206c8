#!/usr/bin/env python3
"""CPU only, this container only (needs /root/reference): times the REFERENCE's own p_sample / optimize_parameters
(tools/ref_baseline.py, reference tree imported in a subprocess) and the oracle port (oracle/sr3_oracle.py, what bench.py's
`cpu_baseline` leg runs on the GPU box, where the reference tree does not exist) on the same weights, inputs, batch and
thread count -- the evidence that the port's timing stands for the reference's.  Writes one JSON object.
  python tools/ref_vs_port_cpu.py --batch 4 --steps 3 --out profiles/r04_ref_vs_port_cpu.json"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GEN = r'''
import sys, json, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(1, sys.argv[2])
from bench import CONFIGS, config_opt
sys.path = [sys.argv[1]] + [p for p in sys.path if 'image-super-resolution-via-iterative-refinement_amd' not in p]
import logging; logging.disable(logging.CRITICAL)
import model.networks as networks
cfg, B, out = sys.argv[3], int(sys.argv[4]), sys.argv[5]
opt = config_opt(cfg); opt['gpu_ids'] = None
torch.manual_seed(0)
netG = networks.define_G(opt)
S = CONFIGS[cfg]['size']
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 3, S, S, generator=g)
cond = (torch.rand(B, 3, S, S, generator=g) * 2 - 1) if CONFIGS[cfg]['conditional'] else None
torch.save({'sd': {k: v.detach().clone() for k, v in netG.state_dict().items()}, 'x': x, 'cond': cond, 't': 1500}, out)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--config', default='sr3_16_128')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--train-batch', type=int, default=2)
    ap.add_argument('--threads', type=int, default=os.cpu_count())
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    import torch
    torch.set_num_threads(a.threads)
    from bench import CONFIGS, config_opt, oracle_train_baseline
    from oracle import sr3_oracle as O
    state = os.path.join(tempfile.mkdtemp(prefix='sr3_refport_'), 'state.pth')
    # the reference's own constructor draws the weights (its tree in a subprocess: the package names collide)
    subprocess.run([sys.executable, '-c', GEN, a.ref, ROOT, a.config, str(a.batch), state], check=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'ref_baseline.py'), '--ref', a.ref, '--config', a.config,
                        '--state', state, '--threads', str(a.threads), '--budget', '1e9', '--max-steps', str(a.steps),
                        '--train-batch', str(a.train_batch), '--train-steps', '2'], stdout=subprocess.PIPE, check=True)
    ref = json.loads(r.stdout.decode().strip().splitlines()[-1])
    blob = torch.load(state, map_location='cpu')
    sd, x, cond, t = blob['sd'], blob['x'], blob['cond'], int(blob['t'])
    opt = config_opt(a.config)
    desc = O.desc_from_opt(opt)
    tab = O.schedule_tables(opt['model']['beta_schedule']['val'])
    g = torch.Generator().manual_seed(6)
    z = torch.randn(x.shape, generator=g)
    with torch.no_grad():
        t0 = time.time()
        xx = O.p_sample(sd, desc, tab, x, t, z, condition_x=cond)
        warm = time.time() - t0
        times = []
        for n in range(a.steps):
            t1 = time.time()
            xx = O.p_sample(sd, desc, tab, xx, t - 1 - n, z, condition_x=cond)
            times.append(time.time() - t1)
    ts = sum(times) / len(times)
    port = dict(value=a.batch / (2000.0 * ts), unit='images/s', cores=a.threads, kind='port',
                sample='%d reverse steps of oracle p_sample at batch %d after 1 warm-up (%.2f s), %.2f s/step, extrapolated x2000'
                       % (len(times), a.batch, warm, ts))
    port['train'] = oracle_train_baseline(a.config, sd, a.train_batch, 'cpu', a.threads, steps=2)
    rec = dict(config=a.config, batch=a.batch, threads=a.threads, host='build container (no GPU)', reference=ref, port=port,
               port_over_reference_sampling=port['value'] / ref['value'],
               port_over_reference_training=(port['train']['value'] / ref['train']['value']) if 'value' in ref.get('train', {}) else None,
               note='same weights (drawn by the reference constructor, seed 0), same x / cond / t, same torch thread count; the '
                    'reference training step includes its dropout (0.2), the port baseline leaves it out')
    s = json.dumps(rec, indent=1)
    print(s)
    if a.out:
        open(a.out, 'w').write(s + '\n')


if __name__ == '__main__':
    main()
