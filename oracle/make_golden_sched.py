#!/usr/bin/env python3
"""tests/golden/schedules.npz: the reference's make_beta_schedule (model/sr3_modules/diffusion.py:12-49 and
model/ddpm_modules/diffusion.py:12-49) for every schedule name it knows, plus the device buffers
set_new_noise_schedule derives from a non-linear one.  Runs the reference itself (this container only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
from model.sr3_modules import diffusion as RS      # noqa: E402
from model.ddpm_modules import diffusion as RD     # noqa: E402

NAMES = ['quad', 'linear', 'warmup10', 'warmup50', 'const', 'jsd', 'cosine']


def main():
    out = {}
    for n in (20, 2000):
        for name in NAMES:
            a = np.asarray(RS.make_beta_schedule(name, n, linear_start=1e-6, linear_end=1e-2), dtype=np.float64)
            b = np.asarray(RD.make_beta_schedule(name, n, linear_start=1e-4, linear_end=2e-2), dtype=np.float64)
            out['sr3/%s/%d' % (name, n)] = a
            out['ddpm/%s/%d' % (name, n)] = b
    # the buffers of a cosine schedule through the reference's GaussianDiffusion (SR3 flavour)
    g = RS.GaussianDiffusion(torch.nn.Identity(), image_size=8, channels=3, conditional=True)
    g.set_new_noise_schedule(dict(schedule='cosine', n_timestep=50, linear_start=1e-6, linear_end=1e-2), torch.device('cpu'))
    for k, v in g.state_dict().items():
        out['buf/cosine50/' + k] = v.numpy()
    out['buf/cosine50/sqrt_alphas_cumprod_prev'] = np.asarray(g.sqrt_alphas_cumprod_prev)
    dst = os.path.join(ROOT, 'tests', 'golden', 'schedules.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
