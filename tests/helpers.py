"""Shared helpers for the test-suite (golden loading, tiny configs)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

DESCS = {
    'sr3_tiny': dict(variant='sr3', in_channel=6, out_channel=3, inner_channel=8, norm_groups=4,
                     channel_mults=[1, 2, 2], attn_res=[8], res_blocks=1, image_size=16),
    'ddpm_tiny': dict(variant='ddpm', in_channel=3, out_channel=3, inner_channel=8, norm_groups=4,
                      channel_mults=[1, 2], attn_res=[8], res_blocks=2, image_size=16),
    'sr3_seam': dict(variant='sr3', in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                     channel_mults=[1, 2], attn_res=[8], res_blocks=1, image_size=16),
    # unconditional SR3 (the shape of config/sample_sr3_128.json: which_model_G sr3, in_channel 3, conditional false)
    'sr3_uncond': dict(variant='sr3', in_channel=3, out_channel=3, inner_channel=8, norm_groups=4,
                       channel_mults=[1, 2, 2], attn_res=[8], res_blocks=1, image_size=16),
}
SCHEDS = {
    'sr3_tiny': dict(schedule='linear', n_timestep=8, linear_start=1e-6, linear_end=1e-2),
    'ddpm_tiny': dict(schedule='linear', n_timestep=6, linear_start=1e-4, linear_end=2e-2),
    'sr3_seam': dict(schedule='linear', n_timestep=4, linear_start=1e-6, linear_end=1e-2),
    'sr3_uncond': dict(schedule='linear', n_timestep=8, linear_start=1e-6, linear_end=1e-2),
}
CONDITIONAL = {'sr3_tiny': True, 'ddpm_tiny': False, 'sr3_seam': True, 'sr3_uncond': False}


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    g = {k: z[k] for k in z.files}
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd/')}
    return g, sd


def opt_for(name, phase='val', gpu=True):
    d = DESCS[name]
    s = SCHEDS[name]
    return {
        'phase': phase, 'gpu_ids': [0] if gpu else None, 'distributed': False,
        'path': {'checkpoint': '/tmp', 'resume_state': None},
        'train': {'optimizer': {'type': 'adam', 'lr': 1e-4}},
        'model': {
            'which_model_G': d['variant'], 'finetune_norm': False,
            'unet': dict(in_channel=d['in_channel'], out_channel=d['out_channel'],
                         inner_channel=d['inner_channel'], norm_groups=d['norm_groups'],
                         channel_multiplier=d['channel_mults'], attn_res=d['attn_res'],
                         res_blocks=d['res_blocks'], dropout=0),
            'beta_schedule': {'train': dict(s), 'val': dict(s)},
            'diffusion': dict(image_size=d['image_size'], channels=3, conditional=CONDITIONAL[name]),
        },
    }


_EXPERIMENTS = None


def experiments_built():
    """True when libsr3_mi355x was built with -DSR3_EXPERIMENTS (round-1 split_bf16 halo tiles 7 / 8 / 10): the default build
    leaves them out and refuses the option that selects them."""
    global _EXPERIMENTS
    if _EXPERIMENTS is None:
        from sr3_hip import engine as E, lib as L
        p = E.Plan('sr3', 6, 3, 8, 4, [1, 2], [8], 1, 16)
        try:
            p.set_option('split_bf16', 1)
            _EXPERIMENTS = True
        except L.Sr3Error as e:
            assert 'SR3_EXPERIMENTS' in str(e), str(e)
            _EXPERIMENTS = False
    return _EXPERIMENTS


EXPERIMENT_TILES = (7, 8, 10)
