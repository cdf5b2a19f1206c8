"""Optimizer face for the engine (reference: torch.optim.Adam in model/model.py:39-40,54-55).

The gradients are produced by the engine's fused forward+backward (`EngineUNet.train_step`, called from
`GaussianDiffusion.p_losses`); this object owns Adam's moments as two arena-shaped buffers and applies
the update with one fused kernel over the whole parameter arena (sr3_adam_step).
"""
import ctypes as C

import torch

from . import lib as L


class EngineAdam(object):
    def __init__(self, netG, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.netG = netG
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    def zero_grad(self):
        pass                        # every gradient is overwritten by the next train_step

    def _moments(self, arena):
        if self.exp_avg is None or self.exp_avg.device != arena.device:
            self.exp_avg = torch.zeros_like(arena) if self.exp_avg is None else self.exp_avg.to(arena.device)
            self.exp_avg_sq = torch.zeros_like(arena) if self.exp_avg_sq is None else self.exp_avg_sq.to(arena.device)

    def step(self):
        un = self.netG.denoise_fn
        arena = un.arena.data
        if getattr(un, 'grad_arena', None) is None:
            raise L.Sr3Error('optimizer step without gradients: call netG(data) first')
        self._moments(arena)
        self.step_count += 1
        d = self.defaults
        stream = C.c_void_p(torch.cuda.current_stream(arena.device).cuda_stream)
        L.check(L.load().sr3_adam_step(L.ptr(arena), L.ptr(un.grad_arena), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                       arena.numel(), C.c_float(d['lr']), C.c_float(d['betas'][0]),
                                       C.c_float(d['betas'][1]), C.c_float(d['eps']), self.step_count, stream))

    def state_dict(self):
        return {'state': {0: {'step': self.step_count,
                              'exp_avg': None if self.exp_avg is None else self.exp_avg.cpu(),
                              'exp_avg_sq': None if self.exp_avg_sq is None else self.exp_avg_sq.cpu()}},
                'param_groups': [dict(self.defaults, params=[0])], 'engine_arena': True}

    def load_state_dict(self, sd):
        st = sd.get('state', {}).get(0, {})
        self.step_count = int(st.get('step', sd.get('engine_step', 0)))
        if st.get('exp_avg') is not None:
            self.exp_avg = st['exp_avg'].clone()
            self.exp_avg_sq = st['exp_avg_sq'].clone()


def make_optimizer(netG, lr):
    return EngineAdam(netG, lr=lr)
