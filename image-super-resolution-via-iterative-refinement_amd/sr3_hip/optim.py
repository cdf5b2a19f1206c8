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
        un.weights_changed()            # the Winograd filters of the inference plan are stale now

    # ---- checkpoint format: torch.optim.Adam's (model/model.py:137-142, 160-163) --------------------------
    # One state entry per parameter in `netG.parameters()` order (= the plan table order, which is the
    # reference's registration order), moments in the reference shapes (OIHW), so `*_opt.pth` files are
    # interchangeable with the reference in both directions.
    def state_dict(self):
        un = self.netG.denoise_fn
        plan = un.plan
        state = {}
        if self.exp_avg is not None and self.step_count > 0:
            for i, e in enumerate(plan.table):
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': plan.view(self.exp_avg, e).detach().cpu().clone().contiguous(),
                            'exp_avg_sq': plan.view(self.exp_avg_sq, e).detach().cpu().clone().contiguous()}
        d = self.defaults
        group = {'lr': d['lr'], 'betas': tuple(d['betas']), 'eps': d['eps'], 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(plan.table)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        un = self.netG.denoise_fn
        plan = un.plan
        groups = sd.get('param_groups') or [{}]
        for k in ('lr', 'eps'):
            if k in groups[0]:
                self.defaults[k] = groups[0][k]
        if 'betas' in groups[0]:
            self.defaults['betas'] = tuple(groups[0]['betas'])
        state = sd.get('state', {})
        if not state:
            self.step_count = 0
            return
        if len(state) != len(plan.table):
            raise ValueError('optimizer state has %d entries, the model has %d parameters' % (len(state), len(plan.table)))
        arena = un.arena.data
        self.exp_avg = torch.zeros_like(arena)
        self.exp_avg_sq = torch.zeros_like(arena)
        steps = set()
        for i, e in enumerate(plan.table):
            st = state[i] if i in state else state[str(i)]
            if tuple(st['exp_avg'].shape) != tuple(e['shape']):
                raise ValueError('optimizer state %d (%s): shape %s vs %s' % (i, e['name'], tuple(st['exp_avg'].shape), e['shape']))
            plan.view(self.exp_avg, e).copy_(st['exp_avg'].to(arena.device, torch.float32))
            plan.view(self.exp_avg_sq, e).copy_(st['exp_avg_sq'].to(arena.device, torch.float32))
            steps.add(int(float(st['step'])))
        if len(steps) != 1:
            raise ValueError('per-parameter Adam step counts differ: %s' % sorted(steps))
        self.step_count = steps.pop()


def make_optimizer(netG, lr):
    return EngineAdam(netG, lr=lr)
