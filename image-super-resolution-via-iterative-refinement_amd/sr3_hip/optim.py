"""Optimizer face for the engine (reference: torch.optim.Adam in model/model.py:39-40,54-55).

The fused multi-tensor Adam and the UNet backward kernels are not built yet in this round: the
object exists so DDPM can be constructed in the train phase (checkpoint plumbing, schedule
switches, validation during training all work), and fails loudly -- never silently falls back to
autograd -- when a training step is requested.
"""


class EngineAdam(object):
    def __init__(self, netG, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.netG = netG
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        self.step_count = 0

    def zero_grad(self):
        pass

    def backward_and_step(self, netG, scale):
        raise NotImplementedError('training step kernels (UNet backward + fused Adam) are not built yet')

    def state_dict(self):
        return {'state': {}, 'param_groups': [dict(self.defaults, params=[])], 'engine_step': self.step_count}

    def load_state_dict(self, sd):
        self.step_count = int(sd.get('engine_step', 0))


def make_optimizer(netG, lr):
    return EngineAdam(netG, lr=lr)
