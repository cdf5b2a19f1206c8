"""BaseModel of the drop-in `model` package: the small host-side base the DDPM wrapper builds on.

Contract kept from the reference (model/base_model.py:6-48): attributes `opt`, `device`, `begin_step`,
`begin_epoch`; `set_device(x)` moves a tensor, a module, a dict of tensors (in place, None entries kept) or a list of
tensors to the model's device; `get_network_description(net)` returns (repr, parameter count).  The reference's empty
hook methods are provided through one table instead of five stubs."""
import torch
from torch import nn


def _to_device(value, device):
    # tensors / modules move; None and plain Python values (the data-parallel loader's bookkeeping entries) pass through
    return value.to(device) if hasattr(value, 'to') else value


class BaseModel(object):
    #: hooks a subclass overrides; the base versions do nothing (as in the reference)
    _HOOKS = ('feed_data', 'optimize_parameters', 'get_current_visuals', 'get_current_losses', 'print_network')

    def __init__(self, opt):
        self.opt = opt
        # 'cuda' is HIP on ROCm; with gpu_ids unset the object can be built (checkpoint I/O) but the engine refuses to run.
        # One process per GPU: inside a data-parallel job the model lives on this rank's own device (cuda:LOCAL_RANK,
        # already made current by sr3_hip.dist.bootstrap), where the reference's DataParallel root is always cuda:0.
        from sr3_hip import dist as _dist
        _, world, local = _dist.bootstrap()
        if opt['gpu_ids'] is None:
            self.device = torch.device('cpu')
        elif _dist.dp_active() and torch.cuda.is_available():
            self.device = torch.device('cuda', local)
        else:
            self.device = torch.device('cuda')
        self.begin_step = self.begin_epoch = 0

    def set_device(self, x):
        if isinstance(x, dict):
            x.update({key: _to_device(val, self.device) for key, val in x.items()})
            return x
        if isinstance(x, list):
            return [_to_device(val, self.device) for val in x]
        return _to_device(x, self.device)

    def get_network_description(self, network):
        net = network.module if isinstance(network, nn.DataParallel) else network
        return str(net), sum(w.numel() for w in net.parameters())


def _make_hook(takes_data):
    if takes_data:
        def hook(self, data):
            return None
    else:
        def hook(self):
            return None
    return hook


for _name in BaseModel._HOOKS:          # same parameter lists as the reference's empty methods (base_model.py:14-27)
    _h = _make_hook(_name == 'feed_data')
    _h.__name__ = _h.__qualname__ = _name
    setattr(BaseModel, _name, _h)
