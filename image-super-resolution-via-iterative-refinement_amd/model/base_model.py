"""BaseModel of the drop-in `model` package (reference: model/base_model.py:6-48)."""
import torch
import torch.nn as nn


class BaseModel(object):
    def __init__(self, opt):
        self.opt = opt
        # 'cuda' is HIP on ROCm; the engine itself refuses to run on 'cpu'
        self.device = torch.device('cuda' if opt['gpu_ids'] is not None else 'cpu')
        self.begin_step = 0
        self.begin_epoch = 0

    def feed_data(self, data):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        pass

    def get_current_losses(self):
        pass

    def print_network(self):
        pass

    def set_device(self, x):
        """Move a tensor / module / dict of tensors / list of tensors to self.device."""
        if isinstance(x, dict):
            for k in list(x.keys()):
                if x[k] is not None:
                    x[k] = x[k].to(self.device)
            return x
        if isinstance(x, list):
            return [None if v is None else v.to(self.device) for v in x]
        return x.to(self.device)

    def get_network_description(self, network):
        if isinstance(network, nn.DataParallel):
            network = network.module
        return str(network), sum(p.numel() for p in network.parameters())
