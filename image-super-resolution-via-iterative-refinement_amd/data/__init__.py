'''create dataset and dataloader -- drop-in for the reference's data/__init__.py (create_dataloader, create_dataset:
same arguments), with the per-sample float transform moved onto the MI355X: the torch DataLoader collates uint8
batches, DeviceBatches turns each into the reference's batch dict ({'HR', 'SR', ['LR'], 'Index'}: fp32 NCHW in
[-1, 1]) already resident on the device, which DDPM.feed_data then takes as is.'''
import logging

import torch
import torch.utils.data


class DeviceBatches(object):
    """Iterable over a uint8 DataLoader that yields the reference's batch dicts with device-resident fp32 tensors.
    `len()`, iteration order and the dict keys are the DataLoader's; `.dataset` / `.batch_size` pass through."""

    def __init__(self, loader, device=None, min_max=(-1, 1)):
        self.loader = loader
        self.device = device
        self.min_max = min_max
        self.dataset = loader.dataset
        self.batch_size = loader.batch_size
        self.epoch = 0               # reshuffles a rank-sharded sampler every pass (DistributedSampler.set_epoch)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        import data.util as Util
        sampler = getattr(self.loader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(self.epoch)
        self.epoch += 1
        for batch in self.loader:
            flip = batch.pop('flip', None)
            out = {}
            for k, v in batch.items():
                if torch.is_tensor(v) and v.dtype == torch.uint8 and v.dim() == 4:
                    out[k] = Util.u8_batch_to_f32(v, flip, self.min_max, self.device)
                else:
                    out[k] = v
            yield out


def _dp():
    """(rank, world) of the data-parallel job this process belongs to; (0, 1) outside torch.distributed."""
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_rank(), tdist.get_world_size()
    return 0, 1


def create_dataloader(dataset, dataset_opt, phase, device=None):
    '''create dataloader.  Data parallel (one process per GPU): `batch_size` stays the GLOBAL batch of the config, as
    under the reference's nn.DataParallel, which scatters each loader batch over the GPUs (model/networks.py:113-115);
    every rank draws batch_size / world samples per step from its own disjoint shard of the (shuffled) index list.'''
    if phase == 'train':
        rank, world = _dp()
        bs, sampler, shuffle = dataset_opt['batch_size'], None, dataset_opt['use_shuffle']
        if world > 1:
            if bs % world:
                raise ValueError('batch_size %d is not divisible by the %d data-parallel ranks' % (bs, world))
            bs //= world
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank,
                                                                      shuffle=bool(shuffle), drop_last=False)
            shuffle = False
        loader = torch.utils.data.DataLoader(dataset, batch_size=bs, shuffle=shuffle, sampler=sampler,
                                             num_workers=dataset_opt['num_workers'], pin_memory=True)
    elif phase == 'val':
        loader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=1, pin_memory=True)
    else:
        raise NotImplementedError('Dataloader [{:s}] is not found.'.format(phase))
    return DeviceBatches(loader, device)


def create_dataset(dataset_opt, phase):
    '''create dataset'''
    mode = dataset_opt['mode']
    from data.LRHR_dataset import LRHRDataset as D
    dataset = D(dataroot=dataset_opt['dataroot'], datatype=dataset_opt['datatype'], l_resolution=dataset_opt['l_resolution'],
                r_resolution=dataset_opt['r_resolution'], split=phase, data_len=dataset_opt['data_len'],
                need_LR=(mode == 'LRHR'))
    logging.getLogger('base').info('Dataset [{:s} - {:s}] is created.'.format(dataset.__class__.__name__, dataset_opt['name']))
    return dataset
