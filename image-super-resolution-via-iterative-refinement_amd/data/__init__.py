'''create dataset and dataloader -- drop-in for the reference's data/__init__.py (create_dataloader, create_dataset:
same arguments), with the per-sample float transform moved onto the MI355X: the torch DataLoader collates uint8
batches, DeviceBatches turns each into the reference's batch dict ({'HR', 'SR', ['LR'], 'Index'}: fp32 NCHW in
[-1, 1]) already resident on the device, which DDPM.feed_data then takes as is.'''
import logging
import os

import torch
import torch.utils.data


class DeviceBatches(object):
    """Iterable over a uint8 DataLoader that yields the reference's batch dicts with device-resident fp32 tensors.
    `len()`, iteration order and the dict keys are the DataLoader's; `.dataset` / `.batch_size` pass through."""

    def __init__(self, loader, device=None, min_max=(-1, 1), deal_waves=False):
        self.loader = loader
        self.device = device
        self.min_max = min_max
        self.dataset = loader.dataset
        self.batch_size = loader.batch_size
        self.epoch = 0               # reshuffles a rank-sharded sampler every pass (DistributedSampler.set_epoch)
        self.deal_waves = deal_waves

    def __len__(self):
        return len(self.loader)

    def _device_batches(self):
        import data.util as Util
        sampler = getattr(self.loader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            if self.epoch == 0:
                from sr3_hip import dist as _dist
                self.epoch = _dist.resume_epoch      # a resumed run continues the reshuffle sequence
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

    def __iter__(self):
        _, world = _dp()
        from sr3_hip.dist import dp_active, val_chain_batch
        chain = val_chain_batch() if self.deal_waves else 1
        # (SR3_VAL_ITEM_STREAMS=1 set explicitly: one-item waves too, so a one-chain-per-image run draws the per-item streams a
        # batched run draws -- the comparison the streams exist for; unset, a plain batch-1 run keeps the default generator)
        force = self.deal_waves and os.environ.get('SR3_VAL_ITEM_STREAMS') == '1'
        if not (self.deal_waves and (dp_active() or chain > 1 or force)):
            yield from self._device_batches()
            return
        # validation: every rank walks ALL items in order, `world * chain` at a time; the wave object lets DDPM.test run the
        # reverse chains once per wave -- a rank's items as one chain batch -- and share the finished images (sr3_hip.dist.ValWave)
        from sr3_hip.dist import ValWave
        world = world if dp_active() else 1
        group = []
        first = 0                     # index of group[0] in this pass over the validation set: item k draws stream k in every pass

        def flush():
            wave = ValWave([b['SR'] for b in group], first_item=first)
            for pos, b in enumerate(group):
                b['_dp_wave'], b['_dp_pos'] = wave, pos
                yield b
        for b in self._device_batches():
            group.append(b)
            if len(group) == world * chain:
                yield from flush()
                first += len(group)
                group = []
        if group:
            yield from flush()


def _dp():
    """(rank, world) of the data-parallel job this process belongs to; (0, 1) for a single process.  sr.py builds its
    loaders BEFORE the model (sr.py:52-66), so this is where a torchrun-started script usually joins the job."""
    from sr3_hip.dist import bootstrap
    rank, world, _ = bootstrap()
    return rank, world


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
            # the permutation seed comes from the global torch RNG like RandomSampler's (so `torch.manual_seed` governs
            # the order as it does in the reference) -- rank 0's draw, shared, so the shards stay disjoint
            from sr3_hip.dist import broadcast_int
            seed = broadcast_int(int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFF)
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank,
                                                                      shuffle=bool(shuffle), seed=seed, drop_last=False)
            shuffle = False
        loader = torch.utils.data.DataLoader(dataset, batch_size=bs, shuffle=shuffle, sampler=sampler,
                                             num_workers=dataset_opt['num_workers'], pin_memory=True)
    elif phase == 'val':
        _dp()
        loader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=1, pin_memory=True)
        return DeviceBatches(loader, device, deal_waves=True)
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
