"""DDPM model wrapper of the drop-in `model` package.

Same public methods and attributes as the reference wrapper (model/model.py:12-166): netG, device,
begin_step, begin_epoch, optG, log_dict, SR, data, schedule_phase; feed_data, optimize_parameters,
test, sample, set_loss, set_new_noise_schedule, get_current_log, get_current_visuals,
print_network, save_network, load_network -- so sr.py / infer.py / sample.py run unchanged.
"""
import logging
import os
from collections import OrderedDict

import torch

from sr3_hip import dist as _dist

from . import networks
from .base_model import BaseModel

logger = logging.getLogger('base')


class DDPM(BaseModel):
    def __init__(self, opt):
        super(DDPM, self).__init__(opt)
        self.netG = self.set_device(networks.define_G(opt))
        self.schedule_phase = None
        self.SR = None
        self.data = None
        self.set_loss()
        self.set_new_noise_schedule(opt['model']['beta_schedule']['train'], schedule_phase='train')
        if self.opt['phase'] == 'train':
            self.netG.train()
            if opt['model']['finetune_norm']:
                # model/model.py:26-35: every parameter is frozen and only names containing 'transformer' are
                # optimised -- no such name exists in either UNet, so the reference hands
                # torch.optim.Adam an empty list, which raises.  Same outcome here, same message.
                names = [k for k, _ in self.netG.named_parameters() if k.find('transformer') >= 0]
                if not names:
                    raise ValueError('optimizer got an empty parameter list')
                raise NotImplementedError('finetune_norm over %d "transformer" parameters' % len(names))
            from sr3_hip.optim import make_optimizer
            self.optG = make_optimizer(self.netG, lr=opt['train']['optimizer']['lr'])
            self.log_dict = OrderedDict()
        self.load_network()
        # data parallel: equalise the replicas once (each process initialised its own weights; a resumed checkpoint is
        # already identical) -- the counterpart of DataParallel's per-step replicate (model/networks.py:113-115)
        self._sam_wave = None
        if _dist.dp_active():
            _dist.sync_replicas(self.netG.denoise_fn)
        self.print_network()

    # ---- data / step ---------------------------------------------------------------------------
    def feed_data(self, data):
        self.data = self.set_device(data)

    def optimize_parameters(self):
        self.optG.zero_grad()
        # forward + backward run inside the engine; the gradients already carry the 1/(b*c*h*w) factor
        l_pix = self.netG(self.data)
        b, c, h, w = self.data['HR'].shape
        # data parallel: the engine returns the loss summed over ALL ranks, so normalise by the global element count
        # (the reference's `l_pix.sum() / int(b*c*h*w)` over DataParallel's gathered per-replica sums, :52-53)
        l_pix = l_pix.sum() / int(b * c * h * w * _dist.dp_world_size())
        self.optG.step()
        self.log_dict['l_pix'] = l_pix.item()

    def test(self, continous=False):
        self.netG.eval()
        with torch.no_grad():
            wave = self.data.get('_dp_wave') if isinstance(self.data, dict) else None
            if wave is not None:
                # the validation loader groups consecutive items into waves (sr3_hip.dist.ValWave): the chains of a wave run once,
                # a rank's items as one batch (and, data parallel, dealt over the ranks); every caller gets its item's image back
                self.SR = wave.result(self.netG, self.data['_dp_pos'], continous)
            else:
                self.SR = self.netG.super_resolution(self.data['SR'], continous)
        self.netG.train()

    def sample(self, batch_size=1, continous=False):
        self.netG.eval()
        with torch.no_grad():
            if _dist.dp_active():
                if self._sam_wave is None:
                    self._sam_wave = _dist.SampleWave()
                self.SR = self._sam_wave.next(self.netG, batch_size, continous)
            else:
                self.SR = self.netG.sample(batch_size, continous)
        self.netG.train()

    def set_loss(self):
        self.netG.set_loss(self.device)

    def set_new_noise_schedule(self, schedule_opt, schedule_phase='train'):
        if self.schedule_phase is None or self.schedule_phase != schedule_phase:
            self.schedule_phase = schedule_phase
            self.netG.set_new_noise_schedule(schedule_opt, self.device)

    # ---- reporting -----------------------------------------------------------------------------
    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self, need_LR=True, sample=False):
        out = OrderedDict()
        if sample:
            out['SAM'] = self.SR.detach().float().cpu()
            return out
        out['SR'] = self.SR.detach().float().cpu()
        out['INF'] = self.data['SR'].detach().float().cpu()
        out['HR'] = self.data['HR'].detach().float().cpu()
        if need_LR and 'LR' in self.data:
            out['LR'] = self.data['LR'].detach().float().cpu()
        else:
            out['LR'] = out['INF']
        return out

    def print_network(self):
        if not _dist.is_primary():
            return
        s, n = self.get_network_description(self.netG)
        logger.info('Network G structure: {}, with parameters: {:,d}'.format(self.netG.__class__.__name__, n))
        logger.info(s)

    # ---- checkpoints: the reference's file names and key schema (model/model.py:124-166) ----------
    def _ckpt_paths(self, stem):
        return '{}_gen.pth'.format(stem), '{}_opt.pth'.format(stem)

    def save_network(self, epoch, iter_step):
        stem = os.path.join(self.opt['path']['checkpoint'], 'I{}_E{}'.format(iter_step, epoch))
        gen_path, opt_path = self._ckpt_paths(stem)
        if _dist.is_primary():          # replicas are identical: one writer
            state = OrderedDict((k, v.cpu()) for k, v in self.netG.state_dict().items())
            torch.save(state, gen_path)
            torch.save({'epoch': epoch, 'iter': iter_step, 'scheduler': None,
                        'optimizer': self.optG.state_dict()}, opt_path)
            logger.info('Saved model in [{:s}] ...'.format(gen_path))
        _dist.barrier()                 # nobody runs ahead (or resumes) before the files are complete

    def load_network(self):
        stem = self.opt['path']['resume_state']
        if stem is None:
            return
        logger.info('Loading pretrained model for G [{:s}] ...'.format(stem))
        gen_path, opt_path = self._ckpt_paths(stem)
        self.netG.load_state_dict(torch.load(gen_path, map_location='cpu'),
                                  strict=(not self.opt['model']['finetune_norm']))
        if self.opt['phase'] == 'train':
            ck = torch.load(opt_path, map_location='cpu')
            self.optG.load_state_dict(ck['optimizer'])
            self.begin_step = ck['iter']
            self.begin_epoch = ck['epoch']
            _dist.resume_epoch = int(ck['epoch'])
