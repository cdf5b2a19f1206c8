"""Network factory of the drop-in `model` package (reference: model/networks.py:83-116).

`define_G(opt)` reads the same `opt['model']` subtree and returns a GaussianDiffusion whose
denoiser runs on libsr3_mi355x.  Differences by design: no nn.DataParallel wrap -- multi-GPU
is one process per GPU (see sr3_hip/dist.py); `distributed` in opt is accepted and ignored.
"""
import logging

logger = logging.getLogger('base')


def init_weights(net, init_type='kaiming', scale=1, std=0.02):
    """model/networks.py:58-77: `scale` applies to 'kaiming', `std` to 'normal'; define_G uses 'orthogonal'.  The draws
    happen in the reference's `net.apply` order (sr3_hip.nn.EngineUNet.init_scheme)."""
    logger.info('Initialization method [{:s}]'.format(init_type))
    net.denoise_fn.init_scheme(init_type, scale=scale, std=std)


def define_G(opt):
    model_opt = opt['model']
    which = model_opt['which_model_G']
    if which == 'ddpm':
        from .ddpm_modules import diffusion, unet
    elif which == 'sr3':
        from .sr3_modules import diffusion, unet
    else:
        raise NotImplementedError('which_model_G [{}]'.format(which))
    u = model_opt['unet']
    if ('norm_groups' not in u) or u['norm_groups'] is None:
        u['norm_groups'] = 32
    denoiser = unet.UNet(
        in_channel=u['in_channel'], out_channel=u['out_channel'], norm_groups=u['norm_groups'],
        inner_channel=u['inner_channel'], channel_mults=u['channel_multiplier'], attn_res=u['attn_res'],
        res_blocks=u['res_blocks'], dropout=u['dropout'], image_size=model_opt['diffusion']['image_size'])
    netG = diffusion.GaussianDiffusion(
        denoiser, image_size=model_opt['diffusion']['image_size'], channels=model_opt['diffusion']['channels'],
        loss_type='l1', conditional=model_opt['diffusion']['conditional'],
        schedule_opt=model_opt['beta_schedule']['train'])
    if opt['phase'] == 'train':
        init_weights(netG, init_type='orthogonal')
    return netG
