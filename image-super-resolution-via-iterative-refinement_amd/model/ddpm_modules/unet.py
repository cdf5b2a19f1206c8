"""DDPM denoiser (integer-timestep conditioning) on the MI355X engine.
Reference: model/ddpm_modules/unet.py:147-243 -- same constructor and forward(x, time)."""
from sr3_hip.nn import EngineUNet


class UNet(EngineUNet):
    variant = 'ddpm'

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8), res_blocks=3, dropout=0,
                 with_time_emb=True, image_size=128):
        super().__init__(in_channel, out_channel, inner_channel, norm_groups, channel_mults, attn_res,
                         res_blocks, dropout, with_time_emb, image_size)
