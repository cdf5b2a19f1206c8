"""SR3 denoiser (continuous noise-level conditioning) on the MI355X engine.
Reference: model/sr3_modules/unet.py:161-259 -- same constructor and forward(x, time)."""
from sr3_hip.nn import EngineUNet


class UNet(EngineUNet):
    variant = 'sr3'
