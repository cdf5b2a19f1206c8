"""SR3 GaussianDiffusion on the MI355X engine (reference: model/sr3_modules/diffusion.py:64-249)."""
from sr3_hip.diffusion import EngineDiffusion, make_beta_schedule  # noqa: F401


class GaussianDiffusion(EngineDiffusion):
    variant = 'sr3'
