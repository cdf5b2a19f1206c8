"""DDPM GaussianDiffusion on the MI355X engine (reference: model/ddpm_modules/diffusion.py:78-297)."""
from sr3_hip.diffusion import EngineDiffusion, make_beta_schedule  # noqa: F401


class GaussianDiffusion(EngineDiffusion):
    variant = 'ddpm'
