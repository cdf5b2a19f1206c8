"""SR3 GaussianDiffusion on the MI355X engine (reference: model/sr3_modules/diffusion.py:64-249).

The class body carries the SR3-specific public signatures of the reference (`p_sample` without `repeat_noise`,
`q_sample` driven by the continuous noise level); everything else -- schedule buffers, the hipGraph reverse loop,
`p_losses` -- is shared engine code in sr3_hip/diffusion.py.  Parameters after `*` are engine extensions (noise
injection for the parity tests); the reference's callers never pass them."""
import torch

from sr3_hip.diffusion import EngineDiffusion, make_beta_schedule  # noqa: F401


class GaussianDiffusion(EngineDiffusion):
    variant = 'sr3'

    @torch.no_grad()
    def p_sample(self, x, t, clip_denoised=True, condition_x=None, *, noise=None):
        """reference :169-174 (integer t; draws nothing at t == 0)."""
        return self._p_sample(x, t, clip_denoised=clip_denoised, condition_x=condition_x, noise=noise)

    def q_sample(self, x_start, continuous_sqrt_alpha_cumprod, noise=None):
        """reference :212-219: sqrt(abar) * x0 + sqrt(1 - abar) * noise with a per-sample continuous sqrt(abar)."""
        return self._q_sample(x_start, continuous_sqrt_alpha_cumprod, noise)
