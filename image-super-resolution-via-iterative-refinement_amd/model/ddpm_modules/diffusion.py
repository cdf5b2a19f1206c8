"""DDPM GaussianDiffusion on the MI355X engine (reference: model/ddpm_modules/diffusion.py:78-297).

The class body carries the DDPM-specific public surface of the reference: per-sample integer timesteps in `p_sample` /
`q_sample`, `q_mean_variance` (:151-156) and `interpolate` (:242-257).  Everything else is shared engine code in
sr3_hip/diffusion.py.  Parameters after `*` are engine extensions; the reference's callers never pass them."""
import torch

from sr3_hip.diffusion import EngineDiffusion, make_beta_schedule  # noqa: F401


class GaussianDiffusion(EngineDiffusion):
    variant = 'ddpm'

    @torch.no_grad()
    def p_sample(self, x, t, clip_denoised=True, repeat_noise=False, condition_x=None, *, noise=None):
        """reference :184-198 (t: (B,) int64; the t == 0 rows get no noise through the sigma table)."""
        return self._p_sample(x, t, clip_denoised=clip_denoised, repeat_noise=repeat_noise, condition_x=condition_x,
                              noise=noise)

    def q_posterior(self, x_start, x_t, t):
        """reference :162-172: the DDPM class returns THREE values (mean, variance, clipped log variance)."""
        mean, logvar = super().q_posterior(x_start, x_t, t)
        return mean, self._coef('posterior_variance', t, x_t), logvar

    def p_mean_variance(self, x, t, clip_denoised: bool, condition_x=None):
        """reference :174-189: (model_mean, posterior_variance, posterior_log_variance), per-sample t."""
        mean, logvar = super().p_mean_variance(x, t, clip_denoised, condition_x=condition_x)
        return mean, self._coef('posterior_variance', t, x), logvar

    def q_sample(self, x_start, t, noise=None):
        """reference :259-267: sqrt(abar_t) * x0 + sqrt(1 - abar_t) * noise."""
        return self._q_sample(x_start, t, noise)

    def q_mean_variance(self, x_start, t):
        """reference :151-156 (table gathers only; no caller on the hot path)."""
        mean = self._coef('sqrt_alphas_cumprod', t, x_start) * x_start
        variance = self._coef('alphas_cumprod', t, x_start).neg().add(1.0)
        return mean, variance, self._coef('log_one_minus_alphas_cumprod', t, x_start)

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, lam=0.5):
        """reference :242-257: diffuse both images to step t, blend, run the reverse chain from t - 1 down to 0.
        Each reverse step is the engine's p_sample (UNet forward + fused update)."""
        if x1.shape != x2.shape:
            raise AssertionError('interpolate: shapes differ')
        b, dev = x1.shape[0], x1.device
        if t is None:
            t = self.num_timesteps - 1
        tb = torch.full((b,), int(t), device=dev, dtype=torch.long)
        xt1, xt2 = self.q_sample(x1, t=tb), self.q_sample(x2, t=tb)
        img = (1 - lam) * xt1 + lam * xt2
        for i in reversed(range(0, int(t))):
            img = self.p_sample(img, torch.full((b,), i, device=dev, dtype=torch.long))
        return img
