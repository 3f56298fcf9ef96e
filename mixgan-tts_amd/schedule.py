"""Noise schedule and the GaussianDiffusion buffers (host side, float64 numpy).

Mirrors utils/tools.py:425-445 and model/diffusion.py:45-83 of the reference; init-time only.
"""
import numpy as np

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


def beta_schedule(schedule_mode, timesteps, min_beta=0.0, max_beta=0.01, s=0.008):
    T = int(timesteps)
    if schedule_mode == "linear":
        return np.linspace(1e-4, max_beta, T)
    if schedule_mode == "cosine":
        steps = T + 1
        x = np.linspace(0, steps, steps)
        ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        return np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    if schedule_mode == "vpsde":
        return np.array([1.0 - np.exp(-min_beta / T - 0.5 * (max_beta - min_beta) * ((2 * ti - 1) / (T ** 2)))
                         for ti in range(1, T + 1)])
    raise NotImplementedError(schedule_mode)


def diffusion_buffers(betas):
    betas = np.asarray(betas, dtype=np.float64)
    with np.errstate(all="ignore"):
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        vals = (
            betas, ac, ac_prev, np.sqrt(ac), np.sqrt(1.0 - ac), np.log(1.0 - ac), np.sqrt(1.0 / ac),
            np.sqrt(1.0 / ac - 1), pv, np.log(np.maximum(pv, 1e-20)),
            betas * np.sqrt(ac_prev) / (1.0 - ac), (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
        )
    return {k: v.astype(np.float32) for k, v in zip(BUFFER_NAMES, vals)}
