"""Noise schedules and the derived diffusion buffers (float64 numpy, cast to float32 last).

Restates utils/tools.py:425-445 (`vpsde_beta_t`, `get_noise_schedule_list`) and the buffer
algebra of model/diffusion.py:45-83.  Test infrastructure (see oracle/__init__.py).
"""
import numpy as np

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def beta_schedule(mode, timesteps, min_beta=0.0, max_beta=0.01, s=0.008):
    """utils/tools.py:430-445."""
    T = int(timesteps)
    if mode == "linear":
        return np.linspace(1e-4, max_beta, T)
    if mode == "cosine":
        n = T + 1
        grid = np.linspace(0, n, n)
        abar = np.cos((grid / n + s) / (1 + s) * np.pi * 0.5) ** 2
        abar = abar / abar[0]
        return np.clip(1 - abar[1:] / abar[:-1], a_min=0, a_max=0.999)
    if mode == "vpsde":
        # utils/tools.py:425-427: beta_t = 1 - exp(-bmin/T - 0.5 (bmax-bmin)(2t-1)/T^2), t = 1..T
        out = []
        for t in range(1, T + 1):
            coef = (2 * t - 1) / (T ** 2)
            out.append(1.0 - np.exp(-min_beta / T - 0.5 * (max_beta - min_beta) * coef))
        return np.array(out)
    raise NotImplementedError(mode)


def diffusion_buffers(betas):
    """model/diffusion.py:53-83 -- all float64 until the final cast."""
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    abar = np.cumprod(alphas, axis=0)
    abar_prev = np.append(1.0, abar[:-1])
    post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
    f64 = {
        "betas": betas,
        "alphas_cumprod": abar,
        "alphas_cumprod_prev": abar_prev,
        "sqrt_alphas_cumprod": np.sqrt(abar),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - abar),
        "log_one_minus_alphas_cumprod": np.log(1.0 - abar),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / abar),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / abar - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.maximum(post_var, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(abar_prev) / (1.0 - abar),
        "posterior_mean_coef2": (1.0 - abar_prev) * np.sqrt(alphas) / (1.0 - abar),
    }
    return {k: v.astype(np.float32) for k, v in f64.items()}
