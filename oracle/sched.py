"""Oracle (test infrastructure): noise-schedule tables and the three steppers on the hot path.

Float32 torch on CPU; integer tables in numpy int64.  Citations are to /root/reference.

* LCM one-step preview: schedulers/lcm_single_step_scheduler.py:194-249 (tables), :401-407
  (boundary scalings), :421-489 (step), :492-513 (add_noise), :331-399 (set_timesteps).
  PINNED by tests/golden/lcm_scheduler.npz generated from that file.
* DDPM ancestral step / DDIM(eta) step: diffusers==0.28.1 DDPMScheduler / DDIMScheduler are not
  in the container (PARITY UNPINNED).  Formulas: SURVEY.md section 8a row S1; in-tree analogue
  train_previewer_lora.py:194-219 (DDIMSolver.ddim_step) and :239-254 (x0 from epsilon).
"""
from __future__ import annotations

import numpy as np
import torch


def make_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    # schedulers/lcm_single_step_scheduler.py:221-234: scaled_linear betas in float32, cumprod in float32
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    alphas = 1.0 - betas
    return torch.cumprod(alphas, dim=0)


def leading_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=1):
    """diffusers 'leading' spacing: t_i = (i * floor(T/N))[::-1] + offset, int64."""
    step_ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
    return ts + steps_offset


def lcm_timesteps(num_inference_steps, num_train_timesteps=1000, original_steps=50, strength=1.0):
    # schedulers/lcm_single_step_scheduler.py:389-397
    c = num_train_timesteps // original_steps
    origin = np.asarray(list(range(1, int(original_steps * strength) + 1))) * c - 1
    skipping = len(origin) // num_inference_steps
    return origin[::-skipping][:num_inference_steps].astype(np.int64)


def lcm_scalings(timestep, sigma_data=0.5, timestep_scaling=10.0):
    # schedulers/lcm_single_step_scheduler.py:401-407
    st = timestep * timestep_scaling
    c_skip = sigma_data ** 2 / (st ** 2 + sigma_data ** 2)
    c_out = st / (st ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out


def _bcast(v, ndim):
    return v.reshape(v.shape[0], *((1,) * (ndim - 1)))


def lcm_step(alphas_cumprod, model_output, timestep, sample):
    """schedulers/lcm_single_step_scheduler.py:421-489, epsilon prediction, no clipping."""
    t = torch.as_tensor(timestep, dtype=torch.int64)
    if t.ndim == 0:
        t = t[None]
    a = _bcast(alphas_cumprod.gather(-1, t), sample.ndim)
    b = 1 - a
    c_skip, c_out = lcm_scalings(t)
    c_skip, c_out = _bcast(c_skip, sample.ndim), _bcast(c_out, sample.ndim)
    x0 = (sample - torch.sqrt(b) * model_output) / torch.sqrt(a)
    return c_out * x0 + c_skip * sample


def add_noise(alphas_cumprod, original, noise, timesteps):
    # schedulers/lcm_single_step_scheduler.py:492-513 (copy of DDPMScheduler.add_noise)
    t = torch.as_tensor(timesteps, dtype=torch.int64).reshape(-1)
    acp = alphas_cumprod.to(original.dtype)
    sa = _bcast(acp[t] ** 0.5, original.ndim)
    sb = _bcast((1 - acp[t]) ** 0.5, original.ndim)
    return sa * original + sb * noise


def ddpm_step(alphas_cumprod, model_output, t, sample, num_inference_steps, noise=None,
              num_train_timesteps=1000, prev_t=None):
    """DDPM ancestral step, epsilon prediction, variance 'fixed_small', no clipping.

    Returns (prev_sample, pred_original_sample).  `noise` is the N(0,1) draw the reference takes
    from `generator` iff t > 0; pass it explicitly.
    """
    t = int(t)
    if prev_t is None:          # custom timetables (diffusers DDPM `previous_timestep`): the next entry of the list, -1 at the end
        prev_t = t - num_train_timesteps // num_inference_steps
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t = 1 - a_t
    b_prev = 1 - a_prev
    cur_alpha = a_t / a_prev
    cur_beta = 1 - cur_alpha
    x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    coef_x0 = (a_prev ** 0.5 * cur_beta) / b_t
    coef_xt = cur_alpha ** 0.5 * b_prev / b_t
    prev = coef_x0 * x0 + coef_xt * sample
    if t > 0:
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
        prev = prev + (var ** 0.5) * noise
    return prev, x0


def ddim_step(alphas_cumprod, model_output, t, sample, num_inference_steps, eta=0.0, noise=None,
              num_train_timesteps=1000, final_alpha_cumprod=None):
    """DDIM step (train_previewer_lora.py:194-219 for the eta=0 direction term).

    final_alpha_cumprod: value used when prev_t < 0 (`set_alpha_to_one=False` -> alphas_cumprod[0]).
    """
    t = int(t)
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = alphas_cumprod[t]
    if final_alpha_cumprod is None:
        final_alpha_cumprod = alphas_cumprod[0]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else final_alpha_cumprod
    b_t = 1 - a_t
    x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
    std = eta * var ** 0.5
    direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
    prev = a_prev ** 0.5 * x0 + direction
    if eta > 0:
        prev = prev + std * noise
    return prev, x0


def gating_tables(n, control_guidance_start=0.0, control_guidance_end=1.0, preview_start=0.0, preview_end=1.0):
    """pipelines/sdxl_instantir.py:1415-1421: Python double division and compares, reproduced as is."""
    keep, prev = [], []
    for i in range(n):
        keep.append(1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end))
        prev.append(1.0 - float(i / n < preview_start or (i + 1) / n > preview_end))
    return keep, prev
