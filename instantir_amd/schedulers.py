"""Noise schedulers of the InstantIR path, MI355X host side.

Host logic (integer timetables, fp32 coefficient tables) is plain Python/torch-CPU; tensor updates
run through the HIP kernels `iir_sched_step_f32` / `iir_axpby_f32` / `iir_lcm_step` and, inside the
denoising loop, fused with classifier-free guidance in `iir_sched_step`.

API mirrors what `pipelines/sdxl_instantir.py` touches on a scheduler object:
`set_timesteps`, `timesteps`, `config.{steps_offset,num_train_timesteps}`, `order`,
`init_noise_sigma`, `scale_model_input`, `add_noise`, `step(...) -> .prev_sample /
.pred_original_sample`, `alphas_cumprod`, `from_config`.

* `LCMSingleStepScheduler`: schedulers/lcm_single_step_scheduler.py:194-249 (tables), :331-399
  (set_timesteps and its ValueErrors), :401-407, :421-489 (step), :492-513 (add_noise).
* `DDPMScheduler` / `DDIMScheduler`: diffusers-0.28.1 classes constructed at infer.py:137 from SDXL's
  scheduler_config.json: scaled_linear betas 0.00085..0.012, 1000 train
  steps, steps_offset 1, "leading" spacing, epsilon prediction, fixed_small variance, no clipping.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch


class _Config(dict):
    __getattr__ = dict.__getitem__


class SchedulerOutput(SimpleNamespace):
    """`.prev_sample`, `.pred_original_sample` (and `.denoised` for the LCM scheduler)."""

    def __getitem__(self, i):
        return tuple(self.__dict__.values())[i]


_SDXL_DEFAULTS = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                      steps_offset=1, timestep_spacing="leading", prediction_type="epsilon", clip_sample=False,
                      set_alpha_to_one=False)


def _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    elif beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    else:
        raise NotImplementedError(f"{beta_schedule} does is not implemented")
    return torch.cumprod(1.0 - betas, dim=0), betas


def _dev_coef(vals, device):
    return torch.tensor(vals, dtype=torch.float32).to(device)


class _Base:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, **kw):
        cfg = dict(_SDXL_DEFAULTS)
        cfg.update(kw)
        self.config = _Config(cfg)
        self.alphas_cumprod, self.betas = _alphas_cumprod(cfg["num_train_timesteps"], cfg["beta_start"], cfg["beta_end"],
                                                          cfg["beta_schedule"])
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg.get("set_alpha_to_one", False) else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, cfg["num_train_timesteps"])[::-1].copy().astype(np.int64))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kw):
        """`DDPMScheduler.from_pretrained(sdxl_path, subfolder="scheduler")` (infer.py:137): reads
        `<dir>[/<subfolder>]/scheduler_config.json`; local directories only."""
        import json
        import os
        d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        path = os.path.join(d, "scheduler_config.json")
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path} not found (hub ids cannot be fetched: no network)")
        with open(path) as f:
            return cls.from_config(json.load(f), **kw)

    @classmethod
    def from_config(cls, config, **kw):
        import inspect
        names = set(_SDXL_DEFAULTS) | (set(inspect.signature(cls.__init__).parameters) - {"self", "kw"})
        args = {k: v for k, v in dict(config).items() if k in names}
        args.update(kw)
        return cls(**args)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps: Optional[List[int]] = None):
        T = self.config.num_train_timesteps
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError("Can only pass one of `num_inference_steps` or `custom_timesteps`.")
        if timesteps is not None:
            for i in range(1, len(timesteps)):
                if timesteps[i] >= timesteps[i - 1]:
                    raise ValueError("`custom_timesteps` must be in descending order.")
            if timesteps[0] >= T:
                raise ValueError(f"`timesteps` must start before `self.config.train_timesteps`: {T}.")
            ts = np.array(timesteps, dtype=np.int64)
            self.custom_timesteps = True
            self.num_inference_steps = len(ts)
        else:
            if num_inference_steps > T:
                raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                                 f"`self.config.train_timesteps`: {T}")
            self.num_inference_steps = num_inference_steps
            self.custom_timesteps = False
            # "leading": (i * floor(T/N))[::-1] + steps_offset, int64  
            step_ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def _prev_timestep(self, t):
        if getattr(self, "custom_timesteps", False):
            idx = (self.timesteps.cpu() == t).nonzero(as_tuple=True)[0][0]
            return -1 if idx == len(self.timesteps) - 1 else int(self.timesteps[idx + 1])
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(abar_t) x + sqrt(1-abar_t) noise, one t per batch row.  CUDA fp32 tensors go through the HIP
        kernel; anything else is computed with the same fp32 formula on its own device (host plumbing)."""
        acp = self.alphas_cumprod
        t = torch.as_tensor(timesteps).reshape(-1).cpu().long()
        sa, sb = acp[t] ** 0.5, (1 - acp[t]) ** 0.5
        if original_samples.is_cuda and original_samples.dtype == torch.float32 and bool((t == t[0]).all()):
            from . import ops
            out = torch.empty_like(original_samples)
            ops.axpby_f32(original_samples.contiguous(), noise.contiguous().float(),
                          _dev_coef([float(sa[0]), float(sb[0])], original_samples.device), out)
            return out
        shape = (-1,) + (1,) * (original_samples.dim() - 1)
        sa = sa.to(original_samples.device, original_samples.dtype).reshape(shape)
        sb = sb.to(original_samples.device, original_samples.dtype).reshape(shape)
        return sa * original_samples + sb * noise

    # coefficients of prev = k_x0*x0 + k_x*x + k_eps*eps + k_noise*noise, x0 = (x - sb*eps)/sa
    def step_coefficients(self, t, **kw):
        raise NotImplementedError

    def step(self, model_output, timestep, sample, eta=None, generator=None, variance_noise=None, return_dict=True, **kw):
        from . import ops
        t = int(timestep)
        c = self.step_coefficients(t, eta=eta if eta is not None else 0.0)
        need_noise = c[6] != 0.0
        if need_noise and variance_noise is None:
            variance_noise = torch.randn(model_output.shape, generator=generator,
                                         device=generator.device if generator is not None else model_output.device,
                                         dtype=torch.float32).to(model_output.device)
        if not sample.is_cuda:
            raise RuntimeError("scheduler.step: tensors must live on the GPU (the update runs in the HIP library)")
        x = sample.float().contiguous()
        e = model_output.float().contiguous()
        prev, x0 = torch.empty_like(x), torch.empty_like(x)
        ops.sched_step_f32(e, x, _dev_coef(c, x.device), prev, noise=variance_noise if need_noise else None, x0_out=x0)
        prev, x0 = prev.to(sample.dtype), x0.to(sample.dtype)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class DDPMScheduler(_Base):
    """Ancestral sampler used by infer.py:137.  variance_type fixed_small."""

    def step_coefficients(self, t, **kw):
        acp = self.alphas_cumprod
        prev_t = self._prev_timestep(t)
        a_t = acp[t]
        a_prev = acp[prev_t] if prev_t >= 0 else torch.tensor(1.0)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        k_x0 = (a_prev ** 0.5 * cur_beta) / b_t
        k_x = cur_alpha ** 0.5 * b_prev / b_t
        k_noise = torch.tensor(0.0)
        if t > 0:
            var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
            k_noise = var ** 0.5
        return [0.0, float(b_t ** 0.5), float(a_t ** 0.5), float(k_x0), float(k_x), 0.0, float(k_noise), 0.0]


class DDIMScheduler(_Base):
    """Deterministic (eta = 0) benchmark / parity default."""

    def step_coefficients(self, t, eta=0.0, **kw):
        acp = self.alphas_cumprod
        prev_t = self._prev_timestep(t)
        a_t = acp[t]
        a_prev = acp[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        k_eps = (1 - a_prev - std ** 2) ** 0.5
        return [0.0, float(b_t ** 0.5), float(a_t ** 0.5), float(a_prev ** 0.5), 0.0, float(k_eps),
                float(std) if eta > 0 else 0.0, 0.0]


class LCMSingleStepScheduler(_Base):
    """One-step x0 preview with LCM boundary scalings; never touches `self.timesteps` in `step`."""

    def __init__(self, original_inference_steps: int = 50, timestep_scaling: float = 10.0, **kw):
        kw.setdefault("steps_offset", 0)
        kw.setdefault("set_alpha_to_one", True)
        super().__init__(**kw)
        self.config["original_inference_steps"] = original_inference_steps
        self.config["timestep_scaling"] = timestep_scaling
        self.sigma_data = 0.5

    def set_timesteps(self, num_inference_steps=None, device=None, original_inference_steps=None, strength=1.0,
                      timesteps=None):
        T = self.config.num_train_timesteps
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError("Can only pass one of `num_inference_steps` or `custom_timesteps`.")
        if timesteps is not None:
            for i in range(1, len(timesteps)):
                if timesteps[i] >= timesteps[i - 1]:
                    raise ValueError("`custom_timesteps` must be in descending order.")
            if timesteps[0] >= T:
                raise ValueError(f"`timesteps` must start before `self.config.train_timesteps`: {T}.")
            ts = np.array(timesteps, dtype=np.int64)
        else:
            if num_inference_steps > T:
                raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {T}")
            self.num_inference_steps = num_inference_steps
            orig = original_inference_steps if original_inference_steps is not None else self.config.original_inference_steps
            if orig > T:
                raise ValueError(f"`original_steps`: {orig} cannot be larger than {T}")
            if num_inference_steps > orig:
                raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                                 f"`original_inference_steps`: {orig}")
            c = T // orig
            origin = np.asarray(list(range(1, int(orig * strength) + 1))) * c - 1
            skipping = len(origin) // num_inference_steps
            ts = origin[::-skipping][:num_inference_steps]
        self.timesteps = torch.from_numpy(np.asarray(ts).copy()).to(device=device, dtype=torch.long)

    def get_scalings_for_boundary_condition_discrete(self, timestep):
        st = timestep * self.config.timestep_scaling
        c_skip = self.sigma_data ** 2 / (st ** 2 + self.sigma_data ** 2)
        c_out = st / (st ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out

    def preview_coefficients(self, t):
        """{sqrt(1-abar_t), sqrt(abar_t), c_out, c_skip} in fp32, for iir_lcm_step."""
        tt = torch.tensor([int(t)], dtype=torch.int64)
        a = self.alphas_cumprod.gather(-1, tt)
        c_skip, c_out = self.get_scalings_for_boundary_condition_discrete(tt)
        return [float(torch.sqrt(1 - a)), float(torch.sqrt(a)), float(c_out), float(c_skip)]

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        """denoised = c_out * x0 + c_skip * sample, on the GPU via the scheduler kernel:
        prev = k_x0*x0 + k_x*x with k_x0 = c_out, k_x = c_skip."""
        from . import ops
        if not sample.is_cuda:
            raise RuntimeError("LCMSingleStepScheduler.step: tensors must live on the GPU")
        sb, sa, c_out, c_skip = self.preview_coefficients(int(timestep))
        x, e = sample.float().contiguous(), model_output.float().contiguous()
        out = torch.empty_like(x)
        ops.sched_step_f32(e, x, _dev_coef([0.0, sb, sa, c_out, c_skip, 0.0, 0.0, 0.0], x.device), out)
        out = out.to(sample.dtype)
        if not return_dict:
            return (out,)
        return SchedulerOutput(denoised=out)
