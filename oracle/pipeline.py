"""Oracle (test infrastructure): fp32 CPU restatement of the InstantIR denoising loop,
pipelines/sdxl_instantir.py:1385-1660, for tensor inputs (latent LQ image, prompt / image embeds,
explicit noise tensors -- SURVEY.md Appendix B).  PARITY UNPINNED as a whole (the reference pipeline
needs diffusers/peft); its leaves are pinned individually (see oracle/__init__.py).
"""
from __future__ import annotations

import torch

from . import nets, sched


def denoise(P, PA, lora, cfg, lq, prompt_embeds, pooled, image_embeds, *, negative_prompt_embeds=None,
            negative_pooled=None, num_inference_steps=30, guidance_scale=7.0, sampler="ddim", eta=0.0,
            init_noise=None, step_noises=None, controlnet_conditioning_scale=1.0, control_guidance_start=0.0,
            control_guidance_end=1.0, preview_start=0.0, preview_end=1.0, use_previewer=True, trace=None,
            guidance_rescale=0.0, negative_time_ids=None, adastep_restore=False, reference_latents=None,
            denoising_end=None, timesteps=None, resume=None, on_step=None):
    """Returns the final latents (B,4,h,w).  `lq`: LQ latent (B,4,h,w); `image_embeds`: (2,B,S,E) [neg;pos] under CFG
    (pipelines/sdxl_instantir.py:700-707) else (1,B,S,E).  `trace` (dict) collects per-step tensors when given.
    The loop body follows :1497-1660 statement by statement, including the quirk that a step which skips the
    previewer / Aggregator re-scales the PREVIOUS step's already scaled residuals (:1602-1603).
    Test plumbing (no reference counterpart): `resume = (x, i0, i1)` continues a chain from the latents `x` it had before step
    i0 and stops before step i1 (a default-settings chain carries no other state from step to step: tools/parity_chain30.py runs
    30 steps at 1024^2 in several sittings); `on_step(i, x)` is called with the latents after every step."""
    B = lq.shape[0]
    do_cfg = guidance_scale > 1                                            # :1050-1051
    acp = sched.make_alphas_cumprod()
    ts = sched.leading_timesteps(num_inference_steps) if timesteps is None else list(timesteps)      # :1385, :225-237
    n = len(ts)
    keep, previewing = sched.gating_tables(n, control_guidance_start, control_guidance_end, preview_start, preview_end)
    ccs = controlnet_conditioning_scale if isinstance(controlnet_conditioning_scale, list) else [controlnet_conditioning_scale] * n
    if denoising_end is not None and isinstance(denoising_end, float) and 0 < denoising_end < 1:      # :1470-1483
        cutoff = int(round(1000 - denoising_end * 1000))
        ts = [t for t in ts if int(t) >= cutoff]
    hpx, wpx = lq.shape[2] * 8, lq.shape[3] * 8
    tid = torch.tensor([[hpx, wpx, 0, 0, hpx, wpx]], dtype=torch.float32)    # :965-981
    if do_cfg:                                                             # :1456-1464
        if negative_prompt_embeds is None:
            negative_prompt_embeds, negative_pooled = torch.zeros_like(prompt_embeds), torch.zeros_like(pooled)
        ctx = torch.cat([negative_prompt_embeds, prompt_embeds])
        text_embeds = torch.cat([negative_pooled, pooled])
        image = torch.cat([lq] * 2)
    else:
        ctx, text_embeds, image = prompt_embeds, pooled, lq
    R = ctx.shape[0]
    if do_cfg and negative_time_ids is not None:                           # :1445-1464: cat([neg, pos]).repeat(B, 1), row order as is
        tid = torch.cat([torch.tensor([list(negative_time_ids)], dtype=torch.float32), tid]).repeat(B, 1)
    else:
        tid = tid.repeat(R, 1)
    ip_main = nets.image_projection(P, [image_embeds], cfg.resampler)[0]
    ip_prev = nets.image_projection(P, [image_embeds], cfg.resampler, lora)[0] if lora is not None else None
    x = sched.add_noise(acp, lq, init_noise, [int(ts[0])] * B)               # :1389, :931-939
    previewer_mean = torch.zeros_like(x)                                   # :1488
    preview_factor = torch.ones(B, 1, 1, 1)                                # :1490-1492
    down = mid = preview_latent = None
    n_steps = num_inference_steps if timesteps is None else n
    i0, i1 = 0, len(ts)
    if resume is not None:
        if adastep_restore:
            raise ValueError("resume: adastep_restore carries previewer_mean / preview_factor across steps")
        x = x if resume[0] is None else resume[0].clone()               # (None: from the noised LQ latent, i.e. only the stop index applies)
        i0, i1 = int(resume[1]), min(int(resume[2]), len(ts))
    for i in range(i0, i1):
        t = int(ts[i])
        xin = torch.cat([x] * 2) if do_cfg else x                            # :1503
        ada = preview_factor.clamp(0.0, ccs[i])                              # :1538
        cond_scale = ada * keep[i]                                           # :1539
        cond_scale = torch.cat([cond_scale] * 2) if do_cfg else cond_scale   # :1540
        if (cond_scale > 0.1).sum().item() > 0:                              # :1542
            if previewing[i] > 0 and use_previewer:
                eps1 = nets.unet_forward(P, cfg, xin, t, ctx, text_embeds, tid, ip_prev, lora=lora)     # :1545-1554
                preview_latent = sched.lcm_step(acp, eps1, t, xin)           # :1555-1561
            elif reference_latents is not None:                              # :1579-1580
                preview_latent = torch.cat([reference_latents] * 2) if do_cfg else reference_latents
            else:
                preview_latent = image                                       # :1581-1582
            down, mid = nets.aggregator_forward(PA, cfg, image, t, preview_latent, text_embeds, tid)    # :1591-1599
        if down is None:
            raise NameError("down_block_res_samples")                       # the reference's own failure on step 0 (Q2)
        down = [s * cond_scale for s in down]                                # :1602  (re-scales stale residuals when skipped)
        mid = mid * cond_scale                                               # :1603
        eps = nets.unet_forward(P, cfg, xin, t, ctx, text_embeds, tid, ip_main, down, mid)              # :1606-1616
        if do_cfg:                                                           # :1619-1621
            u, c = eps.chunk(2)
            eps = u + guidance_scale * (c - u)
            if guidance_rescale > 0.0:                                       # rescale_noise_cfg, :181-192 (called at :1623-1626)
                dims = list(range(1, c.ndim))
                std_text, std_cfg = c.std(dim=dims, keepdim=True), eps.std(dim=dims, keepdim=True)
                eps = guidance_rescale * (eps * (std_text / std_cfg)) + (1 - guidance_rescale) * eps
        if sampler == "ddim":
            x_next, x0 = sched.ddim_step(acp, eps, t, x, n_steps, eta=eta, noise=None if step_noises is None else step_noises[i])
        else:
            prev_t = None if timesteps is None else (int(ts[i + 1]) if i + 1 < len(ts) else -1)
            x_next, x0 = sched.ddpm_step(acp, eps, t, x, n_steps, noise=None if step_noises is None else step_noises[i], prev_t=prev_t)
        if adastep_restore:                                                  # :1636-1644 (needs CFG: Q6)
            pv = preview_latent[B:].float()
            pred_x0_l2 = (pv - x0.float()).pow(2).sum(dim=(1, 2, 3))
            previewer_l2 = (pv - previewer_mean.float()).pow(2).sum(dim=(1, 2, 3))
            previewer_mean = preview_latent[B:]
            preview_factor = (pred_x0_l2 / previewer_l2).reshape(-1, 1, 1, 1)
        if trace is not None:
            trace.setdefault("eps", []).append(eps)
            trace.setdefault("x", []).append(x_next)
            trace.setdefault("preview", []).append(preview_latent)
            trace.setdefault("cond_scale", []).append(cond_scale.flatten().clone())
            trace.setdefault("preview_factor", []).append(preview_factor.flatten().clone())
        x = x_next
        if on_step is not None:
            on_step(i, x)
    return x
