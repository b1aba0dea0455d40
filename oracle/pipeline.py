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
            guidance_rescale=0.0, negative_time_ids=None):
    """Returns the final latents (B,4,h,w).  `lq`: LQ latent (B,4,h,w); `image_embeds`: (2,B,S,E) [neg;pos] under CFG
    (pipelines/sdxl_instantir.py:700-707) else (1,B,S,E).  `trace` (dict) collects per-step tensors when given."""
    B = lq.shape[0]
    do_cfg = guidance_scale > 1                                            # :1050-1051
    acp = sched.make_alphas_cumprod()
    ts = sched.leading_timesteps(num_inference_steps)                      # :1385
    n = len(ts)
    keep, previewing = sched.gating_tables(n, control_guidance_start, control_guidance_end, preview_start, preview_end)
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
    down = mid = None
    for i, t in enumerate(ts):
        t = int(t)
        xin = torch.cat([x] * 2) if do_cfg else x                            # :1503
        cond_scale = min(max(1.0, 0.0), controlnet_conditioning_scale) * keep[i]   # :1538-1539 with preview_factor = 1
        if cond_scale > 0.1:                                                 # :1542
            if previewing[i] > 0 and use_previewer:
                eps1 = nets.unet_forward(P, cfg, xin, t, ctx, text_embeds, tid, ip_prev, lora=lora)     # :1545-1554
                preview = sched.lcm_step(acp, eps1, t, xin)                  # :1555-1561
            else:
                preview = image                                              # :1581-1582
            down, mid = nets.aggregator_forward(PA, cfg, image, t, preview, text_embeds, tid)           # :1591-1599
            d, m = [s * cond_scale for s in down], mid * cond_scale          # :1602-1603
        else:
            d = m = None       # reference: previous residuals times 0
            preview = None
        eps = nets.unet_forward(P, cfg, xin, t, ctx, text_embeds, tid, ip_main, d, m)                   # :1606-1616
        if do_cfg:                                                           # :1619-1621
            u, c = eps.chunk(2)
            eps = u + guidance_scale * (c - u)
            if guidance_rescale > 0.0:                                       # rescale_noise_cfg, :181-192 (called at :1623-1626)
                dims = list(range(1, c.ndim))
                std_text, std_cfg = c.std(dim=dims, keepdim=True), eps.std(dim=dims, keepdim=True)
                eps = guidance_rescale * (eps * (std_text / std_cfg)) + (1 - guidance_rescale) * eps
        if sampler == "ddim":
            x_next, x0 = sched.ddim_step(acp, eps, t, x, n, eta=eta, noise=None if step_noises is None else step_noises[i])
        else:
            x_next, x0 = sched.ddpm_step(acp, eps, t, x, n, noise=None if step_noises is None else step_noises[i])
        if trace is not None:
            trace.setdefault("eps", []).append(eps)
            trace.setdefault("x", []).append(x_next)
            trace.setdefault("preview", []).append(preview)
        x = x_next
    return x
