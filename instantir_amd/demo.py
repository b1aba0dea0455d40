"""The request handler behind the reference's web demo, without the web UI.

`gradio_demo/app.py:110-156` is the pipeline's second caller: it converts slider values into the fractions the pipeline
takes, switches the UNet between the `previewer` and `lcm` LoRA adapters per request, builds its own evenly spaced
timestep list and always asks for the preview row.  Those behaviours are reproduced here as plain functions over an
`InstantIRPipeline`; the gradio widgets themselves (sliders, gallery, queue) are UI and out of scope (out-of-scope
list).  `pipe_kwargs` lets a caller without text encoders pass `prompt_embeds` etc. instead of prompt strings.
"""
from __future__ import annotations

import torch
from PIL import Image

from .infer import DEFAULT_NEG_PROMPT, DEFAULT_PROMPT


def slider_to_fraction(value, steps: int):
    """gradio_demo/app.py:121-128: the 'Start Free Rendering' / 'Restoration Previews' sliders hand over step COUNTS; an int,
    or a float above 1.0, is divided by the number of steps; a float in [0, 1] already is the fraction."""
    if isinstance(value, int) and not isinstance(value, bool):
        return value / steps
    if value > 1.0:
        return value / steps
    return value


def demo_timesteps(steps: int, steps_offset: int = 1):
    """gradio_demo/app.py:131-134: `i * (1000 // steps) + steps_offset`, descending -- the 'leading' spacing written out by
    hand and passed as `timesteps=` (so the pipeline's own `set_timesteps(num_inference_steps)` spacing is bypassed)."""
    return [i * (1000 // steps) + steps_offset for i in range(steps)][::-1]


def demo_resize(image: Image.Image, size=None, max_side=1280, base_pixel_number=64, mode=Image.BILINEAR):
    """gradio_demo/app.py:16-29 (`pad_to_max_side` is never used by the demo): an explicit (width, height), else the long side
    to `max_side` and both sides floored to a multiple of 64."""
    w, h = image.size
    if size is not None:
        wn, hn = size
    else:
        r = max_side / max(h, w)
        image = image.resize([round(r * w), round(r * h)], mode)
        wn, hn = (round(r * w) // base_pixel_number) * base_pixel_number, (round(r * h) // base_pixel_number) * base_pixel_number
    return image.resize([wn, hn], mode)


@torch.no_grad()
def instantir_restore(pipe, lcm_scheduler, lq, prompt="", steps=30, cfg_scale=7.0, guidance_end=1.0, creative_restoration=False,
                      seed=3407, height=1024, width=1024, preview_start=0.0, **pipe_kwargs):
    """gradio_demo/app.py:110-156.  Returns (restored image, preview row) like the demo's handler; every preview entry gets
    its gallery caption appended (`preview_{i}`, :154-155)."""
    want = "lcm" if creative_restoration else "previewer"                     # :114-120
    if want not in pipe.unet.active_adapters():
        pipe.unet.set_adapter(want)
    guidance_end = slider_to_fraction(guidance_end, steps)
    preview_start = slider_to_fraction(preview_start, steps)
    if isinstance(lq, Image.Image):
        lq = [demo_resize(lq.convert("RGB"), size=(width, height))]
    generator = torch.Generator(device=pipe.device).manual_seed(seed)
    timesteps = demo_timesteps(steps, pipe.scheduler.config.steps_offset)
    kw = dict(image=lq, num_inference_steps=steps, generator=generator, timesteps=timesteps, guidance_scale=cfg_scale,
              control_guidance_end=guidance_end, preview_start=preview_start, previewer_scheduler=lcm_scheduler,
              return_dict=False, save_preview_row=True)
    if "prompt_embeds" not in pipe_kwargs:
        n = len(lq) if isinstance(lq, (list, tuple)) else lq.shape[0]
        kw.update(prompt=[DEFAULT_PROMPT if len(prompt) == 0 else prompt] * n, negative_prompt=[DEFAULT_NEG_PROMPT] * n)
    kw.update(pipe_kwargs)
    out = pipe(**kw)
    for i, preview_img in enumerate(out[1]):
        preview_img.append(f"preview_{i}")
    return out[0][0], out[1]
