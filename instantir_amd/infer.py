"""Command line restoration driver with the flags and behaviour of the reference's `infer.py`:

    python -m instantir_amd.infer --sdxl_path SDXL --vision_encoder_path DINOV2 --instantir_path INSTANTIR \\
        --test_path INPUTS --out_path OUT [--cfg 7.0 --preview_start 0.0 --creative_start 1.0 ...]

Mirrors: model assembly `infer.py:114-144`, batching + skip-done logic :148-169, `resize_img` :31-66, default
prompts :192-205 (byte for byte: the backslash continuations put the runs of spaces INSIDE the prompt text,
SURVEY.md Appendix C Q10), the pipeline call :211-222, resize-back + save :224-225.  `--denoising_start` is accepted
and has no effect on the schedule, as in the reference (Q1).  Flags the reference parses but never uses
(`--adapter_tokens --resolution --variant --revision --pretrained_vae_model_name_or_path`) are accepted too.

Extension (not in the reference): `--synthetic {tiny,sdxl}` builds every network from seeded synthetic weights so the
driver can be exercised without checkpoints (none are available offline); with it the text prompt is replaced by
seeded embeddings because no tokenizer vocabulary exists here.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch
from PIL import Image

DEFAULT_PROMPT = "Photorealistic, highly detailed, hyper detailed photo - realistic maximum detail, 32k, \
                ultra HD, extreme meticulous detailing, skin pore detailing, \
                hyper sharpness, perfect without deformations, \
                taken using a Canon EOS R camera, Cinematic, High Contrast, Color Grading. "
DEFAULT_NEG_PROMPT = "blurry, out of focus, unclear, depth of field, over-smooth, \
                sketch, oil painting, cartoon, CG Style, 3D render, unreal engine, \
                dirty, messy, worst quality, low quality, frames, painting, illustration, drawing, art, \
                watermark, signature, jpeg artifacts, deformed, lowres"


def resize_img(input_image, max_side=1024, min_side=768, width=None, height=None, pad_to_max_side=False,
               mode=Image.BILINEAR, base_pixel_number=64):
    """infer.py:31-66: requested output size, then min side -> 768, max side -> 1024, floor to a multiple of 64.
    Returns (resized image, (out_w, out_h)) -- the second item is what the result is resized back to."""
    w, h = input_image.size
    if width is not None and height is not None:
        out_w, out_h = width, height
    elif width is not None:
        out_w, out_h = width, round(h * width / w)
    elif height is not None:
        out_w, out_h = round(w * height / h), height
    else:
        out_w, out_h = w, h
    w, h = out_w, out_h
    if min(w, h) < min_side:
        r = min_side / min(w, h)
        w, h = round(r * w), round(r * h)
    if max(w, h) > max_side:
        r = max_side / max(w, h)
        w, h = round(r * w), round(r * h)
    wn, hn = (w // base_pixel_number) * base_pixel_number, (h // base_pixel_number) * base_pixel_number
    input_image = input_image.resize([wn, hn], mode)
    if pad_to_max_side:
        canvas = np.ones([max_side, max_side, 3], dtype=np.uint8) * 255
        ox, oy = (max_side - wn) // 2, (max_side - hn) // 2
        canvas[oy:oy + hn, ox:ox + wn] = np.array(input_image)
        input_image = Image.fromarray(canvas)
    return input_image, (out_w, out_h)


def plan_batches(test_path, out_dir, batch_size):
    """infer.py:148-169: sorted inputs, files already present in the output directory are skipped."""
    done = set(os.listdir(out_dir))
    names = [test_path.split("/")[-1]] if os.path.isfile(test_path) else sorted(os.listdir(test_path))
    batches, cur = [], []
    for f in sorted(names):
        if f in done:
            print(f"Skip {f}")
            continue
        cur.append(f)
        if len(cur) == batch_size:
            batches.append(cur)
            cur = []
    if cur:
        batches.append(cur)
    return batches


def shard_batches(batches, rank, world):
    """Multi-GPU: the batch list infer.py:151-169 forms is cut into contiguous per-rank slices;
    images never interact, so each rank runs the whole loop on its slice and there is no per-step collective."""
    from .parallel import shard_range
    lo, hi = shard_range(len(batches), rank, world)
    return batches[lo:hi]


def build_parser():
    p = argparse.ArgumentParser(description="InstantIR restoration on MI355X")
    p.add_argument("--sdxl_path", type=str, default=None)
    p.add_argument("--previewer_lora_path", type=str, default=None)
    p.add_argument("--pretrained_vae_model_name_or_path", type=str, default=None)
    p.add_argument("--instantir_path", type=str, default=None)
    p.add_argument("--vision_encoder_path", type=str, default="/share/huangrenyuan/model_zoo/vis_backbone/dinov2_large")
    p.add_argument("--adapter_model_path", type=str, default=None)
    p.add_argument("--adapter_tokens", type=int, default=64)
    p.add_argument("--use_clip_encoder", action="store_true")
    p.add_argument("--denoising_start", type=int, default=1000)
    p.add_argument("--num_inference_steps", type=int, default=30)
    p.add_argument("--creative_start", type=float, default=1.0)
    p.add_argument("--preview_start", type=float, default=0.0)
    p.add_argument("--resolution", type=int, default=1024)
    p.add_argument("--batch_size", type=int, default=6)
    p.add_argument("--width", type=int, default=None)
    p.add_argument("--height", type=int, default=None)
    p.add_argument("--cfg", type=float, default=7.0)
    p.add_argument("--post_fix", type=str, default=None)
    p.add_argument("--variant", type=str, default="fp16")
    p.add_argument("--revision", type=str, default=None, required=False)
    p.add_argument("--prompt", type=str, default="", nargs="+")
    p.add_argument("--neg_prompt", type=str, default="", nargs="+")
    p.add_argument("--test_path", type=str, default=None, required=True)
    p.add_argument("--out_path", type=str, default="./output")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--synthetic", choices=["tiny", "sdxl"], default=None, help="seeded synthetic weights (no checkpoints needed)")
    return p


def build_pipeline(args, device):
    """infer.py:117-144 against this build's classes."""
    from . import loaders, weights as W
    from .config import UNetConfig, VAEConfig
    from .encoders import HipCLIPText, HipDinov2
    from .pipeline import InstantIRPipeline
    from .schedulers import DDPMScheduler, LCMSingleStepScheduler
    from .vae import HipVAE
    if args.synthetic:
        cfg = UNetConfig.tiny() if args.synthetic == "tiny" else UNetConfig.sdxl()
        vc = VAEConfig.tiny() if args.synthetic == "tiny" else VAEConfig.sdxl()
        gen = "cpu" if args.synthetic == "tiny" else device
        unet_sd = W.synth_state_dict(W.unet_specs(cfg), 1234, device=gen)
        agg_sd = W.synth_state_dict(W.aggregator_specs(cfg), 1235, device=gen)
        lora, alpha = W.synth_state_dict(W.lora_specs(cfg), 1236, device=gen), max(1, cfg.lora_rank // 8)
        vae = HipVAE(vc, W.synth_state_dict(W.vae_decoder_specs(vc) + W.vae_encoder_specs(vc), 1237, device=gen), device)
        pipe = InstantIRPipeline(cfg, unet_sd, scheduler=DDPMScheduler(), vae=vae, device=device)
        pipe.prepare_previewers(lora, lora_alpha=alpha)
        pipe.aggregator.load_state_dict(agg_sd)
        return pipe, LCMSingleStepScheduler.from_config(pipe.scheduler.config)
    if not args.sdxl_path or not args.instantir_path:
        raise SystemExit("--sdxl_path and --instantir_path are required (or use --synthetic)")
    # infer.py:117-144, call for call
    print("Initializing pipeline...")
    pipe = InstantIRPipeline.from_pretrained(args.sdxl_path, torch_dtype=torch.float16, device=device)
    print("Loading LQ-Adapter...")
    adapter = args.adapter_model_path if args.adapter_model_path is not None else os.path.join(args.instantir_path, "adapter.pt")
    # (the reference parses --adapter_tokens but never forwards it, infer.py:124-129,269; forwarded here: default 64 is the same call)
    loaders.load_adapter_to_pipe(pipe, adapter, args.vision_encoder_path, use_clip_encoder=args.use_clip_encoder,
                                 adapter_tokens=args.adapter_tokens)
    lora_path = args.previewer_lora_path if args.previewer_lora_path is not None else args.instantir_path
    lora_alpha = pipe.prepare_previewers(lora_path)
    print(f"use lora alpha {lora_alpha}")
    pipe.scheduler = DDPMScheduler.from_pretrained(args.sdxl_path, subfolder="scheduler") \
        if os.path.isfile(os.path.join(args.sdxl_path, "scheduler", "scheduler_config.json")) else DDPMScheduler()
    lcm_scheduler = LCMSingleStepScheduler.from_config(pipe.scheduler.config)
    print("Loading checkpoint...")
    pipe.aggregator.load_state_dict(loaders.read_aggregator(os.path.join(args.instantir_path, "aggregator.pt")))
    return pipe, lcm_scheduler


def main(args, device, rank=0, world=1):
    pipe, lcm_scheduler = build_pipeline(args, device)
    post_fix = f"_{args.post_fix}" if args.post_fix else ""
    out_dir = f"{args.out_path}/{post_fix}"
    os.makedirs(out_dir, exist_ok=True)
    cfg = pipe.cfg
    batches = plan_batches(args.test_path, os.path.join(args.out_path, post_fix), args.batch_size)
    for lq_batch in shard_batches(batches, rank, world):
        generator = torch.Generator(device=device).manual_seed(args.seed)                # infer.py:172: fresh per batch
        lq, out_sizes = [], []
        for name in lq_batch:
            path = args.test_path if os.path.isfile(args.test_path) else os.path.join(args.test_path, name)
            im, out_size = resize_img(Image.open(path).convert("RGB"), width=args.width, height=args.height)
            lq.append(im)
            out_sizes.append(out_size)
        if len({im.size for im in lq}) != 1:
            raise ValueError("images of one batch must share a size after resize_img (use --width/--height or --batch_size 1)")
        # infer.py:191-210: `--prompt` / `--neg_prompt` are nargs="*" LISTS and the reference multiplies the list by the batch
        # (`prompt = args.prompt * len(lq)`): `--prompt "a photo" "a drawing"` with one image asks for two restorations, one per
        # entry; with several images AND several entries the pipeline's own batch check (:1316-1319) rejects the call.
        prompts = (list(args.prompt) if args.prompt else [DEFAULT_PROMPT]) * len(lq)
        negs = (list(args.neg_prompt) if args.neg_prompt else [DEFAULT_NEG_PROMPT]) * len(lq)
        kw = dict(image=lq, num_inference_steps=args.num_inference_steps, generator=generator, guidance_scale=args.cfg,
                  previewer_scheduler=lcm_scheduler, preview_start=args.preview_start, control_guidance_end=args.creative_start)
        if args.synthetic:
            g = torch.Generator().manual_seed(args.seed)
            n = len(lq)
            kw.update(prompt_embeds=torch.randn(n, cfg.text_len, cfg.cross_attention_dim, generator=g),
                      pooled_prompt_embeds=torch.randn(n, cfg.pooled_dim, generator=g),
                      negative_prompt_embeds=torch.randn(n, cfg.text_len, cfg.cross_attention_dim, generator=g),
                      negative_pooled_prompt_embeds=torch.randn(n, cfg.pooled_dim, generator=g),
                      ip_adapter_image_embeds=[torch.randn(2 if args.cfg > 1 else 1, n, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)])
        else:
            kw.update(prompt=prompts, negative_prompt=negs, ip_adapter_image=lq)
        images = pipe(**kw).images
        for i, (rec, out_size) in enumerate(zip(images, out_sizes)):
            rec.resize([out_size[0], out_size[1]], Image.BILINEAR).save(f"{out_dir}/{lq_batch[i]}")


if __name__ == "__main__":
    a = build_parser().parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("instantir_amd.infer needs an MI355X: the restoration path has no CPU fallback")
    # one process per GPU: `python -m torch.distributed.run --nproc-per-node N -m instantir_amd.infer ...` shards the inputs
    from .parallel import barrier, init_from_env
    rank_, world_, _, dev_ = init_from_env()
    main(a, dev_, rank_, world_)
    if world_ > 1:
        barrier()
        torch.distributed.destroy_process_group()
