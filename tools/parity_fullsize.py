"""One-off parity run at the BENCHMARK geometry (BASELINE configs[1]: 1024x1024, cfg 7.0, full SDXL / InstantIR shapes):
HIP pipeline vs the CPU fp32 oracle for a few DDIM steps.  Too slow for the test suite (the oracle needs ~70 s per CFG step
on 16 cores); `tests/test_fullsize_gpu.py` asserts the same thing at 512 px.  Prints a heartbeat so the runner sees progress.

    python tools/parity_fullsize.py [--size 1024] [--steps 5] [--guidance 7.0]
"""
import argparse, math, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--guidance", type=float, default=7.0); ap.add_argument("--control_guidance_end", type=float, default=1.0)
ap.add_argument("--timesteps", type=int, nargs="*", default=None, help="custom timesteps (DDPM scheduler + given step noises); a lone DDIM step sits at t = 1, a weak check")
a = ap.parse_args()
from instantir_amd import lib, weights as W
from instantir_amd.config import UNetConfig
from instantir_amd.pipeline import InstantIRPipeline
from instantir_amd.schedulers import DDIMScheduler, DDPMScheduler, LCMSingleStepScheduler
from oracle import pipeline as OP

def cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError): pass
    return n

t00 = time.time()
stop = threading.Event()
threading.Thread(target=lambda: [print(f"[heartbeat {time.time() - t00:.0f} s]", flush=True) for _ in iter(lambda: stop.wait(60), True)], daemon=True).start()
lib.load()
dev = torch.device("cuda:0"); cfg = UNetConfig.sdxl()
sd = W.synth_state_dict(W.unet_specs(cfg), 1234, device=dev); sda = W.synth_state_dict(W.aggregator_specs(cfg), 1235, device=dev)
lora = W.synth_state_dict(W.lora_specs(cfg), 1236, device=dev)
g = torch.Generator().manual_seed(42); B, H = 1, a.size // 8
lq = torch.randn(B, 4, H, H, generator=g) * 0.8
pe, pooled = torch.randn(B, 77, 2048, generator=g).half().float(), torch.randn(B, 1280, generator=g).half().float()
feats = torch.randn(2 if a.guidance > 1 else 1, B, 257, 1024, generator=g).half().float()
npe, npooled = torch.randn(B, 77, 2048, generator=g).half().float(), torch.randn(B, 1280, generator=g).half().float()
noise = torch.randn(B, 4, H, H, generator=g); alpha = 8
extra, okw = {}, dict(sampler="ddim")
if a.timesteps:
    sn = [torch.randn(B, 4, H, H, generator=g) for _ in a.timesteps]
    extra, okw = dict(timesteps=a.timesteps, step_noises=sn), dict(sampler="ddpm", timesteps=a.timesteps, step_noises=sn)
pipe = InstantIRPipeline(cfg, sd, scheduler=DDPMScheduler() if a.timesteps else DDIMScheduler(), device=dev)
pipe.aggregator.load_state_dict(sda); pipe.prepare_previewers(lora, lora_alpha=alpha)
got = pipe(image=lq, prompt_embeds=pe, pooled_prompt_embeds=pooled, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=npooled,
           ip_adapter_image_embeds=[feats], output_type="latent", num_inference_steps=a.steps, guidance_scale=a.guidance, init_noise=noise, control_guidance_end=a.control_guidance_end, **extra,
           previewer_scheduler=LCMSingleStepScheduler.from_config(pipe.scheduler.config)).images.float().cpu()
print(f"HIP path done ({time.time() - t00:.0f} s), finite={bool(torch.isfinite(got).all())}; CPU oracle on {cores()} cores ...", flush=True)
torch.set_num_threads(cores())
P = {k: v.float().cpu() for k, v in sd.items()}; PA = {k: v.float().cpu() for k, v in sda.items()}; L = {k: v.float().cpu() for k, v in lora.items()}
L["scaling"] = alpha / cfg.lora_rank
del sd, sda, lora, pipe; torch.cuda.empty_cache()
with torch.no_grad():
    want = OP.denoise(P, PA, L, cfg, lq, pe, pooled, feats, negative_prompt_embeds=npe, negative_pooled=npooled, init_noise=noise,
                      num_inference_steps=a.steps, guidance_scale=a.guidance, control_guidance_end=a.control_guidance_end, **okw)
stop.set()
mse = ((got - want) ** 2).mean().item()
p = 10 * math.log10(want.abs().max().item() ** 2 / max(mse, 1e-30))
print(f"RESULT size={a.size} cfg={a.guidance} steps={a.steps} timesteps={a.timesteps} control_guidance_end={a.control_guidance_end}: latent PSNR vs CPU fp32 oracle {p:.1f} dB (max |want| {want.abs().max().item():.3f}, rmse {mse ** 0.5:.2e}); total {time.time() - t00:.0f} s", flush=True)
