"""The headline workload end to end against the oracle: BASELINE configs[1] (1024x1024, cfg 7.0, 30 DDIM steps, previewer +
Aggregator + UNet every step), HIP pipeline vs CPU fp32 oracle, ALL 30 chained steps.  The oracle needs ~75 s per step on the
GPU box's 16 cores, more than one `gpurun` call allows, so it runs in sittings: every call recomputes the (deterministic, ~2 s)
HIP chain, continues the oracle chain from the checkpoint of the previous call (the latents after step k: a default-settings
chain carries nothing else), and prints the PSNR of the latents after every step it covered.

    python tools/parity_chain30.py --ckpt tools/_chain30.pt --max-steps 10       # repeat until it prints RESULT
"""
import argparse, math, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--ckpt", default="tools/_chain30.pt"); ap.add_argument("--out", default="gpurun_out/chain30"); ap.add_argument("--max-steps", type=int, default=10)
ap.add_argument("--steps", type=int, default=30); ap.add_argument("--size", type=int, default=1024)
a = ap.parse_args()
from instantir_amd import lib, weights as W
from instantir_amd.config import UNetConfig
from instantir_amd.pipeline import InstantIRPipeline
from instantir_amd.schedulers import DDIMScheduler, LCMSingleStepScheduler
from oracle import pipeline as OP


def cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError): pass
    return n


def psnr(got, want):
    mse = ((got - want) ** 2).mean().item()
    return 10 * math.log10(want.abs().max().item() ** 2 / max(mse, 1e-30))


t00 = time.time()
os.makedirs(a.out, exist_ok=True)
stop = threading.Event()          # heartbeat: an oracle step at 2048^2 outlasts the runner's 7-minute silence limit
threading.Thread(target=lambda: [print(f"[heartbeat {time.time() - t00:.0f} s]", flush=True) for _ in iter(lambda: stop.wait(60), True)], daemon=True).start()
lib.load()
dev = torch.device("cuda:0"); cfg = UNetConfig.sdxl()
sd = W.synth_state_dict(W.unet_specs(cfg), 1234, device=dev); sda = W.synth_state_dict(W.aggregator_specs(cfg), 1235, device=dev)
lora = W.synth_state_dict(W.lora_specs(cfg), 1236, device=dev)
g = torch.Generator().manual_seed(42); B, H = 1, a.size // 8
lq = torch.randn(B, 4, H, H, generator=g) * 0.8
pe, pooled = torch.randn(B, 77, 2048, generator=g).half().float(), torch.randn(B, 1280, generator=g).half().float()
feats = torch.randn(2, B, 257, 1024, generator=g).half().float()
npe, npooled = torch.randn(B, 77, 2048, generator=g).half().float(), torch.randn(B, 1280, generator=g).half().float()
noise = torch.randn(B, 4, H, H, generator=g); alpha = 8
pipe = InstantIRPipeline(cfg, sd, scheduler=DDIMScheduler(), device=dev)
pipe.aggregator.load_state_dict(sda); pipe.prepare_previewers(lora, lora_alpha=alpha)
gpu_x = []
got = pipe(image=lq, prompt_embeds=pe, pooled_prompt_embeds=pooled, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=npooled,
           ip_adapter_image_embeds=[feats], output_type="latent", num_inference_steps=a.steps, guidance_scale=7.0, init_noise=noise,
           previewer_scheduler=LCMSingleStepScheduler.from_config(pipe.scheduler.config),
           callback_on_step_end=lambda p_, i, t, kw: (gpu_x.append(kw["latents"].float().cpu().clone()), {})[1]).images.float().cpu()
assert len(gpu_x) == a.steps and torch.equal(gpu_x[-1], got)
print(f"HIP chain of {a.steps} steps done ({time.time() - t00:.0f} s), finite={bool(torch.isfinite(got).all())}", flush=True)
torch.set_num_threads(cores())
P = {k: v.float().cpu() for k, v in sd.items()}; PA = {k: v.float().cpu() for k, v in sda.items()}; L = {k: v.float().cpu() for k, v in lora.items()}
L["scaling"] = alpha / cfg.lora_rank
del sd, sda, lora, pipe; torch.cuda.empty_cache()
state = torch.load(a.ckpt, weights_only=True) if os.path.isfile(a.ckpt) else {"x": None, "next": 0, "psnr": []}
k0 = int(state["next"]); k1 = min(k0 + a.max_steps, a.steps)
print(f"oracle: steps [{k0}, {k1}) on {cores()} cores", flush=True)


def on_step(i, x):
    p = psnr(gpu_x[i], x)
    state["x"], state["next"] = x.clone(), i + 1
    state["psnr"] = list(state["psnr"]) + [p]
    torch.save(state, os.path.join(a.out, "ckpt.pt"))          # survives the call: copy it to --ckpt for the next sitting
    print(f"step {i + 1:2d}/{a.steps}: latents after the step, HIP vs oracle {p:.1f} dB   [{time.time() - t00:.0f} s]", flush=True)


with torch.no_grad():
    OP.denoise(P, PA, L, cfg, lq, pe, pooled, feats, negative_prompt_embeds=npe, negative_pooled=npooled, init_noise=noise,
               num_inference_steps=a.steps, guidance_scale=7.0, sampler="ddim", resume=(state["x"], k0, k1), on_step=on_step)
stop.set()
if state["next"] >= a.steps:
    print(f"RESULT size={a.size} cfg=7.0 steps={a.steps}: final latents HIP vs CPU fp32 oracle {state['psnr'][-1]:.1f} dB; "
          f"per step min {min(state['psnr']):.1f} dB; curve {[round(v, 1) for v in state['psnr']]}", flush=True)
