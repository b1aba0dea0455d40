"""where the per-image overhead of a pipe(...) call goes: phases timed with device synchronisation (monkeypatched wrappers)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from instantir_amd import pipeline as PL
T = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[label] = T.get(label, 0.0) + (time.perf_counter() - t)
        return r
    setattr(obj, name, g)
orig_leg = B.end_to_end_leg
def leg(pipe, cfg, hv, dev, px, lcm, seed):
    wrap(pipe, "encode_prompt", "encode_prompt")
    wrap(pipe, "prepare_ip_adapter_image_embeds", "image_encoder")
    wrap(hv, "encode_to_latent", "vae_encode")
    wrap(hv, "decode_latent", "vae_decode")
    wrap(pipe._unet, "prepare", "prepare_unet"); wrap(pipe._unet_prev, "prepare", "prepare_prev"); wrap(pipe._agg, "prepare", "prepare_agg")
    wrap(pipe._unet, "resampler", "resampler_unet"); wrap(pipe._unet_prev, "resampler", "resampler_prev")
    wrap(pipe, "_loop_for", "loop_for(adopt)")
    wrap(PL._DenoiseLoop, "step", "steps") if False else None
    r = orig_leg(pipe, cfg, hv, dev, px, lcm, seed)
    return r
B.end_to_end_leg = leg
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--steps", "5"]
import runpy
try:
    B.main() if hasattr(B, "main") else runpy.run_module("bench", run_name="__main__")
finally:
    print("PHASES (sum over the two calls, s):", {k: round(v, 4) for k, v in T.items()})
