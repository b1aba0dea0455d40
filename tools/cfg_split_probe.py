"""Experiment: what would running the two CFG rows as two independent launch streams buy?  Runs the bench step with
`--rep` rows per image (2 = CFG-doubled batch as shipped, 1 = one row) and `--B` images; with `--peer FILE` two such
processes rendezvous through the file system after warm-up so that their timed regions coincide on the one GPU.

    python tools/cfg_split_probe.py --rep 2                       # the shipped step
    python tools/cfg_split_probe.py --rep 1 --tag a --peer b & python tools/cfg_split_probe.py --rep 1 --tag b --peer a
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--rep", type=int, default=2); ap.add_argument("--B", type=int, default=1); ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--warmup", type=int, default=4); ap.add_argument("--tag", default="x"); ap.add_argument("--peer", default=None)
ap.add_argument("--size", type=int, default=1024)
a = ap.parse_args()
from instantir_amd import lib, weights as W
from instantir_amd.config import UNetConfig
from instantir_amd.pipeline import InstantIRPipeline, _DenoiseLoop
from instantir_amd.schedulers import DDIMScheduler, LCMSingleStepScheduler
lib.load()
dev = torch.device("cuda:0"); cfg = UNetConfig.sdxl(); Hl = a.size // 8; B, rep = a.B, a.rep
sd, sda, lora = (W.synth_state_dict(sp, s, device=dev) for sp, s in ((W.unet_specs(cfg), 1234), (W.aggregator_specs(cfg), 1235), (W.lora_specs(cfg), 1236)))
pipe = InstantIRPipeline(cfg, sd, scheduler=DDIMScheduler(), device=dev)
pipe.aggregator.load_state_dict(sda); pipe.prepare_previewers(lora, lora_alpha=8); pipe._build()
del sd, sda, lora; pipe._unet_sd = pipe._agg_sd = pipe._lora = None; torch.cuda.empty_cache()
g = torch.Generator().manual_seed(42)
lq = torch.randn(B, 4, Hl, Hl, generator=g) * 0.8
R = B * rep
ctx, pl = torch.randn(R, cfg.text_len, cfg.cross_attention_dim, generator=g), torch.randn(R, cfg.pooled_dim, generator=g)
img = torch.randn(rep, B, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)
px = Hl * 8
tid = torch.tensor([[px, px, 0, 0, px, px]], dtype=torch.float32).repeat(R, 1)
st = pipe._unet.prepare(ctx, pl, tid, pipe._unet.resampler(img), Hl, Hl)
st_prev = pipe._unet_prev.prepare(ctx, pl, tid, pipe._unet_prev.resampler(img), Hl, Hl)
st_agg = pipe._agg.prepare(pl, tid, Hl, Hl)
lcm = LCMSingleStepScheduler.from_config(pipe.scheduler.config)
pipe.scheduler.set_timesteps(30); ts = [int(t) for t in pipe.scheduler.timesteps]
lqd = lq.to(dev)
loop = _DenoiseLoop(pipe, B, rep, Hl, Hl, st, st_prev, st_agg, lqd, None, lcm)
x = pipe.scheduler.add_noise(lqd, torch.randn(lq.shape, generator=g).to(dev), torch.tensor([ts[0]] * B)).contiguous()
rows = torch.ones(R)
def run(k):
    for i in range(k): loop.step("preview", ts[i % len(ts)], x, rows, 7.0 if rep == 2 else 1.0, 0.0, None, None)
run(a.warmup); torch.cuda.synchronize()
if a.peer:
    os.makedirs("gpurun_out", exist_ok=True)
    open(f"gpurun_out/.probe_{a.tag}", "w").close()
    t0 = time.time()
    while not os.path.exists(f"gpurun_out/.probe_{a.peer}"):
        time.sleep(0.01)
        if time.time() - t0 > 300: raise SystemExit("peer never arrived")
t1 = time.perf_counter(); run(a.steps); torch.cuda.synchronize(); dt = (time.perf_counter() - t1) / a.steps
print(f"PROBE tag={a.tag} B={B} rep={rep} peer={a.peer}: {dt * 1e3:.2f} ms/step, finite={bool(torch.isfinite(x).all())}", flush=True)
