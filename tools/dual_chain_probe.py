"""Timing experiment (not a bench line): would two independent one-row chains (the cond and the uncond row of classifier-free
guidance, each through its own engines, arenas and hipGraph, on two streams) finish sooner than the one two-row chain?
Builds two pipelines from the same synthetic weights and times: (a) one two-row step, (b) one one-row step, (c) two one-row steps
enqueued on two streams.  The CFG combine of (c) is not done here -- this only prices the concurrency."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd.config import UNetConfig
from instantir_amd import weights as W
from instantir_amd.pipeline import InstantIRPipeline, _DenoiseLoop
from instantir_amd.schedulers import DDIMScheduler, LCMSingleStepScheduler

dev = torch.device("cuda:0")
cfg = UNetConfig.sdxl()
seed, Hl, B = 1234, 128, 1
sd = W.synth_state_dict(W.unet_specs(cfg), seed, device=dev)
sda = W.synth_state_dict(W.aggregator_specs(cfg), seed + 1, device=dev)
lora = W.synth_state_dict(W.lora_specs(cfg), seed + 2, device=dev)


def build():
    pipe = InstantIRPipeline(cfg, sd, scheduler=DDIMScheduler(), device=dev)
    pipe.aggregator.load_state_dict(sda)
    pipe.prepare_previewers(lora, lora_alpha=cfg.lora_rank // 8)
    pipe.use_graphs = True
    pipe._build()
    return pipe


def make_loop(pipe, rep, g):
    lq = torch.randn(B, 4, Hl, Hl, generator=g) * 0.8
    pe = torch.randn(rep * B, cfg.text_len, cfg.cross_attention_dim, generator=g)
    pooled = torch.randn(rep * B, cfg.pooled_dim, generator=g)
    img = torch.randn(rep, B, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)
    px = Hl * 8
    time_ids = torch.tensor([[px, px, 0, 0, px, px]], dtype=torch.float32).repeat(rep * B, 1)
    st = pipe._unet.prepare(pe, pooled, time_ids, pipe._unet.resampler(img), Hl, Hl)
    st_prev = pipe._unet_prev.prepare(pe, pooled, time_ids, pipe._unet_prev.resampler(img), Hl, Hl)
    st_agg = pipe._agg.prepare(pooled, time_ids, Hl, Hl)
    lcm = LCMSingleStepScheduler.from_config(pipe.scheduler.config)
    pipe.scheduler.set_timesteps(30)
    loop = _DenoiseLoop(pipe, B, rep, Hl, Hl, st, st_prev, st_agg, lq.to(dev), None, lcm)
    x = torch.randn(B, 4, Hl, Hl, generator=g).to(dev)
    return loop, x, torch.ones(rep * B), [int(t) for t in pipe.scheduler.timesteps]


def timed(fn, n=20, warm=3):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): fn(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


g = torch.Generator().manual_seed(7)
p1, p2 = build(), build()
for rep in (2, 1):
    l1, x1, s1, ts = make_loop(p1, rep, g)
    ms = timed(lambda i: l1.step("preview", ts[i % 30], x1, s1, 7.0 if rep == 2 else 1.0, 0.0, None, None))
    print(f"one chain, {rep} row(s): {ms:.2f} ms per step", flush=True)
l1, x1, s1, ts = make_loop(p1, 1, g)
l2, x2, s2, _ = make_loop(p2, 1, g)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both(i):
    with torch.cuda.stream(sa):
        l1.step("preview", ts[i % 30], x1, s1, 1.0, 0.0, None, None)
    with torch.cuda.stream(sb):
        l2.step("preview", ts[i % 30], x2, s2, 1.0, 0.0, None, None)


for _ in range(2):
    print(f"two one-row chains on two streams: {timed(both):.2f} ms per step pair", flush=True)

# (d) the same two chains without hipGraphs (eager launches from one host thread)
p1.use_graphs = p2.use_graphs = False
for _ in range(2):
    print(f"two one-row chains, two streams, eager launches: {timed(both, n=10):.2f} ms per step pair", flush=True)
p1.use_graphs = False
ms = timed(lambda i: l1.step("preview", ts[i % 30], x1, s1, 1.0, 0.0, None, None), n=10)
print(f"one one-row chain, eager launches: {ms:.2f} ms per step", flush=True)

# (e) both chains inside ONE captured graph (fork / join on a side stream at capture time) -- `onegraph` on the command line;
# the capture aborted the process on ROCm 7.2 when tried (cross-stream capture of two engines), left here for the record
if "onegraph" not in sys.argv:
    sys.exit(0)
key = ("preview", False, False, False)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    main = torch.cuda.current_stream()
    f, j = torch.cuda.Event(), torch.cuda.Event()
    f.record(main)
    sb.wait_event(f)
    with torch.cuda.stream(sb):
        l2._launch(*key)
        j.record(sb)
    l1._launch(*key)
    main.wait_event(j)
for _ in range(2):
    print(f"two one-row chains as branches of one graph: {timed(lambda i: gr.replay()):.2f} ms per step pair", flush=True)
