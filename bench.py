#!/usr/bin/env python3
"""Benchmark of the InstantIR denoising step on MI355X (BASELINE.json metric: denoising steps/sec).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): one 1024x1024 image
per GPU (latent 128x128), classifier-free guidance 7.0 (2 UNet rows), full SDXL-base UNet with TA-IP
adapters, previewer-LoRA UNet + LCM preview + Aggregator + UNet every step (preview_start = 0), DDIM
(eta 0) update, fp16 storage / fp32 accumulate, seeded synthetic weights of the exact shapes (no
checkpoints exist offline).  A "step" is one iteration of pipelines/sdxl_instantir.py:1497 for the
GPU's local image.  Multi-GPU: images shard across ranks, no per-step collective ("weak" scaling);
weights are broadcast once from rank 0 over RCCL before the timed region.

Timed region: K steps replayed from the captured hipGraph of the step, inputs resident in HBM,
bracketed by barrier + synchronize; max over ranks; value = N*K / time.
roofline: a separate, eager pass of 2 steps (queued behind a spin kernel so launches run back to back) where every
MFMA-kernel launch carries a HIP start/stop event pair stamped with the kernel's own begin / end timestamps on its stream
(hipExtLaunchKernelGGL inside the library); the kernel class with the largest total time is reported against the dense
fp16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).  Algorithmic FLOPs per launch = 2*M*N*K (GEMM /
implicit-GEMM conv) or 4*B*h*Tq*Tkv*64 (attention).
cpu_baseline: the CPU fp32 oracle (oracle/, kind "port") timed on the host cores for a bounded sample
(one UNet forward of one row at the same resolution), converted with SURVEY.md section 8d's FLOP model.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch

UNET_TFLOP_ROW = 6.833     # SURVEY.md section 8d, 1024^2
STEP_TFLOP = 39.74         # 2 * (2 * UNet + Aggregator) per image with CFG


def host_cores():
    """CPU share of this process: min(affinity, cgroup v2 quota).  The GPU boxes expose 256 logical
    CPUs but a 16-CPU quota; oversubscribing the quota stalls torch's thread pool."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def synth_encoder_weights(model_cls, cfg, dev, seed):
    """Random weights with the exact inventory (names, shapes) of a transformers architecture: the class is instantiated on the
    meta device (no allocation, no checkpoint -- none exist offline) and every tensor is drawn on the GPU."""
    with torch.device("meta"):
        m = model_cls(cfg)
    g = torch.Generator(device=dev).manual_seed(seed)
    sd = {}
    for k, v in m.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        if v.ndim == 1 and ("norm" in k or "layrnorm" in k) and k.endswith("weight"):
            t = 1.0 + 0.05 * torch.randn(v.shape, generator=g, device=dev)
        elif "lambda1" in k:
            t = 0.5 + 0.5 * torch.rand(v.shape, generator=g, device=dev)
        elif v.ndim == 1:
            t = 0.02 * torch.randn(v.shape, generator=g, device=dev)
        elif "embed" in k or "cls_token" in k:
            t = 0.05 * torch.randn(v.shape, generator=g, device=dev)
        else:
            t = torch.randn(v.shape, generator=g, device=dev) * (v.shape[-1] if v.ndim == 2 else v[0].numel()) ** -0.5
        sd[k] = t.half()
    return sd


def end_to_end_leg(pipe, cfg, hv, dev, px, lcm, seed):
    """One real `pipe(...)` call as infer.py makes it (/root/reference infer.py:211-225): a 1024 x 1024 PIXEL image in, pixels out,
    30 DDIM steps, cfg 7.0 -- DINOv2-L + zero-image features, both CLIP text encoders (prompt given as token ids: no tokenizer
    vocabulary exists offline), Resampler + per-image hoists (`prepare`), the loop, VAE encode and decode.  Returns seconds per
    image of the SECOND call, on a different image and prompt (the first call of a geometry pays one-time set-up: arena sizing, kernel
    attribute calls, the capture of the step graphs, which `InstantIRPipeline._loop_for` keeps)."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection, Dinov2Config, Dinov2Model
    from instantir_amd.encoders import HipCLIPText, HipDinov2
    pipe.vae = hv
    pipe.image_encoder = HipDinov2(synth_encoder_weights(Dinov2Model, Dinov2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                                                                   patch_size=14, image_size=518, mlp_ratio=4), dev, seed + 10), dev)
    c1 = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                        max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=768)
    c2 = CLIPTextConfig(vocab_size=49408, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                        max_position_embeddings=77, hidden_act="gelu", projection_dim=1280)
    pipe.text_encoder = HipCLIPText(synth_encoder_weights(CLIPTextModelWithProjection, c1, dev, seed + 11), dev, hidden_act="quick_gelu", eos_token_id=49407)
    pipe.text_encoder_2 = HipCLIPText(synth_encoder_weights(CLIPTextModelWithProjection, c2, dev, seed + 12), dev, hidden_act="gelu", eos_token_id=49407)
    rs = np.random.RandomState(seed)
    img = Image.fromarray(rs.randint(0, 256, (px, px, 3), dtype=np.uint8))
    ids = torch.full((1, 77), 49407, dtype=torch.long)
    ids[0, 0] = 49406
    ids[0, 1:12] = torch.from_numpy(rs.randint(1000, 40000, 11))
    nids = torch.full((1, 77), 49407, dtype=torch.long)
    nids[0, 0] = 49406
    kw = dict(image=img, prompt_ids=ids, prompt_ids_2=ids, negative_prompt_ids=nids, negative_prompt_ids_2=nids,
              num_inference_steps=30, guidance_scale=7.0, previewer_scheduler=lcm, output_type="pt",
              generator=torch.Generator(device=dev).manual_seed(seed))
    out = pipe(**kw).images
    torch.cuda.synchronize()
    ids2 = ids.clone()
    ids2[0, 1:12] = torch.from_numpy(rs.randint(1000, 40000, 11))
    kw.update(image=Image.fromarray(rs.randint(0, 256, (px, px, 3), dtype=np.uint8)), prompt_ids=ids2, prompt_ids_2=ids2)     # the next image of a batch job
    t0 = time.perf_counter()
    out = pipe(**kw).images
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(out).all().item()) and tuple(out.shape) == (1, 3, px, px)
    pipe.vae = pipe.image_encoder = pipe.text_encoder = pipe.text_encoder_2 = None
    return dt, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024, help="image side (pixels)")
    ap.add_argument("--tiny", action="store_true", help="tiny geometry (debug only, not a valid bench)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end pipe(...) call (images/s measured)")
    ap.add_argument("--no-inkernel-prefetch", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="single stream (no encoder/previewer overlap)")
    ap.add_argument("--rows", type=int, default=2, choices=[1, 2],
                    help="1: the step WITHOUT classifier-free guidance (one UNet row per image) -- a timing experiment on per-launch "
                         "overheads, not the BASELINE workload; the JSON line says so")
    ap.add_argument("--config", type=int, default=1, choices=[1, 4],
                    help="BASELINE.json configs index: 1 = the metric's own workload (default); 4 = LCM single-step previewer "
                         "restoration, 4 images per GPU, fp8-E4M3 weights on the linear layers (a parity-test configuration, "
                         "printed as its own line)")
    ap.add_argument("--fp16", action="store_true", help="--config 4 with fp16 linears (A/B of the fp8 path)")
    args = ap.parse_args()

    # ---- launch contract: `--gpus N` must mean N ranks.  Under torch.distributed.run WORLD_SIZE says how many there are;
    # a bare `python bench.py --gpus N` (N > 1) starts the N ranks itself as a CHILD process group (torchrun on
    # 127.0.0.1) -- decided here, before anything touches the GPU (a process that has initialised HIP must never exec).
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if env_world == 0 and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"--gpus {args.gpus} without a launcher: starting {args.gpus} ranks via torch.distributed.run")
        raise SystemExit(subprocess.run(cmd).returncode)
    if env_world and env_world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: the launcher and the flag disagree")

    from instantir_amd import lib, ops, parallel, weights as W
    from instantir_amd.config import UNetConfig
    from instantir_amd.pipeline import InstantIRPipeline, _DenoiseLoop
    from instantir_amd.schedulers import DDIMScheduler, LCMSingleStepScheduler

    rank, world, local, dev = parallel.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the denoising path has no CPU fallback")
    lib.load()
    if args.config == 4:
        return config4(args, rank, world, local, dev)
    cfg = UNetConfig.tiny() if args.tiny else UNetConfig.sdxl()
    Hl = args.size // 8
    B, guidance, n_steps_sched = 1, 7.0, 30

    ranks_info = parallel.describe_ranks(rank, world, local, dev)      # every rank's device + backend, checked distinct under RCCL
    # ---- frozen weights: ONLY rank 0 generates them; every other rank allocates empty tensors of the same inventory
    # and receives the values over RCCL/xGMI.  The broadcast is timed on its own, outside the
    # timed region of the metric.
    t0 = time.time()
    seed = 1234
    inv = [(W.unet_specs(cfg), seed), (W.aggregator_specs(cfg), seed + 1), (W.lora_specs(cfg), seed + 2)]
    if rank == 0:
        log(f"rank 0/{world} on {dev}: generating weights")
        sd, sda, lora = (W.synth_state_dict(sp, sd_seed, device=dev) for sp, sd_seed in inv)
    else:
        sd, sda, lora = ({n: torch.empty(shape, dtype=torch.float16, device=dev) for n, shape, _ in sp} for sp, _ in inv)
    bcast = None
    if world > 1:
        torch.cuda.synchronize()
        parallel.barrier()
        tb = time.perf_counter()
        nbytes = 0
        for d in (sd, sda, lora):
            parallel.broadcast_state_dict(d, 0)
            nbytes += sum(v.numel() * v.element_size() for v in d.values())
        torch.cuda.synchronize()
        parallel.barrier()
        tb = parallel.max_over_ranks(time.perf_counter() - tb, dev)
        bcast = {"bytes": nbytes, "seconds": round(tb, 3), "GB_per_s": round(nbytes / tb / 1e9, 1)}
        log(f"weights broadcast from rank 0: {nbytes / 1e9:.2f} GB in {tb:.2f} s")
    pipe = InstantIRPipeline(cfg, sd, scheduler=DDIMScheduler(), device=dev)
    pipe.aggregator.load_state_dict(sda)
    pipe.prepare_previewers(lora, lora_alpha=cfg.lora_rank // 8)
    pipe.use_graphs = not args.no_graph
    pipe.overlap_streams = not args.no_overlap
    pipe._build()
    for net in (pipe._unet, pipe._unet_prev, pipe._agg):
        net.inkernel_prefetch = not args.no_inkernel_prefetch
    log(f"engines built ({time.time() - t0:.1f} s)")
    n_params = sum(v.numel() for v in sd.values()) + sum(v.numel() for v in sda.values())
    cpu_sd = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_sd = {k: v.float().cpu() for k, v in sd.items()}
    del sd, sda, lora
    pipe._unet_sd = pipe._agg_sd = pipe._lora = None
    torch.cuda.empty_cache()

    # ---- synthetic inputs (seed 42 + rank: every rank restores a different image) -----------------
    g = torch.Generator().manual_seed(42 + rank)
    lq = torch.randn(B, 4, Hl, Hl, generator=g) * 0.8
    pe = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g)
    npe = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g)
    pooled, npooled = torch.randn(B, cfg.pooled_dim, generator=g), torch.randn(B, cfg.pooled_dim, generator=g)
    img = torch.randn(2, B, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)
    rep = args.rows
    if rep == 1:
        ctx, pl, img, guidance = pe, pooled, img[1:], 1.0
    else:
        ctx, pl = torch.cat([npe, pe]), torch.cat([npooled, pooled])
    px = Hl * 8
    time_ids = torch.tensor([[px, px, 0, 0, px, px]], dtype=torch.float32).repeat(rep * B, 1)
    st = pipe._unet.prepare(ctx, pl, time_ids, pipe._unet.resampler(img), Hl, Hl)
    st_prev = pipe._unet_prev.prepare(ctx, pl, time_ids, pipe._unet_prev.resampler(img), Hl, Hl)
    st_agg = pipe._agg.prepare(pl, time_ids, Hl, Hl)
    lcm = LCMSingleStepScheduler.from_config(pipe.scheduler.config)
    pipe.scheduler.set_timesteps(n_steps_sched)
    ts = [int(t) for t in pipe.scheduler.timesteps]
    lqd = lq.to(dev)
    loop = _DenoiseLoop(pipe, B, rep, Hl, Hl, st, st_prev, st_agg, lqd, None, lcm)
    x = pipe.scheduler.add_noise(lqd, torch.randn(lq.shape, generator=g).to(dev), torch.tensor([ts[0]] * B)).contiguous()
    scale_rows = torch.ones(rep * B)
    setup_s = time.time() - t0
    log(f"prepared ({setup_s:.1f} s); warmup")

    def run(k, start=0):
        for i in range(k):
            loop.step("preview", ts[(start + i) % len(ts)], x, scale_rows, guidance, 0.0, None, None)

    run(args.warmup)
    torch.cuda.synchronize()
    log("warmup done; timing")
    parallel.barrier()
    t1 = time.perf_counter()
    run(args.steps, args.warmup)
    torch.cuda.synchronize()
    parallel.barrier()
    dt_local = time.perf_counter() - t1
    dt = parallel.max_over_ranks(dt_local, dev)
    rank_ms = [round(v / args.steps * 1e3, 3) for v in parallel.gather_floats(dt_local, dev)]      # a straggler shows here
    finite = bool(torch.isfinite(x).all().item())
    log(f"timed: {dt / args.steps * 1e3:.2f} ms/step")

    # ---- VAE legs (once per image, outside the step metric): full SDXL VAE geometry ----------------------
    vae_ms = e2e = None
    if rank == 0 and not args.tiny and not args.no_vae:
        from instantir_amd.config import VAEConfig
        from instantir_amd.vae import HipVAE
        vc = VAEConfig.sdxl()
        hv = HipVAE(vc, W.synth_state_dict(W.vae_decoder_specs(vc) + W.vae_encoder_specs(vc), seed + 3, device=dev), dev)
        px_img = torch.rand(1, 3, px, px, generator=g) * 2 - 1
        eps_l = torch.randn(1, 4, Hl, Hl, generator=g)
        for _ in range(2):
            hv.decode(x / vc.scaling_factor); hv.encode(px_img, eps_l)
        torch.cuda.synchronize()
        tv = time.perf_counter(); hv.decode(x / vc.scaling_factor); torch.cuda.synchronize(); dec_ms = (time.perf_counter() - tv) * 1e3
        tv = time.perf_counter(); hv.encode(px_img, eps_l); torch.cuda.synchronize(); enc_ms = (time.perf_counter() - tv) * 1e3
        vae_ms = {"decode_ms": round(dec_ms, 2), "encode_ms": round(enc_ms, 2)}
        log(f"vae decode {dec_ms:.1f} ms, encode {enc_ms:.1f} ms")
        if not args.no_e2e and args.size == 1024:
            try:
                e2e_s, e2e_ok = end_to_end_leg(pipe, cfg, hv, dev, px, lcm, seed + 20)
                e2e = {"seconds_per_image": round(e2e_s, 4), "images_per_s": round(1.0 / e2e_s, 4), "finite_and_shaped": e2e_ok,
                       "what": "pipe(image=PIL 1024x1024, prompt_ids=..., 30 steps, cfg 7.0, output_type='pt'): DINOv2-L + CLIP-L/bigG + "
                               "Resampler + prepare + loop + VAE encode/decode, second call on the pipeline (a different image and prompt; the "
                               "step graphs captured by the first call of a geometry are kept, as for every later image of a batch job; the negative prompt is the "
                               "job's fixed one, so its two CLIP passes come from the per-token-id cache, the new positive prompt is encoded)"}
                log(f"end-to-end pipe(...) call: {e2e_s:.3f} s per image")
            except Exception as ex:      # (transformers / PIL missing on the box: say so instead of failing the bench line)
                e2e = {"seconds_per_image": None, "images_per_s": None, "error": f"{type(ex).__name__}: {ex}"[:300]}
                log(f"end-to-end leg failed: {ex}")
        del hv
        torch.cuda.empty_cache()

    # ---- roofline leg: eager, every MFMA launch bracketed by HIP events --------------------------
    roof = None
    if rank == 0 and not args.no_roofline:
        pipe.use_graphs = False
        run(1)
        ops.PROFILER = ops.LaunchProfiler()
        # park the GPU behind a spin kernel while the host enqueues both steps: the event pairs then bracket kernels
        # that run back to back (as in the graph replay) instead of measuring the host's launch latency
        torch.cuda._sleep(int(1.2e9))
        run(2)
        ops_prof = ops.PROFILER
        summ = ops_prof.summary()
        ops.PROFILER = None
        pipe.use_graphs = not args.no_graph
        name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = d["ms"] / d["launches"]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch of that class from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
        # separately over this same command, corrected as MI355X_MICROARCH.md prescribes: profiles/*_pmc_traffic.json)
        traffic = None
        if not args.tiny and args.size == 1024:
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            latest = lambda n: (n.split("_")[0], "final" in n, n)          # newest round, its final pass before its mid-round one
            pm = sorted((p_ for p_ in os.listdir(pdir) if p_.endswith("_pmc_traffic.json")), key=latest) if os.path.isdir(pdir) else []
            if pm:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pm[-1])) as f:
                    traffic = json.load(f)["classes"].get(name, {}).get("hbm_bytes_per_launch")
        mfma_busy = None
        if not args.tiny and args.size == 1024:
            pb = sorted((p_ for p_ in os.listdir(pdir) if p_.endswith("_pmc_mfma_busy.json")), key=latest) if os.path.isdir(pdir) else []
            if pb:
                with open(os.path.join(pdir, pb[-1])) as f:
                    mfma_busy = json.load(f)["classes"].get(name, {}).get("mfma_busy_frac")
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(ach / 2500.0, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC, profiles/)",
                "mfma_busy_pmc": mfma_busy,      # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), separate PMC pass (profiles/)
                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]), "launches_per_step": d["launches"] // 2,
                "avg_launch_us": round(avg_ms * 1e3, 2), "timing": "hipExtLaunchKernelGGL start/stop events (kernel begin/end timestamps)",
                "classes": {k: {"launches_per_step": v["launches"] // 2, "ms_per_step": round(v["ms"] / 2, 3),
                                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in sorted(summ.items())}}

    # ---- CPU baseline leg (rank 0 only): oracle UNet forward, 1 row ------------------------------
    cpu = None
    if rank == 0 and cpu_sd is not None:
        from oracle import nets
        ncores = host_cores()
        log(f"cpu baseline on {ncores} cores")
        torch.set_num_threads(ncores)
        with torch.no_grad():
            ip = nets.image_projection(cpu_sd, [img[1:].float()], cfg.resampler)[0]
            tc = time.perf_counter()
            nets.unet_forward(cpu_sd, cfg, lq, ts[0], pe, pooled, time_ids[:1], ip)
            cpu_t = time.perf_counter() - tc
        frac = UNET_TFLOP_ROW / STEP_TFLOP if not args.tiny else 1.0
        cpu = {"value": round(frac / cpu_t, 5), "unit": "steps/s", "cores": ncores, "kind": "port",
               "sample": f"oracle UNet forward, 1 row, {args.size}x{args.size}, fp32, {cpu_t:.1f} s = {frac:.3f} of a CFG step "
                         f"by SURVEY 8d FLOPs ({UNET_TFLOP_ROW}/{STEP_TFLOP} TFLOP)"}

    if rank == 0:
        out = {
            "metric": "denoising steps/sec at 1024x1024 (InstantIR step: previewer UNet + LCM + Aggregator + UNet, CFG)",
            "value": round(world * args.steps / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {args.size}x{args.size}, batch 1/GPU, cfg 7.0 (2 rows), 30-step DDIM "
                                   f"timetable, preview_start 0, full SDXL+TA-IP UNet x2 + Aggregator per step"
                                   + (" [TINY DEBUG GEOMETRY]" if args.tiny else "")
                                   + (" [EXPERIMENT --rows 1: NO classifier-free guidance, not the BASELINE workload]" if args.rows == 1 else ""),
                       "images_per_gpu": B, "latent": [Hl, Hl], "weight_elements": n_params, "graph": not args.no_graph,
                       # SURVEY 8d's model at the metric's 1024^2; the other resolutions are parity-test configurations whose step was
                       # counted once by hand: no figure is claimed for sizes without one
                       "algorithmic_tflop_per_step": {512: 9.16, 1024: STEP_TFLOP, 2048: 225.85}.get(args.size),
                       "algorithmic_tflops_per_gpu": (round({512: 9.16, 1024: STEP_TFLOP, 2048: 225.85}[args.size] * args.steps / dt, 1)
                                                      if args.size in (512, 1024, 2048) and not args.tiny else None),
                       "images_per_s_30step": round(world / (30 * dt / args.steps + (((vae_ms or {}).get("decode_ms", 0) + (vae_ms or {}).get("encode_ms", 0)) * 1e-3)), 4),
                       "images_per_s_measured": (e2e or {}).get("images_per_s"), "end_to_end": e2e,
                       "ms_per_step_by_rank": rank_ms,
                       "vae": vae_ms, "finite": finite,
                       "setup_s": round(setup_s, 1), "world": world, "ranks": ranks_info, "weight_broadcast": bcast},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def config4(args, rank, world, local, dev):
    """BASELINE configs[4]: LCM single-step scheduler, previewer-LoRA path, 32 images of 1024x1024 over 8 GPUs = 4 images per
    GPU, guidance 1.0 (no CFG doubling), fp8 weights.  One "step" = the whole single-step restoration of the GPU's 4 images:
    noise to t = 999, ONE LoRA-UNet pass (R = 4 rows; transformer linears on fp8-E4M3 weights), LCM x0 step.  VAE encode /
    decode are once-per-image stages outside the step and are timed separately (bf16)."""
    from instantir_amd import ops, parallel, weights as W
    from instantir_amd.config import UNetConfig, VAEConfig
    from instantir_amd.engine import CPAD, F16, HipUNet
    from instantir_amd.schedulers import LCMSingleStepScheduler
    cfg = UNetConfig.tiny() if args.tiny else UNetConfig.sdxl()
    Hl, B, t999 = args.size // 8, 4, 999
    seed = 1234
    ranks_info = parallel.describe_ranks(rank, world, local, dev)
    inv = [(W.unet_specs(cfg), seed), (W.lora_specs(cfg), seed + 2)]
    if rank == 0:
        sd, lora = (W.synth_state_dict(sp, sd_seed, device=dev) for sp, sd_seed in inv)
    else:
        sd, lora = ({n: torch.empty(shape, dtype=torch.float16, device=dev) for n, shape, _ in sp} for sp, _ in inv)
    if world > 1:
        for d in (sd, lora):
            parallel.broadcast_state_dict(d, 0)
    net = HipUNet(cfg, sd, dev, lora=lora, lora_scaling=(cfg.lora_rank // 8) / cfg.lora_rank, fp8_linear=not args.fp16)
    del sd, lora
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(42 + rank)
    lq = (torch.randn(B, 4, Hl, Hl, generator=g) * 0.8).to(dev)
    pe = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g)
    pooled = torch.randn(B, cfg.pooled_dim, generator=g)
    img = torch.randn(1, B, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)
    px = Hl * 8
    time_ids = torch.tensor([[px, px, 0, 0, px, px]], dtype=torch.float32).repeat(B, 1)
    st = net.prepare(pe, pooled, time_ids, net.resampler(img), Hl, Hl)
    lcm = LCMSingleStepScheduler()
    noise = torch.randn(B, 4, Hl, Hl, generator=g).to(dev)
    coef = torch.tensor(lcm.preview_coefficients(t999), dtype=torch.float32).to(dev)
    t_dev = torch.full((B, 1), float(t999), dtype=torch.float32, device=dev)
    lat16 = torch.zeros(B * Hl * Hl, CPAD, dtype=F16, device=dev)
    out16 = torch.zeros(B * Hl * Hl, CPAD, dtype=F16, device=dev)
    out = torch.empty(B, 4, Hl, Hl, dtype=torch.float32, device=dev)

    def step():
        x = lcm.add_noise(lq, noise, torch.tensor([t999] * B)).contiguous()
        ops.pack_latent(x, lat16)
        eps = net.forward(lat16, t_dev, st)
        ops.lcm_step(eps, B, 1, coef, x, out16, out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t1, dev)
    finite = bool(torch.isfinite(out).all().item())
    vae_ms = roof = None
    if rank == 0 and not args.tiny and not args.no_vae:
        from instantir_amd.vae import HipVAE
        vc = VAEConfig.sdxl()
        hv = HipVAE(vc, W.synth_state_dict(W.vae_decoder_specs(vc) + W.vae_encoder_specs(vc), seed + 3, device=dev), dev)
        pimg = torch.rand(1, 3, px, px, generator=g) * 2 - 1
        eps_l = torch.randn(1, 4, Hl, Hl, generator=g)
        for _ in range(2):
            hv.decode(out[:1] / vc.scaling_factor); hv.encode(pimg, eps_l)
        torch.cuda.synchronize()
        tv = time.perf_counter(); hv.decode(out[:1] / vc.scaling_factor); torch.cuda.synchronize(); dec_ms = (time.perf_counter() - tv) * 1e3
        tv = time.perf_counter(); hv.encode(pimg, eps_l); torch.cuda.synchronize(); enc_ms = (time.perf_counter() - tv) * 1e3
        vae_ms = {"decode_ms_per_image": round(dec_ms, 2), "encode_ms_per_image": round(enc_ms, 2)}
        del hv
    if rank == 0 and not args.no_roofline:
        step()
        ops.PROFILER = ops.LaunchProfiler()
        torch.cuda._sleep(int(1.2e9))
        step(); step()
        summ = ops.PROFILER.summary()
        ops.PROFILER = None
        name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
                "traffic": None, "peak_note": "v_mfma_f32_16x16x32_fp8_fp8 (non-scaled) issues at the fp16/bf16 rate (MI355X_MICROARCH.md); the "
                                              "5 PFLOP/s fp8 figure belongs to the MX block-scaled K = 128 form, which this path does not use",
                "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2), "launches_per_step": d["launches"] // 2,
                "classes": {k: {"launches_per_step": v["launches"] // 2, "ms_per_step": round(v["ms"] / 2, 3),
                                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in sorted(summ.items())}}
    if rank == 0:
        ms = dt / args.steps * 1e3
        per_img_s = ms * 1e-3 / B + ((vae_ms or {}).get("decode_ms_per_image", 0) + (vae_ms or {}).get("encode_ms_per_image", 0)) * 1e-3
        print(json.dumps({
            "metric": "restoration steps/sec, LCM single-step previewer path at 1024x1024 (BASELINE configs[4]; not the headline metric)",
            "value": round(world * args.steps / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16 (fp32 accumulate)" if args.fp16 else
                      ("fp8-e4m3 weights AND activations on the transformer linears (activations stored as fp8 by the producing launch; fp32 accumulate); fp16 elsewhere"
                       if net.fp8_act else
                       "fp8-e4m3 weights, fp16 activations converted to fp8 in the GEMM's registers, on the transformer linears (fp32 accumulate); fp16 elsewhere")),
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: LCM single step (t = 999, guidance 1.0), previewer LoRA merged, {B} images/GPU at "
                                   f"{args.size}x{args.size}, one UNet pass of {B} rows + LCM step per step" + (" [TINY DEBUG GEOMETRY]" if args.tiny else ""),
                       "images_per_gpu": B, "images_per_s_incl_vae": round(world / per_img_s, 3), "vae": vae_ms, "finite": finite,
                       "algorithmic_tflop_per_step": round(6.833 * B, 2), "algorithmic_tflops_per_gpu": round(6.833 * B / (ms * 1e-3), 1),
                       "world": world, "ranks": ranks_info},
            "roofline": roof, "cpu_baseline": None}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
