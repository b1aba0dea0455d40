"""HBM-bound kernels of the step on the step's own shapes (1024^2, R = 2): achieved GB/s against the 6.3 TB/s the chip
sustains on a float4 copy (MI355X_MICROARCH.md) and the 8 TB/s spec.  Algorithmic bytes = every operand read once, every
output written once.  Burst timing (30 back-to-back launches, torch events); launches_per_step from the engine's op order."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
h = lambda *s: torch.randn(*s, device=dev).half()
REP = 30

def timeit(fn):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP        # us

rows = []
# GroupNorm(+SiLU): (R, HW, C) as used by the resnets / transformers of the UNet (R = 2) and the Aggregator (2H x W maps)
for name, R, HW, C, n in [("gn L0 320ch 128^2", 2, 128 * 128, 320, 8), ("gn L0 cat 960ch 128^2", 2, 128 * 128, 960, 2), ("gn L1 640ch 64^2", 2, 64 * 64, 640, 12),
                          ("gn L1 cat 1920ch 64^2", 2, 64 * 64, 1920, 2), ("gn L2 1280ch 32^2", 2, 32 * 32, 1280, 40), ("gn L2 cat 2560ch 32^2", 2, 32 * 32, 2560, 4),
                          ("gn agg L0 320ch 256x128", 2, 256 * 128, 320, 5), ("gn agg L2 1280ch 64x32", 2, 64 * 32, 1280, 10)]:
    x, y, g, b = h(R * HW, C), h(R * HW, C), h(C), h(C)
    ws = ops.gn_workspace(dev, R)
    us = timeit(lambda: ops.groupnorm(x, y, R, HW, g, b, 1e-5, True, 32, ws))
    alg = R * HW * C * 2 * 2                      # one read + one write (the kernel reads x twice: statistics, apply)
    rows.append((name, "gn_stats+finalize+apply", us, alg, 3 * R * HW * C * 2, n))
for name, M, C, n in [("ln L1 8192x640", 8192, 640, 60), ("ln L2 2048x1280", 2048, 1280, 360), ("ln agg L1 16384x640", 16384, 640, 8), ("ln agg L2 4096x1280", 4096, 1280, 60)]:
    x, y, g, b = h(M, C), h(M, C), h(C), h(C)
    us = timeit(lambda: ops.layernorm(x, y, g, b, 1e-5))
    rows.append((name, "ln_kernel", us, M * C * 4, M * C * 4, n))
for name, M, C, n in [("copy_add L2 2048x1280", 2048, 1280, 12), ("copy_add L1 8192x640", 8192, 640, 12), ("copy_add L0 32768x320", 32768, 320, 12)]:
    src, add, dst = h(M, C), h(M, C), h(M, 2 * C)
    sc = torch.ones(2, device=dev)
    us = timeit(lambda: ops.copy_add(src, dst, 0, add=add, add_scale=sc, rows_per_scale=M // 2))
    rows.append((name, "copy_add", us, M * C * 6, M * C * 6, n))
out = []
print(f"{'case':28s} {'kernel':26s} {'us':>8s} {'alg GB/s':>9s} {'moved GB/s':>10s} {'frac of 6.3 TB/s (moved)':>24s}")
for name, k, us, alg, moved, n in rows:
    print(f"{name:28s} {k:26s} {us:8.1f} {alg / us / 1e3:9.0f} {moved / us / 1e3:10.0f} {moved / us / 1e3 / 6300:24.2f}")
    out.append({"case": name, "kernel": k, "us": round(us, 2), "algorithmic_GBps": round(alg / us / 1e3), "moved_GBps": round(moved / us / 1e3),
                "frac_of_6300": round(moved / us / 1e3 / 6300, 3), "approx_launches_per_step": n})
print(json.dumps({"hbm_bound_kernels": out}))
