"""Attention microbenchmark: iir_attention_d64_f16 on the step's shapes, random data, warm; us per launch and TFLOP/s.
IIR_ATTN_V selects the kernel generation (read once by the library): run once per value on the same box to compare.
Also checks every shape against F.scaled_dot_product_attention in fp32 (max abs error / output range)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from instantir_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).half().to(dev)
SHAPES = [("unet L1 self", 2, 10, 4096, [4096]), ("unet L2 self", 2, 20, 1024, [1024]), ("agg L1 self", 2, 10, 8192, [8192]),
          ("agg L2 self", 2, 20, 2048, [2048]), ("unet L1 cross", 2, 10, 4096, [77, 64]), ("unet L2 cross", 2, 20, 1024, [77, 64]),
          ("2048^2 agg L1", 2, 10, 32768, [32768])]
REP = int(os.environ.get("REP", "30"))
print("IIR_ATTN_V =", os.environ.get("IIR_ATTN_V", "(default)"))
for name, B, h, T, kvs in SHAPES:
    if T > 8192 and os.environ.get("BIG", "0") != "1":
        continue
    C = h * 64
    q = rnd(B * T, C)
    segs, ref = [], 0
    for Tk in kvs:
        k = rnd(B * Tk, C)
        pad = (Tk + 7) // 8 * 8
        v = rnd(B * Tk, C)
        vt = torch.zeros(C, B * pad, dtype=torch.half, device=dev)
        for b in range(B):
            vt[:, b * pad:b * pad + Tk] = v[b * Tk:(b + 1) * Tk].T
        segs.append((k, Tk, vt, pad, Tk))
        if T * Tk <= 4096 * 4096:
            q4 = q.float().reshape(B, T, h, 64).transpose(1, 2); k4 = k.float().reshape(B, Tk, h, 64).transpose(1, 2)
            v4 = v.float().reshape(B, Tk, h, 64).transpose(1, 2)
            ref = ref + F.scaled_dot_product_attention(q4, k4, v4).transpose(1, 2).reshape(B * T, C)
        else:
            ref = None
    o = torch.empty(B * T, C, dtype=torch.half, device=dev)
    ops.attention(q, o, segs, B, h, T)
    torch.cuda.synchronize()
    err = "n/a" if ref is None else f"{((o.float() - ref).abs().max() / ref.abs().max()).item():.2e}"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        ops.attention(q, o, segs, B, h, T)
    e0.record()
    for _ in range(REP):
        ops.attention(q, o, segs, B, h, T)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REP
    fl = 4.0 * B * h * T * 64 * sum(kvs)
    print(f"{name:16s} B={B} h={h} T={T} kv={kvs}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  rel.err {err}", flush=True)
