"""Text + IP cross-attention launch (Tkv = 77 + 64) on the step's shapes: time per launch and error against fp32 SDPA.
Run twice (IIR_ATTN_PRE=0 / default) to compare the ring form with the pre-staged form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for B, heads, T in ((2, 20, 1024), (2, 10, 4096), (2, 20, 2048)):
    C = heads * 64
    g = torch.Generator().manual_seed(T)
    q = torch.randn(B * T, C, generator=g).half().to(dev)
    segs, ref = [], torch.zeros(B, heads, T, 64, device=dev)
    q4 = q.float().view(B, T, heads, 64).transpose(1, 2)
    for L in (77, 64):
        pad = (L + 7) // 8 * 8
        k = torch.randn(B * L, C, generator=g).half().to(dev)
        v = torch.randn(B * L, C, generator=g).half().to(dev)
        vt = torch.zeros(C, B * pad, dtype=torch.half, device=dev)
        for r in range(B):
            vt[:, r * pad:r * pad + L] = v[r * L:(r + 1) * L].t()
        segs.append((k, L, vt, pad, L))
        k4 = k.float().view(B, L, heads, 64).transpose(1, 2); v4 = v.float().view(B, L, heads, 64).transpose(1, 2)
        ref += torch.nn.functional.scaled_dot_product_attention(q4, k4, v4)
    o = torch.empty(B * T, C, dtype=torch.half, device=dev)
    ops.attention(q, o, segs, B, heads, T)
    got = o.float().view(B, T, heads, 64).transpose(1, 2)
    err = (got - ref).abs().max().item()
    t = timeit(lambda: ops.attention(q, o, segs, B, heads, T))
    print(f"cross-attn B={B} h={heads} T={T} Tkv=77+64: {t:6.1f} us   max|err| vs fp32 SDPA {err:.2e}", flush=True)
