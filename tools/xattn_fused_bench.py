"""attn2.to_q + cross-attention: the two launches against the fused one (IIR_EPI_XATTN), warm, on the step's shapes."""
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


for R, T, heads in [(2, 1024, 20), (2, 4096, 10), (2, 2048, 20), (2, 8192, 10)]:
    C = heads * 64
    M, K = R * T, C
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(C, K, generator=g) * K ** -0.5 * ops.attn_q_factor()).half().to(dev)
    segs = []
    for L in (77, 64):
        k = torch.randn(R * L, C, generator=g).half().to(dev)
        tp = (L + 7) // 8 * 8
        vt = torch.randn(C, R * tp, generator=g).half().to(dev)
        segs.append((k, L, vt, tp, L))
    q = torch.empty(M, C, dtype=torch.half, device=dev); o = torch.empty_like(q); o2 = torch.empty_like(q)
    t_g = timeit(lambda: ops.gemm(a, w, q))
    t_a = timeit(lambda: ops.attention(q, o, segs, R, heads, T, q_prescaled=True))
    t_both = timeit(lambda: (ops.gemm(a, w, q), ops.attention(q, o, segs, R, heads, T, q_prescaled=True)))
    t_f = timeit(lambda: ops.gemm(a, w, o2, epi=ops.EPI_XATTN, xattn=(segs, T)))
    print(f"R={R} T={T} heads={heads}: to_q {t_g:.1f} us + attention {t_a:.1f} us = pair {t_both:.1f} us; fused {t_f:.1f} us", flush=True)
