"""LayerNorm folding: what the producer (`ln_out`) and the consumer (`ln_in`) sides cost per launch on the step's shapes,
against the same launches without it and against the LayerNorm kernel they replace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
from instantir_amd.packing import pair_rows
dev = torch.device("cuda:0")

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for M, C in ((2048, 1280), (8192, 640)):
    h = torch.randn(M, C, device=dev).half(); n = torch.empty_like(h)
    gam, bet = torch.ones(C, device=dev).half(), torch.zeros(C, device=dev).half()
    parts = ops.ln_parts(M, C, C)
    st = torch.zeros(parts, M, 2, device=dev)
    a = torch.randn(M, C, device=dev).half(); wo = (torch.randn(C, C, device=dev) * C ** -0.5).half()
    ops.gemm(a, wo, h, res=h, ln_out=st)
    print(f"M={M} C={C} parts={parts}")
    print(f"  layernorm kernel                 {timeit(lambda: ops.layernorm(h, n, gam, bet, 1e-5)):7.1f} us")
    print(f"  producer {M}x{C}x{C} +res        {timeit(lambda: ops.gemm(a, wo, n, res=h)):7.1f} us   with ln_out {timeit(lambda: ops.gemm(a, wo, n, res=h, ln_out=st)):7.1f} us")
    for name, N, geglu, out_t in (("qkv", 3 * C, False, True), ("to_q", C, False, False), ("ff1", 8 * C, True, False)):
        w = (torch.randn(N, C, device=dev) * C ** -0.5).half(); b = torch.randn(N, device=dev).half()
        f = ops.LnFold(w, gam, bet, bias=b, pair=pair_rows if geglu else None)
        out = torch.empty(M, (N // 2 if geglu else (2 * C if out_t else N)), device=dev, dtype=torch.half)
        vt = torch.empty(C, M, device=dev, dtype=torch.half)
        kw = dict(epi=ops.EPI_GEGLU if geglu else ops.EPI_PLAIN)
        if out_t: kw["out_t"] = (vt, 2 * C)
        t0 = timeit(lambda: ops.gemm(n, f.w, out, bias=f.bias, **kw))
        t1 = timeit(lambda: ops.gemm(h, f.w, out, bias=f.bias, ln_in=(st, f.colsum, 1e-5), **kw))
        print(f"  consumer {name:5s} {M}x{N}x{C}     {t0:7.1f} us   with ln_in  {t1:7.1f} us")
