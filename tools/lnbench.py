import sys, os
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
for M, C in [(2048, 1280), (8192, 640)]:
    x = torch.randn(M, C, device=dev).half()
    a = torch.randn(M, C, device=dev).half()
    wo = (torch.randn(C, C, device=dev) * C ** -0.5).half()
    wqk = (torch.randn(2 * C, C, device=dev) * C ** -0.5).half()
    w1 = (torch.randn(8 * C, C, device=dev) * C ** -0.5).half()
    s2 = torch.randn(2 * C, device=dev); s8 = torch.randn(8 * C, device=dev)
    b2 = torch.randn(2 * C, device=dev).half(); b8 = torch.randn(8 * C, device=dev).half()
    g = torch.ones(C, device=dev).half(); z = torch.zeros(C, device=dev).half()
    h = torch.empty(M, C, device=dev, dtype=torch.half); n = torch.empty(M, C, device=dev, dtype=torch.half)
    qk = torch.empty(M, 2 * C, device=dev, dtype=torch.half); f = torch.empty(M, 4 * C, device=dev, dtype=torch.half)
    P = ops.stat_partials(M, C)
    slab = torch.zeros(P * M * 2, device=dev)
    print(f"M={M} C={C} P={P}")
    print("  producer to_out plain      %.1f us" % timeit(lambda: ops.gemm(a, wo, h, bias=z, res=x)))
    print("  producer to_out + stat_out %.1f us" % timeit(lambda: ops.gemm(a, wo, h, bias=z, res=x, stat_out=slab)))
    print("  layernorm kernel           %.1f us" % timeit(lambda: ops.layernorm(h, n, g, z, 1e-5)))
    print("  qk plain                   %.1f us" % timeit(lambda: ops.gemm(n, wqk, qk)))
    print("  qk + ln fold               %.1f us" % timeit(lambda: ops.gemm(h, wqk, qk, bias=b2, ln=(slab, P, 0, 1e-5, s2, None))))
    print("  ff1 geglu plain            %.1f us" % timeit(lambda: ops.gemm(n, w1, f, bias=b8, epi=ops.EPI_GEGLU)))
    print("  ff1 geglu + ln fold        %.1f us" % timeit(lambda: ops.gemm(h, w1, f, bias=b8, epi=ops.EPI_GEGLU, ln=(slab, P, 0, 1e-5, s8, None))))
