"""Split-K vs single-slice on the long-K small-MN shapes (warm and cold weights)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
def run(M, N, K, ws, cold, tile=0, iters=100):
    nw = max(2, int(600e6 / (N * K * 2))) if cold else 1
    a = torch.randn(M, K, device=dev).half()
    wts = [(torch.randn(N, K, device=dev) * K ** -0.5).half() for _ in range(nw)]
    out = torch.empty(M, N, device=dev, dtype=torch.half)
    for i in range(5): ops.gemm(a, wts[i % nw], out, splitk_ws=ws, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): ops.gemm(a, wts[i % nw], out, splitk_ws=ws, tile=tile)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in [(2048, 1280, 5120), (2048, 1280, 11520), (2048, 1280, 2560), (1024, 1280, 5120)]:
    ws = ops.splitk_workspace(M, N, dev)
    print(f"{M}x{N}x{K}: warm single {run(M,N,K,None,False):6.1f} split {run(M,N,K,ws,False):6.1f} t34 {run(M,N,K,None,False,34):6.1f} | cold single {run(M,N,K,None,True):6.1f} split {run(M,N,K,ws,True):6.1f} us", flush=True)
