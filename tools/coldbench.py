"""Cold-vs-warm weights: same GEMM cycling through > 256 MB of distinct weight matrices vs one matrix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
def run(M, N, K, nw, tile=0, iters=200):
    a = torch.randn(M, K, device=dev).half()
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).half() for _ in range(nw)]
    out = torch.empty(M, N, device=dev, dtype=torch.half)
    for i in range(10): ops.gemm(a, ws[i % nw], out, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): ops.gemm(a, ws[i % nw], out, tile=tile)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (2048, 10240, 1280), (8192, 640, 640), (8192, 5120, 640)]:
    nw_cold = max(2, int(600e6 / (N * K * 2)))
    warm, cold = run(M, N, K, 1), run(M, N, K, nw_cold)
    print(f"M={M} N={N} K={K}: warm {warm:7.1f} us ({2*M*N*K/warm/1e6:6.0f} TF)   cold({nw_cold} mats) {cold:7.1f} us ({2*M*N*K/cold/1e6:6.0f} TF)", flush=True)
