"""Fixed cost vs per-K-tile cost: time of one GEMM as K grows (warm operands)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
from kbench import timeit
dev = torch.device("cuda:0")
for (M, N, tile) in [(2048, 10240, 24), (2048, 10240, 21), (4096, 10240, 24), (2048, 1280, 22), (2048, 1280, 45), (2048, 2560, 25), (8192, 5120, 24), (16384, 5120, 24), (8192, 640, 25)]:
    row = f"M={M} N={N} t{tile}: "
    for K in (64, 128, 320, 640, 1280, 2560, 5120):
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        out = torch.empty(M, N, device=dev, dtype=torch.half)
        t = timeit(lambda: ops.gemm(a, w, out, tile=tile), iters=50)
        row += f" K{K}:{t*1e6:6.1f}"
    print(row, flush=True)
