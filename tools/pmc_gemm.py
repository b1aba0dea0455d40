"""Launch a fixed list of GEMM variants a few times each (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
CASES = [(2048, 1280, 1280, 21), (2048, 1280, 1280, 23), (2048, 10240, 1280, 21), (8192, 8192, 8192, 21), (8192, 640, 640, 21)]
for M, N, K, tile in CASES:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    out = torch.empty(M, N, device=dev, dtype=torch.half)
    for _ in range(3):
        ops.gemm(a, w, out, tile=tile)
    torch.cuda.synchronize()
