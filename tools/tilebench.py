"""Tile / ring-depth sweep with COLD weights (cycling through ~600 MB of distinct matrices), per shape class of the loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")

def run(M, N, K, tile, iters=120):
    nw = max(2, int(600e6 / (N * K * 2)))
    a = torch.randn(M, K, device=dev).half()
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).half() for _ in range(nw)]
    out = torch.empty(M, N, device=dev, dtype=torch.half)
    try:
        for i in range(6): ops.gemm(a, ws[i % nw], out, tile=tile)
    except Exception as e:
        return float("nan")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): ops.gemm(a, ws[i % nw], out, tile=tile)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

TILES = [int(t) for t in os.environ.get("TILES", "0,22,32,42,23,33,25,35,45,21,24,34").split(",")]
SHAPES = [(2048, 1280, 1280), (1280, 2048, 1280), (2048, 2560, 1280), (2048, 1280, 5120), (4096, 1280, 1280), (4096, 2560, 1280),
          (8192, 640, 640), (8192, 1280, 640), (640, 8192, 640), (8192, 640, 2560), (16384, 640, 640)]
print("shape".ljust(24) + "".join(f"{('t%d' % t):>8s}" for t in TILES))
for M, N, K in SHAPES:
    row = f"{M:6d}x{N:5d}x{K:5d}".ljust(24)
    for t in TILES:
        row += f"{run(M, N, K, t):8.1f}"
    print(row, flush=True)
