"""In-situ-like GEMM chain: cold weights from one arena, each launch optionally prefetching the next launch's weights,
A = previous launch's output (when N == K).  us per launch by tile / ring depth, prefetch off | on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")

def run(M, N, K, tile, pf, iters=150):
    nw = max(3, int(600e6 / (N * K * 2)))
    arena = (torch.randn(nw * N * K, device=dev) * K ** -0.5).half()
    ws = [arena[i * N * K:(i + 1) * N * K].view(N, K) for i in range(nw)]
    bufs = [torch.randn(M, K, device=dev).half(), torch.empty(M, N, device=dev, dtype=torch.half)]
    chain = N == K
    def go(n):
        for i in range(n):
            a = bufs[i & 1] if chain else bufs[0]
            o = bufs[(i + 1) & 1] if chain else bufs[1]
            nxt = ws[(i + 1) % nw]
            ops.gemm(a, ws[i % nw], o, tile=tile, prefetch=(nxt.data_ptr(), N * K * 2) if pf else None)
    go(6); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(iters); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

TILES = [int(t) for t in os.environ.get("TILES", "22,32,25,35,45,23").split(",")]
print("shape".ljust(22) + "".join(f"{('t%d' % t):>14s}" for t in TILES))
for M, N, K in [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (4096, 1280, 1280), (8192, 640, 640)]:
    row = f"{M:6d}x{N:5d}x{K:5d}".ljust(22)
    for t in TILES:
        row += f"{run(M, N, K, t, False):7.1f}|{run(M, N, K, t, True):6.1f}"
    print(row, flush=True)
