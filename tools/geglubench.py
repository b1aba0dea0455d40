import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, C in [(2048, 1280), (4096, 1280), (8192, 640), (16384, 640)]:
    n = torch.randn(M, C, device=dev).half()
    ws = [(torch.randn(8 * C, C, device=dev) * C ** -0.5).half() for _ in range(12)]
    b = torch.randn(8 * C, device=dev).half()
    f = torch.empty(M, 4 * C, device=dev, dtype=torch.half)
    row = f"ff1 GEGLU M={M} C={C}:"
    for tile in (0, 1, 4, 5, 6):
        i = [0]
        def run():
            i[0] += 1
            ops.gemm(n, ws[i[0] % 12], f, bias=b, epi=ops.EPI_GEGLU, tile=tile)
        t = timeit(run)
        row += f"  t{tile} {2*M*8*C*C/t/1e6:6.0f} TF"
    print(row, flush=True)
