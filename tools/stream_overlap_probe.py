"""Do two of OUR kernels from two streams overlap?  N launches of one half-chip GEMM (M = 1024: 128 workgroups of the 64 x 160
loader-wave tile) on one stream, then the same N on each of two streams (independent operands).  Perfect overlap: the two-stream
run takes as long as the one-stream run; none: twice as long."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")


def mk(M, N, K):
    return (torch.randn(M, K, device=dev).half(), (torch.randn(N, K, device=dev) * K ** -0.5).half(), torch.empty(M, N, device=dev, dtype=torch.half))


def run(sets, streams, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        for (a, w, o), s in zip(sets, streams):
            with torch.cuda.stream(s):
                ops.gemm(a, w, o)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for M, N, K in [(1024, 1280, 1280), (1024, 1280, 5120), (2048, 1280, 1280), (4096, 640, 640)]:
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    A, B = mk(M, N, K), mk(M, N, K)
    run([A, B], [s1, s2], 20)
    one = run([A], [s1], 200)
    two = run([A, B], [s1, s2], 200)
    same = run([A, B], [s1, s1], 200)
    print(f"M={M} N={N} K={K}: one stream {one:.1f} us per launch; two streams {two:.1f} us per pair; both on one stream {same:.1f} us per pair", flush=True)
