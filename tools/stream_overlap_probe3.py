"""Do two half-chip GEMMs overlap when each stream is confined to its own half of the CUs (hipExtStreamCreateWithCUMask)?"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(mask_words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(mask_words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


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


# 256 CUs = 8 words of 32 bits; variant A: lower / upper half of the bit range; variant B: even / odd bits
lo, hi = [0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4
ev, od = [0x55555555] * 8, [0xAAAAAAAA] * 8
for name, (m1, m2) in (("halves", (lo, hi)), ("even/odd", (ev, od))):
    s1, s2 = masked_stream(m1), masked_stream(m2)
    for M, N, K in [(1024, 1280, 1280), (1024, 1280, 5120)]:
        A, B = mk(M, N, K), mk(M, N, K)
        run([A, B], [s1, s2], 20)
        one = run([A], [s1], 200)
        two = run([A, B], [s1, s2], 200)
        print(f"CU masks {name}: M={M} N={N} K={K}: one masked stream {one:.1f} us per launch; two masked streams {two:.1f} us per pair", flush=True)
