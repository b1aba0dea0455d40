"""Kernel micro-benchmarks on SDXL shapes (GPU box).  Prints TFLOP/s per shape and tile."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
from instantir_amd.packing import conv_weight_nhwc

dev = torch.device("cuda:0")
TILES = tuple(int(t) for t in os.environ.get('TILES', '0,22,25,35').split(','))

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def gemm_bench():
    shapes = [tuple(int(v) for v in sh.split('x')) for sh in os.environ['SHAPES'].split(',')] if 'SHAPES' in os.environ else [(2048, 2560, 1280), (4096, 1280, 1280), (8192, 1280, 640), (8192, 640, 640), (8192, 1920, 640), (8192, 5120, 640), (8192, 640, 2560),
              (2048, 1280, 1280), (2048, 3840, 1280), (2048, 10240, 1280), (2048, 1280, 5120),
              (16384, 5120, 640), (4096, 10240, 1280), (4096, 4096, 4096), (8192, 8192, 8192)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        out = torch.empty(M, N, device=dev, dtype=torch.half)
        row = f"gemm M={M:6d} N={N:6d} K={K:5d}:"
        geglu = os.environ.get("GEGLU", "0") == "1"
        outg = torch.empty(M, N // 2, device=dev, dtype=torch.half)
        for tile in TILES:
            t = timeit(lambda: ops.gemm(a, w, outg, tile=tile, epi=ops.EPI_GEGLU) if geglu else ops.gemm(a, w, out, tile=tile))
            row += f"  t{tile} {2*M*N*K/t/1e12:6.0f}"
        if os.environ.get("W8", "0") == "1":          # the same launches on fp8-E4M3 weights
            w8 = ops.Fp8Weight(*ops.quantize_fp8_rows(w))
            row += "  | fp8-w:"
            for tile in TILES:
                t = timeit(lambda: ops.gemm(a, w8, outg, tile=tile, epi=ops.EPI_GEGLU) if geglu else ops.gemm(a, w8, out, tile=tile))
                row += f"  t{tile} {2*M*N*K/t/1e12:6.0f}"
        if os.environ.get("F8", "0") == "1" and K % 128 == 0:       # both operands fp8 (128 K values per K tile)
            w8 = ops.Fp8Weight(*ops.quantize_fp8_rows(w))
            a8, sa = ops.quantize_fp8_tensor(a)
            row += "  | fp8 x fp8:"
            for tile in TILES:
                try:
                    t = timeit(lambda: ops.gemm_fp8(a8, w8, outg, a_scale=sa, tile=tile, epi=ops.EPI_GEGLU) if geglu else ops.gemm_fp8(a8, w8, out, a_scale=sa, tile=tile))
                    row += f"  t{tile} {2*M*N*K/t/1e12:6.0f}"
                except Exception:
                    row += f"  t{tile}    n/a"
        t = timeit(lambda: torch.matmul(a, w.T, out=out))
        row += f"  | hipblaslt {2*M*N*K/t/1e12:7.1f} TF"
        print(row, flush=True)

def conv_bench():
    for R, H, Cin, Cout in [(2, 128, 320, 320), (2, 64, 640, 640), (2, 32, 1280, 1280), (2, 32, 2560, 1280), (2, 64, 1920, 640), (2,128,960,320)]:
        x = torch.randn(R, H, H, Cin, device=dev).half(); w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9*Cin) ** -0.5).half()
        out = torch.empty(R * H * H, Cout, device=dev, dtype=torch.half)
        row = f"conv R={R} H={H:4d} Cin={Cin:5d} Cout={Cout:5d}:"
        fl = 2 * R * H * H * Cout * 9 * Cin
        for tile in TILES:
            t = timeit(lambda: ops.conv2d(x, w, out, tile=tile))
            row += f"  t{tile} {fl/t/1e12:6.0f}"
        print(row, flush=True)

def attn_bench():
    for B, heads, T in [(2, 10, 4096), (2, 20, 1024), (2, 10, 8192), (2, 20, 2048)]:
        C = heads * 64
        qkv = torch.randn(B * T, 2 * C, device=dev).half()
        vt = torch.randn(C, B * T, device=dev).half()
        o = torch.empty(B * T, C, device=dev, dtype=torch.half)
        t = timeit(lambda: ops.attention(qkv[:, :C], o, [(qkv[:, C:], T, vt, T, T)], B, heads, T))
        fl = 4 * B * heads * T * T * 64
        q4 = qkv[:, :C].reshape(B, T, heads, 64).transpose(1, 2); k4 = qkv[:, C:].reshape(B, T, heads, 64).transpose(1, 2)
        t2 = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, k4))
        print(f"attn B={B} h={heads} T={T}: {fl/t/1e12:7.1f} TF ({t*1e6:7.1f} us) | torch sdpa {fl/t2/1e12:7.1f} TF", flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "attn"]
    if "gemm" in which: gemm_bench()
    if "conv" in which: conv_bench()
    if "attn" in which: attn_bench()
