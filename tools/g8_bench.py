"""gemm8.hip (256 x BN tile, 8 waves, two-tile-deep LDS-DMA pipeline) against the 4-wave kernel: bit-for-bit equality on the
epilogue forms it covers (same fp32 accumulation order, same epilogue arithmetic) and the time per launch on the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
from instantir_amd.packing import pair_rows
dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(M, N, K, geglu, fold, res, act=ops.ACT_NONE, tile_new=91, tile_old=24, time_it=True):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    b = torch.randn(N, generator=g).half().to(dev)
    kw = dict(epi=ops.EPI_GEGLU if geglu else ops.EPI_PLAIN, act=act)
    bias = b
    if geglu:
        w = pair_rows(w[:N // 2], w[N // 2:]); bias = pair_rows(b[:N // 2], b[N // 2:])
    if fold:
        gam, bet = (torch.randn(K, generator=g) * 0.2 + 1).half().to(dev), (torch.randn(K, generator=g) * 0.1).half().to(dev)
        f = ops.LnFold(w.float(), gam, bet, bias=bias.float())
        parts = ops.ln_parts(M, K, K)
        st = torch.zeros(parts, M, 2, device=dev)
        h = torch.empty(M, K, device=dev, dtype=torch.half)
        wo = (torch.randn(K, K, generator=g) * K ** -0.5).half().to(dev)
        ops.gemm(a, wo, h, ln_out=st)                 # producer leaves the partials of h
        a, w, bias = h, f.w, f.bias
        kw["ln_in"] = (st, f.colsum, 1e-5)
    n_out = N // 2 if geglu else N
    r = torch.randn(M, n_out, generator=g).half().to(dev) if res else None
    if r is not None: kw["res"] = r
    o_new = torch.zeros(M, n_out, device=dev, dtype=torch.half)
    o_old = torch.zeros(M, n_out, device=dev, dtype=torch.half)
    ops.gemm(a, w, o_old, bias=bias, tile=tile_old, **kw)
    ops.gemm(a, w, o_new, bias=bias, tile=tile_new, **kw)
    torch.cuda.synchronize()
    same = torch.equal(o_new, o_old)
    md = (o_new.float() - o_old.float()).abs().max().item()
    ref = (a.float() @ w.float().t())
    msg = f"M={M:5d} N={N:5d} K={K:4d} geglu={int(geglu)} fold={int(fold)} res={int(res)} act={act}: bit-equal={same} max|d|={md:.3g} finite={bool(torch.isfinite(o_new).all())}"
    if time_it:
        fl = 2 * M * N * K
        t_old = timeit(lambda: ops.gemm(a, w, o_old, bias=bias, tile=tile_old, **kw))
        t_auto = timeit(lambda: ops.gemm(a, w, o_old, bias=bias, tile=0, **kw))
        t_new = timeit(lambda: ops.gemm(a, w, o_new, bias=bias, tile=tile_new, **kw))
        msg += f" | t{tile_old} {t_old:6.1f} us ({fl / t_old / 1e6:5.0f} TF)  auto {t_auto:6.1f} us  t{tile_new} {t_new:6.1f} us ({fl / t_new / 1e6:5.0f} TF)"
    print(msg, flush=True)
    return same


def case_qkv(M, C, fold=True, time_it=True):
    """fused q|k|v projection: columns >= 2C leave the GEMM transposed (out_t)."""
    g = torch.Generator(device="cpu").manual_seed(M + C)
    N, K = 3 * C, C
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    b = torch.randn(N, generator=g).half().to(dev)
    kw = {}
    if fold:
        gam, bet = (torch.randn(K, generator=g) * 0.2 + 1).half().to(dev), (torch.randn(K, generator=g) * 0.1).half().to(dev)
        f = ops.LnFold(w.float(), gam, bet, bias=b.float())
        st = torch.zeros(ops.ln_parts(M, K, K), M, 2, device=dev)
        h = torch.empty(M, K, device=dev, dtype=torch.half)
        ops.gemm(a, (torch.randn(K, K, generator=g) * K ** -0.5).half().to(dev), h, ln_out=st)
        a, w, b = h, f.w, f.bias
        kw["ln_in"] = (st, f.colsum, 1e-5)
    outs = []
    for tile in (21, 91):
        qk = torch.zeros(M, 2 * C, device=dev, dtype=torch.half); vt = torch.zeros(C, M, device=dev, dtype=torch.half)
        ops.gemm(a, w, qk, bias=b, tile=tile, out_t=(vt, 2 * C), **kw)
        outs.append((qk, vt))
    torch.cuda.synchronize()
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    msg = f"qkv M={M} C={C} fold={int(fold)}: bit-equal={same}"
    if time_it:
        qk, vt = outs[0]
        ts = [timeit(lambda: ops.gemm(a, w, qk, bias=b, tile=t_, out_t=(vt, 2 * C), **kw)) for t_ in (21, 0, 91)]
        msg += f" | t21 {ts[0]:6.1f} us  auto {ts[1]:6.1f} us  t91 {ts[2]:6.1f} us ({2 * M * N * K / ts[2] / 1e6:5.0f} TF)"
    print(msg, flush=True)
    return same


if __name__ == "__main__":
    ok = True
    if "qkv" in sys.argv:
        ok &= case_qkv(512, 320, time_it=False)
        ok &= case_qkv(8192, 640)
        ok &= case_qkv(16384, 640)
        ok &= case_qkv(2048, 1280)
        print("ALL BIT-EQUAL" if ok else "MISMATCH")
        sys.exit(0)
    if "quick" in sys.argv:
        ok &= case(512, 640, 256, True, False, False, time_it=False)
        ok &= case(2048, 10240, 1280, True, True, False)
        ok &= case(4096, 10240, 1280, True, True, False)
        ok &= case(8192, 5120, 640, True, True, False)
        ok &= case(8192, 7680, 8192, False, False, False, tile_new=91)
        print("ALL BIT-EQUAL" if ok else "MISMATCH")
        sys.exit(0)
    ok &= case(256, 320, 128, False, False, False, time_it=False)
    ok &= case(512, 640, 192, False, False, True, time_it=False)
    ok &= case(512, 640, 256, True, False, False, time_it=False)
    ok &= case(256, 640, 320, False, False, True, act=ops.ACT_SILU, time_it=False)
    ok &= case(512, 512, 256, False, False, True, tile_new=92, time_it=False)
    ok &= case(2048, 10240, 1280, True, True, False)
    ok &= case(2048, 10240, 1280, True, False, False)
    ok &= case(2048, 10240, 1280, False, False, True)
    ok &= case(4096, 10240, 1280, True, True, False)
    ok &= case(8192, 5120, 640, True, True, False)
    ok &= case(16384, 5120, 640, True, True, False)
    ok &= case(8192, 5120, 640, False, False, True)
    ok &= case(8192, 8192, 8192, False, False, False, tile_new=92)
    ok &= case(8192, 8192, 8192, False, False, False, tile_new=91) if 8192 % 320 == 0 else True
    ok &= case(8192, 7680, 8192, False, False, False, tile_new=91)
    print("ALL BIT-EQUAL" if ok else "MISMATCH")
