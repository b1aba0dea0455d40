"""Every kernel must give bit-identical results when a second stream keeps the chip busy: the step runs the main UNet's encoder
beside the previewer UNet + Aggregator, and a kernel that is only reproducible when it has the chip to itself makes the whole
pipeline irreproducible.  The full screen is `tools/racecheck_concurrent.py` (60 cases); this is the
driver-run subset: one case per kernel family, 8 launches each, beside GEMM + attention + 64-row-tile conv noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_kernels_are_reproducible_beside_a_busy_stream():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).half().to(dev)
    side = torch.cuda.Stream()
    nx, nw, no = rnd(4096, 1280), rnd(2560, 1280, scale=0.03), torch.empty(4096, 2560, dtype=torch.half, device=dev)
    nq, nvt, nao = rnd(2 * 2048, 2 * 640), rnd(640, 2 * 2048), torch.empty(2 * 2048, 640, dtype=torch.half, device=dev)
    ncx, ncw, nco = rnd(2, 64, 64, 640), rnd(640, 3, 3, 640, scale=0.02), torch.empty(2 * 64 * 64, 640, dtype=torch.half, device=dev)

    def noise():
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.gemm(nx, nw, no)
                ops.attention(nq[:, :640], nao, [(nq[:, 640:], 2048, nvt, 2048, 2048)], 2, 10, 2048)
                ops.conv2d(ncx, ncw, nco)                     # resolves to the 64x160 tile: the neighbour that exposed section 5.8

    def screen(name, launch, shapes, dtype=torch.half):
        first = None
        for it in range(8):
            outs = [torch.zeros(s, dtype=dtype, device=dev) for s in shapes]
            noise()
            launch(*outs)
            torch.cuda.synchronize()
            cat = torch.cat([o.flatten().float() for o in outs])
            if first is None:
                first = cat
            else:
                assert torch.equal(cat, first), f"{name}: launch {it} differs from launch 0 ({int((cat != first).sum())} elements)"

    M, C = 2048, 1280
    x, w, b, res = rnd(M, C), rnd(C, C, scale=C ** -0.5), rnd(C), rnd(M, C)
    for tile in (0, 25, 24):
        screen(f"gemm tile {tile}", lambda o: ops.gemm(x, w, o, bias=b, res=res, tile=tile), [(M, C)])
    w1, b1 = rnd(8 * C, C, scale=C ** -0.5), rnd(8 * C)
    screen("gemm GEGLU", lambda o: ops.gemm(x, pair_rows(w1[:4 * C], w1[4 * C:]), o, bias=pair_rows(b1[:4 * C], b1[4 * C:]), epi=ops.EPI_GEGLU), [(M, 4 * C)])
    parts = ops.ln_parts(M, C, C)
    st = torch.zeros(parts, M, 2, device=dev)
    h = torch.empty(M, C, dtype=torch.half, device=dev)
    ops.gemm(x, w, h, res=res, ln_out=st)
    w3 = rnd(3 * C, C, scale=C ** -0.5)
    f = ops.LnFold(w3, rnd(C) + 1, rnd(C))
    screen("gemm with ln_in (q|k|v, V transposed)", lambda qk, vt: ops.gemm(h, f.w, qk, bias=f.bias, out_t=(vt, 2 * C), ln_in=(st, f.colsum, 1e-5)), [(M, 2 * C), (C, M)])
    cx, cw, cb = rnd(2, 32, 32, 1280), rnd(1280, 3, 3, 1280, scale=0.01), rnd(1280)
    screen("conv3x3 level 2 (loader-wave tile)", lambda o: ops.conv2d(cx, cw, o, bias=cb), [(2 * 32 * 32, 1280)])
    c2x, c2w = rnd(2, 64, 64, 640), rnd(640, 3, 3, 640, scale=0.02)
    screen("conv3x3 stride 2", lambda o: ops.conv2d(c2x, c2w, o, stride=2), [(2 * 32 * 32, 640)])
    for (B, heads, T, kvs) in ((2, 20, 1024, [1024]), (2, 10, 4096, [77, 64])):
        Cc = heads * 64
        q = rnd(B * T, Cc)
        segs = [(rnd(B * Tk, Cc), Tk, rnd(Cc, B * ((Tk + 7) // 8 * 8)), (Tk + 7) // 8 * 8, Tk) for Tk in kvs]
        screen(f"attention T={T} kv={kvs}", lambda o: ops.attention(q, o, segs, B, heads, T), [(B * T, Cc)])
    gx, gg, gb = rnd(2 * 4096, 640), rnd(640) + 1, rnd(640)
    ws = ops.gn_workspace(dev, 2, 32)
    screen("groupnorm", lambda o: ops.groupnorm(gx, o, 2, 4096, gg, gb, 1e-5, True, 32, ws), [(2 * 4096, 640)])
    lx, lg, lb = rnd(2048, 1280), rnd(1280) + 1, rnd(1280)
    screen("layernorm", lambda o: ops.layernorm(lx, o, lg, lb, 1e-5), [(2048, 1280)])
