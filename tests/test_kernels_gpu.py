"""GPU: every HIP kernel, through the C ABI, against a plain PyTorch fp32 reference of the same op."""
import math

import numpy as np

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from instantir_amd import lib
    lib.load()          # fails loudly if the HIP library is missing
    return torch.device("cuda:0")


def _rand(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).half()


def _close(got, want, rtol=2e-3, atol=2e-3, what=""):
    got = got.float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{want.numel()} off, max err {err.max().item():.4g} (ref max {want.abs().max().item():.4g})"


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 31, 36, 55, 65, 75])      # 55 / 65 / 75: 64x160 with loader waves, 3 / 4 / 2 stages
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 132, 64), (2, 320, 320), (1000, 640, 1280)])
def test_gemm_plain(dev, tile, M, N, K):
    from instantir_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a, w = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5)
    bias, res = _rand(g, N), _rand(g, M, N)
    rb = _rand(g, (M + 3) // 4, N)
    want = F.silu(a.float() @ w.float().T + bias.float() + rb.float().repeat_interleave(4, 0)[:M]) + res.float()
    out = torch.zeros(M, N + 8, dtype=torch.half, device=dev)      # wider buffer: exercises ldc
    ops.gemm(a.to(dev), w.to(dev), out[:, :N], bias=bias.to(dev), rowbias=rb.to(dev), rows_per_rb=4, res=res.to(dev),
             act=ops.ACT_SILU, tile=tile)
    torch.cuda.synchronize()
    _close(out[:, :N], want, what="gemm")
    assert out[:, N:].abs().max().item() == 0          # nothing written outside the view


@pytest.mark.parametrize("M,N,K,geglu", [(2048, 1280, 5120, False), (300, 320, 2560, False), (512, 640, 2688, True), (32, 256, 4608, False)])
def test_gemm_split_k(dev, M, N, K, geglu):
    """Two-slice split-K (tile = 0 with a workspace on long-K, few-tile problems): the last of a tile's two workgroups
    reduces.  Same result as the unsplit launch up to fp32 summation order, identical across repeated launches
    (the arrival order must not matter), counters left zeroed."""
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    g = torch.Generator().manual_seed(M + K)
    a, w, bias, res = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N), _rand(g, M, N // 2 if geglu else N)
    ws = ops.splitk_workspace(M, N, dev)
    assert ws is not None and ws.numel() == 4096 + ((M + 127) // 128) * ((N + 159) // 160) * 2 * 128 * 160 * 4
    ad = a.to(dev)
    if geglu:
        h = a.float() @ w.float().T + bias.float()
        want = h[:, :N // 2] * F.gelu(h[:, N // 2:])
        wd, bd = pair_rows(w[:N // 2], w[N // 2:]).to(dev), pair_rows(bias[:N // 2], bias[N // 2:]).to(dev)
        kw = dict(bias=bd, epi=ops.EPI_GEGLU)
    else:
        want = a.float() @ w.float().T + bias.float() + res.float()
        wd = w.to(dev)
        kw = dict(bias=bias.to(dev), res=res.to(dev))
    No = want.shape[1]
    plain = torch.empty(M, No, dtype=torch.half, device=dev)
    ops.gemm(ad, wd, plain, **kw)
    outs = []
    for _ in range(3):
        o = torch.zeros(M, No, dtype=torch.half, device=dev)
        ops.gemm(ad, wd, o, splitk_ws=ws, **kw)
        outs.append(o)
    torch.cuda.synchronize()
    _close(outs[0], want, what="split-K gemm")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert (outs[0].float() - plain.float()).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())
    assert ws[:4096].view(torch.int32).abs().max().item() == 0


def test_conv_split_k(dev):
    from instantir_amd import ops
    from instantir_amd.packing import conv_weight_nhwc
    g = torch.Generator().manual_seed(5)
    R, H, W, Cin, Cout = 2, 16, 16, 320, 320                  # K = 2880 = 45 K tiles: odd -> falls back to the unsplit form
    for Cin in (320, 384):                                       # 384: K = 3456 = 54 tiles -> split
        x, w, b = _rand(g, R, Cin, H, W), _rand(g, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5), _rand(g, Cout)
        want = _nhwc(F.conv2d(x.float(), w.float(), b.float(), padding=1)).reshape(R * H * W, Cout)
        ws = ops.splitk_workspace(R * H * W, Cout, dev)
        out = torch.empty(R * H * W, Cout, dtype=torch.half, device=dev)
        ops.conv2d(_nhwc(x).to(dev), conv_weight_nhwc(w).to(dev), out, ksize=3, bias=b.to(dev), splitk_ws=ws)
        torch.cuda.synchronize()
        _close(out, want, what=f"split-K conv Cin={Cin}")
        assert ws[:4096].view(torch.int32).abs().max().item() == 0


@pytest.mark.parametrize("M,C,tile", [(256, 160, 0), (2048, 640, 0), (200, 128, 0), (512, 128, 2), (96, 64, 3), (1024, 320, 4), (1024, 320, 55)])
def test_gemm_transposed_column_range(dev, M, C, tile):
    """Fused q|k|v projection: columns [0, 2C) row-major, columns [2C, 3C) stored transposed (iir_gemm_desc.Ct) -- whole
    transposed tiles, tiles straddling tr_from and ragged M all give the two plain products."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(M + C)
    K = 128
    a, w, bias = _rand(g, M, K), _rand(g, 3 * C, K, scale=K ** -0.5), _rand(g, 3 * C)
    want = a.float() @ w.float().T + bias.float()
    qk = torch.zeros(M, 2 * C + 8, dtype=torch.half, device=dev)
    vt = torch.zeros(C, M + 8, dtype=torch.half, device=dev)
    ops.gemm(a.to(dev), w.to(dev), qk[:, :2 * C], bias=bias.to(dev), out_t=(vt[:, :M], 2 * C), tile=tile)
    torch.cuda.synchronize()
    _close(qk[:, :2 * C], want[:, :2 * C], what="q|k part")
    _close(vt[:, :M], want[:, 2 * C:].T, what="V^T part")
    assert qk[:, 2 * C:].abs().max().item() == 0 and vt[:, M:].abs().max().item() == 0


def test_gemm_strided_a_and_scale(dev):
    from instantir_amd import ops
    g = torch.Generator().manual_seed(3)
    big = _rand(g, 200, 256)
    a = big[:, 64:192]                                   # column slice: lda 256, K 128
    w = _rand(g, 64, 128, scale=0.1)
    out = torch.empty(200, 64, dtype=torch.half, device=dev)
    bigd = big.to(dev)
    ops.gemm(bigd[:, 64:192], w.to(dev), out, out_scale=0.5)
    torch.cuda.synchronize()
    _close(out, 0.5 * (a.float() @ w.float().T), what="gemm strided")


@pytest.mark.parametrize("tile", [0, 55, 90])          # 90: 256x320, 8 waves (paired epilogues only)
@pytest.mark.parametrize("M,n_out,K", [(256, 128, 64), (130, 48, 128), (64, 2560, 640), (520, 400, 256)])
def test_gemm_geglu(dev, M, n_out, K, tile):
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    g = torch.Generator().manual_seed(n_out)
    a, w, b = _rand(g, M, K), _rand(g, 2 * n_out, K, scale=K ** -0.5), _rand(g, 2 * n_out)
    h = a.float() @ w.float().T + b.float()
    want = h[:, :n_out] * F.gelu(h[:, n_out:])
    wp = pair_rows(w[:n_out], w[n_out:]).to(dev)
    bp = pair_rows(b[:n_out], b[n_out:]).to(dev)
    out = torch.empty(M, n_out, dtype=torch.half, device=dev)
    ops.gemm(a.to(dev), wp, out, bias=bp, epi=ops.EPI_GEGLU, tile=tile)
    torch.cuda.synchronize()
    _close(out, want, what="geglu")


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("R,H,W,Cin,Cout,stride,ups", [
    (2, 16, 16, 64, 64, 1, False), (1, 9, 7, 128, 96, 1, False), (2, 16, 16, 64, 128, 2, False),
    (2, 8, 8, 64, 64, 1, True), (1, 32, 16, 320, 320, 1, False)])
@pytest.mark.parametrize("tile", [0, 55])          # 55: 64x160 with loader waves (the one-workgroup-per-CU convs)
def test_conv3x3(dev, R, H, W, Cin, Cout, stride, ups, tile):
    from instantir_amd import ops
    from instantir_amd.packing import conv_weight_nhwc
    g = torch.Generator().manual_seed(H * W + Cin)
    x, w, b = _rand(g, R, Cin, H, W), _rand(g, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5), _rand(g, Cout)
    temb = _rand(g, R, Cout)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    want = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=1) + temb.float()[:, :, None, None]
    Ho, Wo = want.shape[2:]
    res = _rand(g, R, Cout, Ho, Wo)
    want = _nhwc(want + res.float()).reshape(R * Ho * Wo, Cout)
    xd = torch.zeros(R, H, W, Cin + 64, dtype=torch.half, device=dev)       # pixel stride > Cin
    xd[..., :Cin] = _nhwc(x).to(dev)
    out = torch.empty(R * Ho * Wo, Cout, dtype=torch.half, device=dev)
    ops.conv2d(xd[..., :Cin], conv_weight_nhwc(w).to(dev), out, ksize=3, stride=stride, upsample=ups, bias=b.to(dev),
               rowbias=temb.to(dev), rows_per_rb=Ho * Wo, res=_nhwc(res).reshape(-1, Cout).to(dev), tile=tile)
    torch.cuda.synchronize()
    _close(out, want, what="conv3x3")


def test_conv_padded_input_channels_and_1x1(dev):
    from instantir_amd import ops
    from instantir_amd.packing import conv_weight_nhwc
    g = torch.Generator().manual_seed(5)
    x, w, b = _rand(g, 2, 4, 16, 16), _rand(g, 64, 4, 3, 3, scale=1 / 6), _rand(g, 64)
    want = _nhwc(F.conv2d(x.float(), w.float(), b.float(), padding=1)).reshape(-1, 64)
    xd = torch.zeros(2, 16, 16, 64, dtype=torch.half, device=dev)
    xd[..., :4] = _nhwc(x).to(dev)
    out = torch.empty(2 * 256, 64, dtype=torch.half, device=dev)
    ops.conv2d(xd, conv_weight_nhwc(w, 64).to(dev), out, bias=b.to(dev))
    torch.cuda.synchronize()
    _close(out, want, what="conv_in")
    w1 = _rand(g, 128, 64, 1, 1, scale=1 / 8)
    y = torch.empty(512, 128, dtype=torch.half, device=dev)
    ops.conv2d(out.view(2, 16, 16, 64), conv_weight_nhwc(w1).to(dev), y, ksize=1)
    torch.cuda.synchronize()
    _close(y, out.float().cpu() @ w1.float().reshape(128, 64).T, what="conv1x1")


def test_conv_sft_epilogue(dev):
    """module/aggregator.py:76-86: h * (mul(actv) + 1) + add(actv) with the two convs fused."""
    from instantir_amd import ops
    from instantir_amd.packing import conv_weight_nhwc, pair_rows
    g = torch.Generator().manual_seed(9)
    R, H, W, Ch, C = 2, 8, 8, 64, 96
    actv, h = _rand(g, R, Ch, H, W), _rand(g, R, C, H, W)
    wm, wa = _rand(g, C, Ch, 3, 3, scale=1 / 24), _rand(g, C, Ch, 3, 3, scale=1 / 24)
    bm, ba = _rand(g, C), _rand(g, C)
    gamma = F.conv2d(actv.float(), wm.float(), bm.float(), padding=1)
    beta = F.conv2d(actv.float(), wa.float(), ba.float(), padding=1)
    want = _nhwc(h.float() * (gamma + 1) + beta).reshape(-1, C)
    wp = pair_rows(conv_weight_nhwc(wm), conv_weight_nhwc(wa)).to(dev)
    bp = pair_rows(bm, ba).to(dev)
    out = torch.empty(R * H * W, C, dtype=torch.half, device=dev)
    ops.conv2d(_nhwc(actv).to(dev), wp, out, bias=bp, res=_nhwc(h).reshape(-1, C).to(dev), epi=ops.EPI_SFT)
    torch.cuda.synchronize()
    _close(out, want, what="sft")


def _sdpa_ref(q, k, v, heads):
    b, t, c = q.shape
    sp = lambda x: x.float().reshape(b, -1, heads, 64).transpose(1, 2)
    o = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / 8.0, dim=-1) @ sp(v)
    return o.transpose(1, 2).reshape(b, t, c)


@pytest.mark.parametrize("B,heads,Tq,Tkv", [(2, 2, 256, 256), (1, 3, 200, 200), (2, 1, 50, 136), (1, 2, 16, 16)])
def test_self_attention(dev, B, heads, Tq, Tkv):
    from instantir_amd import ops
    g = torch.Generator().manual_seed(Tq + heads)
    C = heads * 64
    q, k, v = _rand(g, B, Tq, C), _rand(g, B, Tkv, C), _rand(g, B, Tkv, C)
    want = _sdpa_ref(q, k, v, heads).reshape(B * Tq, C)
    # q, k live as column slices of one fused buffer (like the QKV GEMM output)
    qk = torch.zeros(B * max(Tq, Tkv), 2 * C, dtype=torch.half, device=dev)
    qd = torch.empty(B * Tq, C, dtype=torch.half, device=dev); qd.copy_(q.reshape(-1, C))
    kbuf = torch.zeros(B * Tkv, 2 * C, dtype=torch.half, device=dev)
    kbuf[:, C:] = k.reshape(-1, C).to(dev)
    tpad = (Tkv + 7) // 8 * 8
    vt = torch.zeros(C, B * tpad, dtype=torch.half, device=dev)
    for b in range(B):
        vt[:, b * tpad:b * tpad + Tkv] = v[b].T.to(dev)
    o = torch.empty(B * Tq, C, dtype=torch.half, device=dev)
    ops.attention(qd, o, [(kbuf[:, C:], Tkv, vt, tpad, Tkv)], B, heads, Tq)
    torch.cuda.synchronize()
    _close(o, want, rtol=3e-3, atol=3e-3, what="self-attn")


def test_two_segment_attention(dev):
    """text KV (13 keys) + IP KV (16 keys) sharing one query, outputs summed
    (module/ip_adapter/attention_processor.py:1165,1185,1192)."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(21)
    B, heads, Tq = 2, 2, 100
    C = heads * 64
    q = _rand(g, B, Tq, C)
    k1, v1, k2, v2 = _rand(g, B, 13, C), _rand(g, B, 13, C), _rand(g, B, 16, C), _rand(g, B, 16, C)
    want = (_sdpa_ref(q, k1, v1, heads) + _sdpa_ref(q, k2, v2, heads)).reshape(B * Tq, C)

    def vt_of(v, T):
        tp = (T + 7) // 8 * 8
        out = torch.zeros(C, B * tp, dtype=torch.half, device=dev)
        for b in range(B):
            out[:, b * tp:b * tp + T] = v[b].T.to(dev)
        return out, tp

    vt1, tp1 = vt_of(v1, 13)
    vt2, tp2 = vt_of(v2, 16)
    o = torch.empty(B * Tq, C, dtype=torch.half, device=dev)
    ops.attention(q.reshape(-1, C).to(dev), o,
                  [(k1.reshape(-1, C).to(dev), 13, vt1, tp1, 13), (k2.reshape(-1, C).to(dev), 16, vt2, tp2, 16)],
                  B, heads, Tq)
    torch.cuda.synchronize()
    _close(o, want, rtol=3e-3, atol=3e-3, what="2-seg attn")


def test_attention_online_softmax_rescale(dev):
    """Force the running-max update late: one key far above the rest sits in the last tile."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(2)
    B, heads, T = 1, 1, 256
    q, k, v = _rand(g, B, T, 64), _rand(g, B, T, 64), _rand(g, B, T, 64)
    k[0, 250] = q[0, 3] * 4.0
    want = _sdpa_ref(q, k, v, heads).reshape(T, 64)
    vt = v[0].T.contiguous().to(dev)
    o = torch.empty(T, 64, dtype=torch.half, device=dev)
    ops.attention(q.reshape(-1, 64).to(dev), o, [(k.reshape(-1, 64).to(dev), T, vt, T, T)], B, heads, T)
    torch.cuda.synchronize()
    _close(o, want, rtol=3e-3, atol=3e-3, what="attn rescale")


@pytest.mark.parametrize("R,HW,C,silu,eps", [(2, 256, 64, True, 1e-5), (2, 1024, 320, True, 1e-5), (1, 64, 2560, False, 1e-6),
                                              (3, 100, 960, True, 1e-5)])
def test_groupnorm(dev, R, HW, C, silu, eps):
    from instantir_amd import ops
    g = torch.Generator().manual_seed(C)
    x = _rand(g, R, HW, C) + 0.5
    gm, bt = _rand(g, C) + 1, _rand(g, C)
    want = F.group_norm(x.float().permute(0, 2, 1), 32, gm.float(), bt.float(), eps).permute(0, 2, 1)
    if silu:
        want = F.silu(want)
    xd = torch.zeros(R * HW, C + 64, dtype=torch.half, device=dev)
    xd[:, :C] = x.reshape(-1, C).to(dev)
    out = torch.empty(R * HW, C, dtype=torch.half, device=dev)
    ops.groupnorm(xd[:, :C], out, R, HW, gm.to(dev), bt.to(dev), eps, silu)
    torch.cuda.synchronize()
    _close(out, want.reshape(-1, C), rtol=3e-3, atol=3e-3, what="groupnorm")


def test_layernorm_variants(dev):
    from instantir_amd import ops
    g = torch.Generator().manual_seed(4)
    rows, C = 37, 640
    x, gm, bt = _rand(g, rows, C) * 2 + 0.3, _rand(g, C) + 1, _rand(g, C)
    out = torch.empty(rows, C, dtype=torch.half, device=dev)
    ops.layernorm(x.to(dev), out, gm.to(dev), bt.to(dev), 1e-5)
    torch.cuda.synchronize()
    _close(out, F.layer_norm(x.float(), (C,), gm.float(), bt.float(), 1e-5), what="ln")
    # adaLN: 2 batches x 16 tokens, LN without affine, (1+scale), shift  (attention_processor.py:24-25)
    x2 = _rand(g, 32, 128)
    mod = _rand(g, 2, 256, scale=0.3)                    # [shift | scale] per batch row
    want = F.layer_norm(x2.float(), (128,), None, None, 1e-6).reshape(2, 16, 128) * (1 + mod.float()[:, None, 128:]) \
        + mod.float()[:, None, :128]
    modd = mod.to(dev)
    out2 = torch.empty(32, 128, dtype=torch.half, device=dev)
    ops.layernorm(x2.to(dev), out2, eps=1e-6, shift=modd[:, :128], scale=modd[:, 128:], rows_per_mod=16)
    torch.cuda.synchronize()
    _close(out2, want.reshape(32, 128), what="adaLN")
    outT = torch.zeros(128, 2 * 16, dtype=torch.half, device=dev)
    ops.layernorm(x2.to(dev), outT, eps=1e-6, shift=modd[:, :128], scale=modd[:, 128:], rows_per_mod=16, transposed=True,
                  tr_rows=16, tr_bstride=16)
    torch.cuda.synchronize()
    _close(outT, want.reshape(32, 128).T, what="adaLN^T")


def test_adaln_batch_matches_single_launches(dev):
    """iir_adaln_batch_f16: one launch over a job table == the per-block iir_layernorm_f16 launches, bit for bit."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(41)
    R, nip, ipad = 2, 16, 16
    mod = _rand(g, R, 2 * (128 + 256) * 2, scale=0.3).to(dev)
    jobs, singles, off = [], [], 0
    for C in (128, 256):
        for tr in (False, True):
            x = (_rand(g, R * nip, C) * 1.5 + 0.2).to(dev)
            out = torch.zeros((C, R * ipad) if tr else (R * nip, C), dtype=torch.half, device=dev)
            ref = torch.zeros_like(out)
            sh, sc = mod[:, off:off + C], mod[:, off + C:off + 2 * C]
            off += 2 * C
            jobs.append((x, out, sh, sc, tr))
            ops.layernorm(x, ref, eps=1e-6, shift=sh, scale=sc, rows_per_mod=nip, transposed=tr, tr_rows=nip, tr_bstride=ipad)
            singles.append(ref)
    table = ops.adaln_job_table(jobs, dev)
    ops.adaln_batch(table, len(jobs), R * nip, 256, mod.stride(0), nip, nip, ipad)
    torch.cuda.synchronize()
    for (x, out, *_), ref in zip(jobs, singles):
        assert torch.equal(out, ref)
    want = F.layer_norm(jobs[0][0].float(), (128,), None, None, 1e-6).reshape(R, nip, 128) * (1 + jobs[0][3].float()[:, None]) \
        + jobs[0][2].float()[:, None]
    _close(jobs[0][1], want.reshape(R * nip, 128).cpu(), what="adaLN batch")


def test_cfg_rescale_factor(dev):
    """iir_cfg_rescale_factor + eps_factor of iir_sched_step == rescale_noise_cfg (pipelines/sdxl_instantir.py:181-192)."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(77)
    B, H, Wd, phi, gs = 3, 12, 20, 0.7, 6.5
    eps = (_rand(g, 2 * B * H * Wd, 64) * 1.3)
    eps_d = eps.to(dev)
    x = torch.randn(B, 4, H, Wd, generator=g)
    coef = torch.tensor([gs, 0.6, 0.8, 0.0, 0.0, 1.0, 0.0, 0.0], device=dev)       # prev = eps (k_eps = 1): exposes the guided eps
    fac = torch.zeros(B, device=dev)
    ops.cfg_rescale_factor(eps_d, B, coef, x.to(dev), phi, fac)
    e4 = eps.float()[:, :4].reshape(2, B, H * Wd, 4).permute(0, 1, 3, 2).reshape(2, B, 4, H, Wd)
    u, c = e4[0], e4[1]
    cfg_ = u + gs * (c - u)
    dims = [1, 2, 3]
    ratio = c.std(dim=dims, keepdim=True) / cfg_.std(dim=dims, keepdim=True)
    want_fac = (phi * ratio + (1 - phi)).reshape(B)
    torch.cuda.synchronize()
    assert torch.allclose(fac.cpu(), want_fac, rtol=1e-5, atol=1e-6)
    prev = torch.empty(B, 4, H, Wd, device=dev)
    ops.sched_step(eps_d, B, coef, x.to(dev), prev, cfg=True, eps_factor=fac)
    torch.cuda.synchronize()
    want = phi * (cfg_ * ratio) + (1 - phi) * cfg_
    assert torch.allclose(prev.cpu(), want, rtol=1e-5, atol=1e-5)


def test_pointwise(dev):
    from instantir_amd import ops
    g = torch.Generator().manual_seed(8)
    # sinusoid: [cos | sin], module/min_sdxl.py:205-224
    vals = torch.tensor([[958.0, 1024.0, 0.0], [1.0, 512.0, 3.0]])
    out = torch.zeros(2, 3 * 64 + 8, dtype=torch.half, device=dev)
    ops.sinusoid(vals.to(dev), out, 64, col_off=8)
    k = torch.arange(32, dtype=torch.float32)
    ang = vals.reshape(-1, 1) * torch.exp(-math.log(10000) * k / 32)[None]
    want = torch.cat([ang.cos(), ang.sin()], -1).reshape(2, 192)
    torch.cuda.synchronize()
    _close(out[:, 8:], want, rtol=2e-3, atol=2e-3, what="sinusoid")
    # silu, copy_add, transpose
    x = _rand(g, 40, 64)
    y = torch.empty(40, 64, dtype=torch.half, device=dev)
    ops.silu(x.to(dev), y)
    _close(y, F.silu(x.float()), what="silu")
    dst = torch.zeros(40, 192, dtype=torch.half, device=dev)
    add = _rand(g, 40, 64)
    sc = torch.tensor([0.5, 2.0], device=dev)
    ops.copy_add(x.to(dev), dst, dst_off=128, add=add.to(dev), add_scale=sc, rows_per_scale=20)
    want = x.float() + add.float() * torch.tensor([0.5] * 20 + [2.0] * 20)[:, None]
    torch.cuda.synchronize()
    _close(dst[:, 128:], want, what="copy_add")
    assert dst[:, :128].abs().max().item() == 0
    t = torch.full((64, 48), 7.0, dtype=torch.half, device=dev)
    ops.transpose(x.to(dev), t, rows_pad=48)
    torch.cuda.synchronize()
    assert torch.equal(t[:, :40].cpu(), x.T) and t[:, 40:].abs().max().item() == 0




@pytest.mark.parametrize("M,N,K,mode,tile", [(256, 320, 640, "plain", 0), (300, 132, 128, "plain", 2), (2048, 1280, 1280, "plain", 0),
                                              (2048, 1280, 1280, "plain", 24), (512, 1280, 640, "geglu", 0), (130, 96, 256, "geglu", 3),
                                              (256, 3 * 128, 128, "qkv", 0), (1024, 640, 1280, "plain", 5), (64, 64, 64, "plain", 1)])
def test_gemm_fp8_weights(dev, M, N, K, mode, tile):
    """`wscale=`: fp8-E4M3 weights with per-output-channel scales, activations converted to fp8 inside the kernel
    (v_mfma_f32_16x16x32_fp8_fp8).  Reference: the SAME quantised operands (torch's E4M3 casts) multiplied in fp32 -- this
    isolates the kernel (layout, k order of the packed conversions, scale / bias / residual / GEGLU / transposed epilogues)
    from the quantisation error itself, which is configs[4]'s own tolerance (tests/test_pipeline_gpu.py)."""
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    g = torch.Generator().manual_seed(M + N + K)
    x = _rand(g, M, K) * 1.5
    w, b = _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    xq = x.to(torch.float8_e4m3fn).float()
    if mode == "geglu":
        wp, bp = pair_rows(w[:N // 2], w[N // 2:]), pair_rows(b[:N // 2], b[N // 2:])
        q, sc = ops.quantize_fp8_rows(wp)
        full = xq @ (q.float() * sc[:, None]).T + bp.float()
        # paired rows: in every 16-row group the first 8 are values, the next 8 their gates
        v = full.reshape(M, N // 16, 2, 8)
        want = (v[:, :, 0] * F.gelu(v[:, :, 1])).reshape(M, N // 2)
        out = torch.empty(M, N // 2, dtype=torch.half, device=dev)
        ops.gemm(x.to(dev), q.to(dev), out, bias=bp.to(dev), epi=ops.EPI_GEGLU, tile=tile, wscale=sc.to(dev))
        torch.cuda.synchronize()
        _close(out, want, rtol=4e-3, atol=4e-3, what="fp8 GEGLU")
        return
    q, sc = ops.quantize_fp8_rows(w)
    full = xq @ (q.float() * sc[:, None]).T
    if mode == "qkv":
        C = N // 3
        qk = torch.empty(M, 2 * C, dtype=torch.half, device=dev)
        vt = torch.empty(C, M, dtype=torch.half, device=dev)
        ops.gemm(x.to(dev), q.to(dev), qk, out_t=(vt, 2 * C), tile=tile, wscale=sc.to(dev))
        torch.cuda.synchronize()
        _close(qk, full[:, :2 * C], rtol=4e-3, atol=4e-3, what="fp8 q|k")
        _close(vt, full[:, 2 * C:].T, rtol=4e-3, atol=4e-3, what="fp8 V^T")
        return
    res = _rand(g, M, N)
    out = torch.empty(M, N, dtype=torch.half, device=dev)
    ops.gemm(x.to(dev), q.to(dev), out, bias=b.to(dev), res=res.to(dev), tile=tile, wscale=sc.to(dev))
    torch.cuda.synchronize()
    _close(out, full + b.float() + res.float(), rtol=4e-3, atol=4e-3, what="fp8 gemm")


@pytest.mark.parametrize("M,C,N,geglu", [(2048, 1280, 1280, False), (2048, 1280, 2560, True), (8192, 640, 1920, False),
                                         (512, 128, 384, False), (256, 320, 640, True)])
def test_gemm_layernorm_fold(dev, M, C, N, geglu):
    """LayerNorm without a LayerNorm launch: the GEMM that writes the residual stream leaves per-row (mean, M2) partials per
    column tile (`ln_out`), the GEMM after the LayerNorm reads the RAW rows with gamma folded into its weight and applies
    rstd * acc - rstd * mean * colsum + (bias + W beta) in its epilogue (`ln_in`).  Against torch: LayerNorm -> Linear [-> GEGLU]
    in fp32 on the fp16 residual stream the producer wrote; rows carry a mean several times their spread."""
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    g = torch.Generator().manual_seed(M + C + N)
    K0 = C                                                                 # the producer's K, as in the engine (to_out / proj_in)
    a0, w0 = _rand(g, M, K0), _rand(g, C, K0, scale=K0 ** -0.5)
    res = (_rand(g, M, C).float() + 4.0 * torch.randn(M, 1, generator=g)).half()          # per-row offsets: |mean| >> std
    gamma, beta = (1.0 + 0.2 * torch.randn(C, generator=g)).half(), (0.3 * torch.randn(C, generator=g)).half()
    w, b = _rand(g, N, C, scale=C ** -0.5), _rand(g, N)
    parts = ops.ln_parts(M, C, K0)
    assert 0 < parts <= 8 and C % parts == 0
    h = torch.empty(M, C, dtype=torch.half, device=dev)
    stats = torch.zeros(parts, M, 2, dtype=torch.float32, device=dev)
    ops.gemm(a0.to(dev), w0.to(dev), h, res=res.to(dev), ln_out=stats)
    torch.cuda.synchronize()
    hf = h.float().cpu()
    _close(h, a0.float() @ w0.float().T + res.float(), what="producer output")
    # the partials merge to the row statistics of what was stored
    cols = C // parts
    blocks = hf.reshape(M, parts, cols)
    np.testing.assert_allclose(stats[:, :, 0].cpu().T.numpy(), blocks.mean(-1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(stats[:, :, 1].cpu().T.numpy(), ((blocks - blocks.mean(-1, keepdim=True)) ** 2).sum(-1).numpy(), rtol=2e-4, atol=1e-3)
    y = F.layer_norm(hf, (C,), gamma.float(), beta.float(), 1e-5) @ w.float().T + b.float()
    fold = ops.LnFold(w.to(dev), gamma.to(dev), beta.to(dev), bias=b.to(dev), eps=1e-5, pair=pair_rows if geglu else None)
    if geglu:
        want = y[:, :N // 2] * F.gelu(y[:, N // 2:])
        out = torch.empty(M, N // 2, dtype=torch.half, device=dev)
        ops.gemm(h, fold.w, out, bias=fold.bias, epi=ops.EPI_GEGLU, ln_in=(stats, fold.colsum, fold.eps))
    else:
        want = y
        out = torch.empty(M, N, dtype=torch.half, device=dev)
        ops.gemm(h, fold.w, out, bias=fold.bias, ln_in=(stats, fold.colsum, fold.eps))
    torch.cuda.synchronize()
    tol = 8e-3 if geglu else 4e-3              # (GEGLU multiplies two quantities that each carry the fp16 / folding error)
    _close(out, want, rtol=tol, atol=tol, what="LayerNorm-folded gemm")
    with pytest.raises(ValueError):
        ops.gemm(h, fold.w, out, bias=fold.bias, epi=ops.EPI_GEGLU if geglu else ops.EPI_PLAIN, ln_in=(stats[:, :M // 2], fold.colsum, fold.eps))


def test_groupnorm_reproducible_beside_a_conv(dev):
    """Regression: GroupNorm must give bit-identical results when another stream keeps the chip busy (the step runs the main
    UNet's encoder beside the previewer UNet + Aggregator).  A wave-butterfly form of the finalize kernel produced run-to-run
    different statistics exactly then -- only beside conv kernels, never alone -- which made the whole pipeline irreproducible
    (found by `tools/racecheck_concurrent.py`)."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(1)
    side = torch.cuda.Stream()
    cx, cw = _rand(g, 2, 64, 64, 640).to(dev), _rand(g, 640, 3, 3, 640, scale=0.02).to(dev)
    co = torch.empty(2 * 64 * 64, 640, dtype=torch.half, device=dev)
    for R, HW, C in ((2, 4096, 640), (2, 16384, 320), (2, 1024, 1280)):
        x, gm, bt = _rand(g, R * HW, C).to(dev), (_rand(g, C) + 1).to(dev), _rand(g, C).to(dev)
        ws = ops.gn_workspace(dev, R, 32)
        first = None
        for it in range(16):
            ws.fill_(float(it))                          # stale workspace contents must never show
            out = torch.zeros(R * HW, C, dtype=torch.half, device=dev)
            with torch.cuda.stream(side):
                for _ in range(3):
                    ops.conv2d(cx, cw, co)
            ops.groupnorm(x, out, R, HW, gm, bt, 1e-5, True, 32, ws)
            torch.cuda.synchronize()
            if first is None:
                first = out
            else:
                assert torch.equal(out, first), (R, HW, C, it)


# ---- round 3: the 8-wave 256 x BN kernel (gemm8.hip) --------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,mode,tile", [(256, 320, 128, "plain", 91), (512, 640, 192, "res", 91), (512, 640, 256, "geglu", 91),
                                             (256, 640, 320, "silu", 91), (512, 512, 256, "res", 92), (768, 960, 1280, "geglu", 91),
                                             (2048, 10240, 1280, "geglu", 0)])
def test_gemm8_tile(dev, M, N, K, mode, tile):
    """256 x 320 / 256 x 256 output tile, 8 waves, two-tile-deep LDS-DMA pipeline, chunked epilogue: against torch fp32, and bit
    for bit against the 4-wave kernel (same fp32 accumulation order per K step, same epilogue arithmetic).  tile = 0 with the
    2048 x 10240 x 1280 GEGLU projection is the launch the engine's chooser sends to this kernel."""
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    g = torch.Generator().manual_seed(M + N + K)
    a, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    geglu = mode == "geglu"
    n_out = N // 2 if geglu else N
    res = _rand(g, M, n_out) if mode == "res" else None
    y = a.float() @ w.float().T + b.float()
    if geglu:
        want = y[:, :N // 2] * F.gelu(y[:, N // 2:])
        wd, bd = pair_rows(w[:N // 2], w[N // 2:]).to(dev), pair_rows(b[:N // 2], b[N // 2:]).to(dev)
    else:
        want = F.silu(y) if mode == "silu" else y
        if res is not None:
            want = want + res.float()
        wd, bd = w.to(dev), b.to(dev)
    kw = dict(bias=bd, epi=ops.EPI_GEGLU if geglu else ops.EPI_PLAIN, act=ops.ACT_SILU if mode == "silu" else ops.ACT_NONE)
    if res is not None:
        kw["res"] = res.to(dev)
    out, old = (torch.zeros(M, n_out, dtype=torch.half, device=dev) for _ in range(2))
    ops.gemm(a.to(dev), wd, out, tile=tile, **kw)
    ops.gemm(a.to(dev), wd, old, tile=24, **kw)                 # 128 x 160, 4 waves
    torch.cuda.synchronize()
    tol = 8e-3 if geglu else 3e-3
    _close(out, want, rtol=tol, atol=tol, what=f"gemm8 {mode}")
    assert torch.equal(out, old), "gemm8 differs from the 4-wave kernel"


def test_gemm8_transposed_v_and_layernorm_fold(dev):
    """The fused q|k|v projection on the 256 x 320 tile: LayerNorm folded in (`ln_in`), the V third written transposed."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(5)
    M, C = 512, 320
    a0, w0 = _rand(g, M, C), _rand(g, C, C, scale=C ** -0.5)
    gamma, beta = (1.0 + 0.2 * torch.randn(C, generator=g)).half(), (0.3 * torch.randn(C, generator=g)).half()
    w, b = _rand(g, 3 * C, C, scale=C ** -0.5), _rand(g, 3 * C)
    stats = torch.zeros(ops.ln_parts(M, C, C), M, 2, dtype=torch.float32, device=dev)
    h = torch.empty(M, C, dtype=torch.half, device=dev)
    ops.gemm(a0.to(dev), w0.to(dev), h, ln_out=stats)
    fold = ops.LnFold(w.to(dev), gamma.to(dev), beta.to(dev), bias=b.to(dev), eps=1e-5)
    qk = torch.zeros(M, 2 * C, dtype=torch.half, device=dev)
    vt = torch.zeros(C, M, dtype=torch.half, device=dev)
    ops.gemm(h, fold.w, qk, bias=fold.bias, tile=91, out_t=(vt, 2 * C), ln_in=(stats, fold.colsum, fold.eps))
    torch.cuda.synchronize()
    y = F.layer_norm(h.float().cpu(), (C,), gamma.float(), beta.float(), 1e-5) @ w.float().T + b.float()
    _close(qk, y[:, :2 * C], rtol=4e-3, atol=4e-3, what="q|k")
    _close(vt, y[:, 2 * C:].T, rtol=4e-3, atol=4e-3, what="V^T")


def test_gemm8_rejects_what_it_does_not_cover(dev):
    from instantir_amd import ops
    a = torch.zeros(300, 128, dtype=torch.half, device=dev)             # M not a multiple of 256
    w = torch.zeros(320, 128, dtype=torch.half, device=dev)
    with pytest.raises(Exception):
        ops.gemm(a, w, torch.zeros(300, 320, dtype=torch.half, device=dev), tile=91)


# ---- round 3: the large-grid attention builds and the pre-staged short-KV form ---------------------------------------------
@pytest.mark.parametrize("T", [4096, 8192])
def test_self_attention_large_grid(dev, T):
    """T = 4096 (the level-1 self-attention of the step: 3 waves per SIMD build, the one with spilled registers) and T = 8192
    (the Aggregator's): B = 1, 2 heads would only make 64 / 128 workgroups, so the grid is widened to the step's own size
    with B = 2, 10 heads and ONE (batch, head) pair is checked against fp32 SDPA, the others against the first kernel
    generation-independent property that equal inputs give equal outputs."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(T)
    B, heads = 2, 10
    C = heads * 64
    pair = _rand(g, 3, T, 64)                                    # q, k, v of one (batch, head) pair, copied into every pair
    q = pair[0].repeat(B, heads).reshape(B * T, C).contiguous().to(dev)
    k = pair[1].repeat(B, heads).reshape(B * T, C).contiguous().to(dev)
    vt = pair[2].T.contiguous().repeat(heads, B).to(dev)          # (C, B*T): V^T of every pair
    o = torch.empty(B * T, C, dtype=torch.half, device=dev)
    ops.attention(q, o, [(k, T, vt, T, T)], B, heads, T)
    torch.cuda.synchronize()
    want = _sdpa_ref(pair[0][None], pair[1][None], pair[2][None], 1).reshape(T, 64)
    got = o.reshape(B, T, heads, 64)
    _close(got[0, :, 0], want, rtol=3e-3, atol=3e-3, what=f"attention T={T}")
    for b_ in range(B):
        for h_ in range(heads):
            assert torch.equal(got[b_, :, h_], got[0, :, 0]), "identical (batch, head) pairs gave different outputs"


@pytest.mark.parametrize("T", [1024, 4096])
def test_cross_attention_prestaged(dev, T):
    """Text (77) + IP (64) keys: 3 tiles in all, the pre-staged form (every tile requested at entry, one wait) at the step's
    query counts."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(T + 1)
    B, heads = 2, 4
    C = heads * 64
    q = _rand(g, B, T, C)
    segs, want = [], 0
    for L in (77, 64):
        k, v = _rand(g, B, L, C), _rand(g, B, L, C)
        tp = (L + 7) // 8 * 8
        vt = torch.zeros(C, B * tp, dtype=torch.half, device=dev)
        for b_ in range(B):
            vt[:, b_ * tp:b_ * tp + L] = v[b_].T.to(dev)
        segs.append((k.reshape(-1, C).to(dev), L, vt, tp, L))
        want = want + _sdpa_ref(q, k, v, heads)
    o = torch.empty(B * T, C, dtype=torch.half, device=dev)
    ops.attention(q.reshape(-1, C).to(dev), o, segs, B, heads, T)
    torch.cuda.synchronize()
    _close(o, want.reshape(B * T, C), rtol=3e-3, atol=3e-3, what="text + IP cross-attention")


# ---- round 3: GroupNorm statistics from the producing launch ---------------------------------------------------------------
def _slab_stats(y):
    """(mean, M2) per 64-row slab and channel of a (rows, C) tensor, fp64."""
    z = y.double().reshape(-1, 64, y.shape[1])
    mu = z.mean(1)
    return mu, ((z - mu[:, None]) ** 2).sum(1)


@pytest.mark.parametrize("M,N,K,res", [(2048, 1280, 1280, True), (8192, 640, 640, True), (2048, 320, 640, False), (256, 640, 128, True)])
def test_gemm_groupnorm_partials(dev, M, N, K, res):
    """`gn_out`: the GEMM that writes a GroupNorm's input leaves (mean, M2) of the STORED fp16 values per 64-row slab and channel
    (module/min_sdxl.py:565-595: proj_out + residual -> next block's GroupNorm)."""
    from instantir_amd import ops
    if not ops.gn_supported(M, N, K):
        pytest.skip("tile of this shape cannot emit partials")
    g = torch.Generator().manual_seed(M + N)
    a, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    r = (_rand(g, M, N).float() + 3.0 * torch.randn(1, N, generator=g)).half() if res else None      # channel means far above the spread
    out = torch.empty(M, N, dtype=torch.half, device=dev)
    part = torch.zeros(M // 64, N, 2, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), w.to(dev), out, bias=b.to(dev), res=r.to(dev) if res else None, gn_out=part)
    plain = torch.empty_like(out)
    ops.gemm(a.to(dev), w.to(dev), plain, bias=b.to(dev), res=r.to(dev) if res else None)
    torch.cuda.synchronize()
    assert torch.equal(out, plain), "the statistics path changed the stored output"
    mu, m2 = _slab_stats(out.cpu())
    np.testing.assert_allclose(part[:, :, 0].cpu().numpy(), mu.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(part[:, :, 1].cpu().numpy(), m2.numpy(), rtol=3e-4, atol=2e-3)


@pytest.mark.parametrize("R,H,Cin,Cout,stride,with_res", [(2, 32, 1280, 1280, 1, True), (2, 64, 640, 640, 1, False), (1, 64, 320, 320, 1, True),
                                                          (2, 64, 640, 640, 2, False), (2, 16, 64, 320, 1, False)])
@pytest.mark.parametrize("silu", [True, False])
def test_conv_groupnorm_from_partials(dev, R, H, Cin, Cout, stride, with_res, silu):
    """conv (+ temb row bias, + residual) -> GroupNorm (+ SiLU) without a statistics pass: the conv's `gn_out` partials feed
    `groupnorm(partials=...)`; against torch's group_norm on the conv's stored output, and against the three-launch form."""
    from instantir_amd import ops
    from instantir_amd.packing import conv_weight_nhwc
    Ho = H // stride
    Mo, HW = R * Ho * Ho, Ho * Ho
    if HW % 64 or not ops.gn_supported(Mo, Cout, 9 * Cin, True):
        pytest.skip("geometry without whole 64-row slabs / tile cannot emit partials")
    g = torch.Generator().manual_seed(H + Cin + stride)
    x = _rand(g, R, H, H, Cin)
    w = _rand(g, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
    b, rb = _rand(g, Cout), _rand(g, R, Cout)
    res = (_rand(g, Mo, Cout).float() + 2.0).half() if with_res else None
    gam, bet = (1.0 + 0.2 * torch.randn(Cout, generator=g)).half(), (0.2 * torch.randn(Cout, generator=g)).half()
    y = torch.empty(Mo, Cout, dtype=torch.half, device=dev)
    part = torch.zeros(Mo // 64, Cout, 2, dtype=torch.float32, device=dev)
    kw = dict(stride=stride, bias=b.to(dev), rowbias=rb.to(dev), rows_per_rb=HW, res=res.to(dev) if with_res else None)
    ops.conv2d(x.to(dev), conv_weight_nhwc(w).to(dev), y, gn_out=part, **kw)
    y0 = torch.empty_like(y)
    ops.conv2d(x.to(dev), conv_weight_nhwc(w).to(dev), y0, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    mu, m2 = _slab_stats(y.cpu())
    np.testing.assert_allclose(part[:, :, 0].cpu().numpy(), mu.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(part[:, :, 1].cpu().numpy(), m2.numpy(), rtol=3e-4, atol=2e-3)
    ws = ops.gn_workspace(dev, R, 32)
    fused, three = torch.empty_like(y), torch.empty_like(y)
    ops.groupnorm(y, fused, R, HW, gam.to(dev), bet.to(dev), 1e-5, silu, 32, ws, partials=part)
    ops.groupnorm(y, three, R, HW, gam.to(dev), bet.to(dev), 1e-5, silu, 32, ws)
    torch.cuda.synchronize()
    yn = y.float().cpu().reshape(R, HW, Cout).permute(0, 2, 1)
    want = F.group_norm(yn, 32, gam.float(), bet.float(), 1e-5)
    want = (F.silu(want) if silu else want).permute(0, 2, 1).reshape(Mo, Cout)
    _close(fused, want, rtol=4e-3, atol=4e-3, what="GroupNorm from producer partials")
    assert (fused.float() - three.float()).abs().max().item() <= 4e-3 * max(1.0, want.abs().max().item())


# ---- round 3: to_q + text / IP cross-attention in one launch (IIR_EPI_XATTN) ------------------------------------------------------
@pytest.mark.parametrize("R,T,heads,K,tk,ti,fold", [(2, 1024, 20, 1280, 77, 64, True), (2, 4096, 10, 640, 77, 64, True), (1, 128, 2, 128, 77, 64, False),
                                                    (3, 64, 4, 192, 13, 4, False), (2, 256, 2, 256, 80, 1, True)])
def test_gemm_cross_attention_epilogue(dev, R, T, heads, K, tk, ti, fold):
    """`attn2.to_q` (LayerNorm folded or not) whose workgroups run the two SDPA calls + add of TA_IPAttnProcessor2_0
    (module/ip_adapter/attention_processor.py:1140,1165,1185,1192) on the q tile they have just finished.  Against (a) the two
    launches it replaces -- the same q projection, then `iir_attention_d64_f16` -- and (b) torch in fp32 on that q."""
    from instantir_amd import ops
    g = torch.Generator().manual_seed(R * T + heads + tk)
    C, M = heads * 64, R * T
    fac = ops.attn_q_factor()
    a = _rand(g, M, K)
    w, b = _rand(g, C, K, scale=K ** -0.5), _rand(g, C, scale=0.3)
    kw = {}
    if fold:                      # the residual stream and its LayerNorm partials, as the launch before `to_q` leaves them
        w0 = _rand(g, K, K, scale=K ** -0.5)
        res = (_rand(g, M, K).float() + 2.0 * torch.randn(M, 1, generator=g)).half()
        gamma, beta = (1.0 + 0.2 * torch.randn(K, generator=g)).half(), (0.3 * torch.randn(K, generator=g)).half()
        parts = ops.ln_parts(M, K, K)
        if parts == 0:
            pytest.skip("producer tile of this shape leaves no LayerNorm partials")
        h = torch.empty(M, K, dtype=torch.half, device=dev)
        stats = torch.zeros(parts, M, 2, dtype=torch.float32, device=dev)
        ops.gemm(a.to(dev), w0.to(dev), h, res=res.to(dev), ln_out=stats)
        f = ops.LnFold((w.float() * fac).to(dev), gamma.to(dev), beta.to(dev), bias=(b.float() * fac).to(dev), eps=1e-5)
        a_dev, w_dev, b_dev = h, f.w, f.bias
        kw["ln_in"] = (stats, f.colsum, f.eps)
    else:
        a_dev, w_dev, b_dev = a.to(dev), (w.float() * fac).half().to(dev), (b.float() * fac).half().to(dev)
    segs, kvh = [], []
    for L in (tk, ti):
        k, v = _rand(g, R, L, C), _rand(g, R, L, C)
        tp = (L + 7) // 8 * 8
        vt = torch.zeros(C, R * tp, dtype=torch.half, device=dev)
        for r in range(R):
            vt[:, r * tp:r * tp + L] = v[r].T.to(dev)
        segs.append((k.reshape(-1, C).to(dev), L, vt, tp, L))
        kvh.append((k, v))
    q = torch.empty(M, C, dtype=torch.half, device=dev)
    ops.gemm(a_dev, w_dev, q, bias=b_dev, **kw)
    two = torch.empty(M, C, dtype=torch.half, device=dev)
    ops.attention(q, two, segs, R, heads, T, q_prescaled=True)
    one = torch.full((M, C), float("nan"), dtype=torch.half, device=dev)
    ops.gemm(a_dev, w_dev, one, bias=b_dev, epi=ops.EPI_XATTN, xattn=(segs, T), **kw)
    torch.cuda.synchronize()
    qf = q.float().cpu().reshape(R, T, heads, 64).transpose(1, 2) * math.log(2.0)          # q carries scale * log2(e)
    want = 0
    for k, v in kvh:
        sp = lambda x: x.float().reshape(R, -1, heads, 64).transpose(1, 2)
        want = want + torch.softmax(qf @ sp(k).transpose(-1, -2), dim=-1) @ sp(v)
    want = want.transpose(1, 2).reshape(M, C)
    assert torch.isfinite(one).all()
    _close(one, want, rtol=3e-3, atol=3e-3, what="fused to_q + cross-attention vs torch")
    _close(one, two.float().cpu(), rtol=3e-3, atol=3e-3, what="fused vs the two launches")
    with pytest.raises(ValueError):
        ops.gemm(a_dev, w_dev, one, bias=b_dev, epi=ops.EPI_XATTN, **kw)
    from instantir_amd.lib import HipLibraryError
    with pytest.raises(HipLibraryError):                # rows of an image must be whole 64-row tiles
        ops.gemm(a_dev, w_dev, one, bias=b_dev, epi=ops.EPI_XATTN, xattn=(segs, T + 8), **kw)


# ---- round 3: both operands in fp8 (iir_gemm_desc.a_fp8) ---------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,mode,tile", [(256, 320, 256, "plain", 0), (2048, 1280, 1280, "res", 0), (2048, 1280, 5120, "plain", 0),
                                             (8192, 640, 640, "res", 0), (512, 2560, 640, "geglu", 0), (300, 160, 384, "plain", 25),
                                             (4096, 1280, 1280, "plain", 21), (2048, 1280, 1280, "plain", 55),
                                             (512, 640, 256, "geglu", 91), (256, 320, 256, "plain", 91), (512, 640, 384, "res", 91),
                                             (4096, 10240, 1280, "geglu", 0)])      # 91 / the last one: the 8-wave 256 x 320 kernel
def test_gemm_fp8_both_operands(dev, M, N, K, mode, tile):
    """A and W as E4M3 bytes, 128 K values per K tile (two fp8 MFMAs on the 16 bytes a lane reads).  Exact reference: the same
    bytes dequantised, multiplied in fp64 -- products of fp8 values are exact in fp32, only the accumulation order differs."""
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    g = torch.Generator().manual_seed(M + N + K)
    a, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    a8, sa = ops.quantize_fp8_tensor(a.to(dev))
    if mode == "geglu":
        w, b = pair_rows(w[:N // 2], w[N // 2:]), pair_rows(b[:N // 2], b[N // 2:])
    q, sc = ops.quantize_fp8_rows(w.to(dev))
    w8 = ops.Fp8Weight(q, sc)
    res = _rand(g, M, N) if mode == "res" else None
    y = (a8.float().double().cpu() * sa) @ (q.float().double().cpu() * sc.double().cpu()[:, None]).T + b.double()
    if mode == "geglu":
        # un-permute: rows [16 blk, 16 blk + 8) are value rows, the next 8 their gates
        yb = y.reshape(M, N // 16, 16)
        want = (yb[:, :, :8] * F.gelu(yb[:, :, 8:])).reshape(M, N // 2)
        out = torch.empty(M, N // 2, dtype=torch.half, device=dev)
        ops.gemm_fp8(a8, w8, out, a_scale=sa, bias=b.to(dev), epi=ops.EPI_GEGLU, tile=tile)
    else:
        want = y + (res.double() if res is not None else 0)
        out = torch.empty(M, N, dtype=torch.half, device=dev)
        ops.gemm_fp8(a8, w8, out, a_scale=sa, bias=b.to(dev), res=res.to(dev) if res is not None else None, tile=tile)
    torch.cuda.synchronize()
    _close(out, want.float(), rtol=2e-3, atol=2e-3, what="fp8 x fp8 gemm")
    with pytest.raises(ValueError):
        ops.gemm_fp8(a8[:, :K - 64], ops.Fp8Weight(q[:, :K - 64].contiguous(), sc), out)       # K % 128


def _fp8_bytes_of(x16):
    """What `iir_fp8x8` stores for fp16 values: E4M3, round to nearest even, saturating."""
    return x16.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn)


def test_producers_store_fp8_operands(dev):
    """LayerNorm, attention and the GEGLU epilogue can store their result as E4M3 bytes (the A operand of the next all-fp8 GEMM):
    exactly the fp16 result of the same launch, converted as the fp8-weight GEMM converts its activation fragments."""
    from instantir_amd import ops
    from instantir_amd.packing import pair_rows
    g = torch.Generator().manual_seed(5)
    M, C = 512, 640
    x = (_rand(g, M, C).float() * 3 + torch.randn(M, 1, generator=g)).half().to(dev)
    gam, bet = (1 + 0.2 * torch.randn(C, generator=g)).half().to(dev), (0.3 * torch.randn(C, generator=g)).half().to(dev)
    y16 = torch.empty(M, C, dtype=torch.half, device=dev)
    y8 = torch.zeros(M, C, dtype=torch.uint8, device=dev)
    ops.layernorm(x, y16, gam, bet, 1e-5)
    ops.layernorm(x, y8, gam, bet, 1e-5)
    assert torch.equal(y8.view(torch.float8_e4m3fn).float(), _fp8_bytes_of(y16).float()), "LayerNorm fp8 store"
    # attention (self, and text + IP), fp8 output
    B, heads, T = 2, 4, 256
    Ch = heads * 64
    q, k, v = _rand(g, B * T, Ch).to(dev), _rand(g, B * T, Ch).to(dev), _rand(g, B, T, Ch)
    vt = torch.cat([v[b].T for b in range(B)], 1).contiguous().to(dev)
    o16 = torch.empty(B * T, Ch, dtype=torch.half, device=dev)
    o8 = torch.zeros(B * T, Ch, dtype=torch.uint8, device=dev)
    for segs in ([(k, T, vt, T, T)], [(k[:2 * 77 * 0 + B * T], T, vt, T, 77), (k, T, vt, T, 64)]):
        ops.attention(q, o16, segs, B, heads, T)
        ops.attention(q, o8, segs, B, heads, T)
        assert torch.equal(o8.view(torch.float8_e4m3fn).float(), _fp8_bytes_of(o16).float()), "attention fp8 store"
    # GEGLU projection with fp8 operands AND an fp8 result; q|k|v with a transposed V third
    K, N = 640, 2560
    a8, sa = ops.quantize_fp8_tensor(_rand(g, M, K).to(dev))
    w, b = _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    wq = ops.Fp8Weight(*ops.quantize_fp8_rows(pair_rows(w[:N // 2], w[N // 2:]).to(dev)))
    bp = pair_rows(b[:N // 2], b[N // 2:]).to(dev)
    f16o = torch.empty(M, N // 2, dtype=torch.half, device=dev)
    f8o = torch.zeros(M, N // 2, dtype=torch.uint8, device=dev)
    ops.gemm_fp8(a8, wq, f16o, a_scale=sa, bias=bp, epi=ops.EPI_GEGLU)
    ops.gemm_fp8(a8, wq, f8o, a_scale=sa, bias=bp, epi=ops.EPI_GEGLU)
    assert torch.equal(f8o.view(torch.float8_e4m3fn).float(), _fp8_bytes_of(f16o).float()), "GEGLU fp8 store"
    w3 = ops.Fp8Weight(*ops.quantize_fp8_rows(_rand(g, 3 * K, K, scale=K ** -0.5).to(dev)))
    full = torch.empty(M, 3 * K, dtype=torch.half, device=dev)
    qk, vtr = torch.empty(M, 2 * K, dtype=torch.half, device=dev), torch.zeros(K, M, dtype=torch.half, device=dev)
    ops.gemm_fp8(a8, w3, full, a_scale=sa)
    ops.gemm_fp8(a8, w3, qk, a_scale=sa, out_t=(vtr, 2 * K))
    torch.cuda.synchronize()
    assert torch.equal(qk, full[:, :2 * K]) and torch.equal(vtr, full[:, 2 * K:].T.contiguous())
    with pytest.raises(ValueError):
        ops.layernorm(x, y8[:, :C - 8], gam, bet, 1e-5)
