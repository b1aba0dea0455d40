"""GPU parity: whole networks through the HIP engine vs the CPU fp32 oracle on the tiny geometry
(same seeded fp16 weights, same inputs)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def psnr(got, want):
    import inspect
    from conftest import record_psnr
    mse = ((got - want) ** 2).mean().item()
    peak = want.abs().max().item()
    v = 10 * math.log10(peak * peak / max(mse, 1e-30))
    record_psnr("engine." + inspect.stack()[1].function, v)
    return v


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from instantir_amd import lib, weights as W
    from instantir_amd.config import UNetConfig
    lib.load()
    cfg = UNetConfig.tiny()
    sd = W.synth_state_dict(W.unet_specs(cfg), 11)
    sda = W.synth_state_dict(W.aggregator_specs(cfg), 12)
    lora = W.synth_state_dict(W.lora_specs(cfg), 13)
    g = torch.Generator().manual_seed(5)
    R, H = 2, 16
    inp = dict(
        R=R, H=H,
        x=torch.randn(R, 4, H, H, generator=g).half(),
        prev=torch.randn(R, 4, H, H, generator=g).half(),
        ctx=torch.randn(R, cfg.text_len, cfg.cross_attention_dim, generator=g).half(),
        pooled=torch.randn(R, cfg.pooled_dim, generator=g).half(),
        tid=torch.tensor([[H * 8.0, H * 8.0, 0, 0, H * 8.0, H * 8.0]] * R),
        img=torch.randn(2, 1, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g).half(),
    )
    return cfg, sd, sda, lora, inp, torch.device("cuda:0")


def _nhwc_pad(x, dev):
    R, C, H, W = x.shape
    out = torch.zeros(R * H * W, 64, dtype=torch.half, device=dev)
    out[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C).to(dev)
    return out


def _to_nchw(y2d, R, H, W):
    C = y2d.shape[1]
    return y2d.float().cpu().reshape(R, H, W, C).permute(0, 3, 1, 2)


def test_resampler_matches_oracle(env):
    from instantir_amd.engine import HipUNet
    from oracle import nets
    cfg, sd, _, _, inp, dev = env
    P = {k: v.float() for k, v in sd.items()}
    want = nets.image_projection(P, [inp["img"].float()], cfg.resampler)[0]
    net = HipUNet(cfg, sd, dev)
    got = net.resampler(inp["img"]).float().cpu()
    torch.cuda.synchronize()
    assert psnr(got, want) > 55, psnr(got, want)


@pytest.mark.parametrize("use_lora", [False, True])
def test_unet_forward_matches_oracle(env, use_lora):
    from instantir_amd.engine import HipUNet
    from oracle import nets
    cfg, sd, sda, lora, inp, dev = env
    R, H = inp["R"], inp["H"]
    P = {k: v.float() for k, v in sd.items()}
    L = None
    if use_lora:
        L = {k: v.float() for k, v in lora.items()}
        L["scaling"] = 2.0
    ip = nets.image_projection(P, [inp["img"].float()], cfg.resampler, L)[0]
    # aggregator residuals as extra inputs (random, small) to exercise the additive-residual path
    g = torch.Generator().manual_seed(1)
    from instantir_amd.weights import skip_channels
    hs = [16, 16, 16, 8, 8, 8, 4, 4, 4]
    down = [(torch.randn(R, c, h, h, generator=g) * 0.3).half() for c, h in zip(skip_channels(cfg), hs)]
    mid = (torch.randn(R, cfg.block_out_channels[-1], 4, 4, generator=g) * 0.3).half()
    scale = torch.tensor([0.75, 1.0])
    want = nets.unet_forward(P, cfg, inp["x"].float(), 499, inp["ctx"].float(), inp["pooled"].float(), inp["tid"], ip,
                             [d.float() * scale[:, None, None, None] for d in down], mid.float() * scale[:, None, None, None], L)
    net = HipUNet(cfg, sd, dev, lora=lora if use_lora else None, lora_scaling=2.0)
    ipd = net.resampler(inp["img"])
    st = net.prepare(inp["ctx"], inp["pooled"], inp["tid"], ipd, H, H)
    t = torch.full((R, 1), 499.0, device=dev)
    to2d = lambda z: z.permute(0, 2, 3, 1).reshape(-1, z.shape[1]).contiguous().to(dev)
    eps = net.forward(_nhwc_pad(inp["x"], dev), t, st, [to2d(d) for d in down], to2d(mid), scale.to(dev))
    torch.cuda.synchronize()
    got = _to_nchw(eps, R, H, H)
    p = psnr(got, want)
    assert torch.isfinite(got).all() and p >= 50, p


def test_aggregator_matches_oracle(env):
    from instantir_amd.engine import HipAggregator
    from oracle import nets
    cfg, _, sda, _, inp, dev = env
    R, H = inp["R"], inp["H"]
    PA = {k: v.float() for k, v in sda.items()}
    wd, wm = nets.aggregator_forward(PA, cfg, inp["x"].float(), 499, inp["prev"].float(), inp["pooled"].float(), inp["tid"])
    agg = HipAggregator(cfg, sda, dev)
    st = agg.prepare(inp["pooled"], inp["tid"], H, H)
    t = torch.full((R, 1), 499.0, device=dev)
    down, mid = agg.forward(_nhwc_pad(inp["x"], dev), _nhwc_pad(inp["prev"], dev), t, st)
    torch.cuda.synchronize()
    for k, (d, w_) in enumerate(zip(down, wd)):
        h = w_.shape[2]
        p = psnr(_to_nchw(d, R, h, w_.shape[3]), w_)
        assert p >= 50, (k, p)
    assert psnr(_to_nchw(mid, R, wm.shape[2], wm.shape[3]), wm) >= 50


def test_merged_lora_equals_side_branch(env):
    """W + s*B*A folded into a second weight copy by the library's own GEMM (engine._merge_lora) == peft's side branch
    (SURVEY 8a row L0), linear and conv (k x k `lora_A`, 1x1 `lora_B`), rank zero-padded to a K tile; fp16 tolerance."""
    import numpy as np
    from instantir_amd.engine import _merge_lora
    from oracle import nets
    dev = env[5]
    g = torch.Generator().manual_seed(0)
    sd = {"lin.weight": torch.randn(24, 16, generator=g).half(), "lin.bias": torch.randn(24, generator=g).half(),
          "cv.weight": torch.randn(12, 8, 3, 3, generator=g).half(), "cv.bias": torch.randn(12, generator=g).half()}
    lora = {"lin.lora_A.weight": torch.randn(4, 16, generator=g).half(), "lin.lora_B.weight": torch.randn(24, 4, generator=g).half(),
            "cv.lora_A.weight": torch.randn(4, 8, 3, 3, generator=g).half(), "cv.lora_B.weight": torch.randn(12, 4, 1, 1, generator=g).half()}
    merged = {k: v.float().cpu() for k, v in _merge_lora({k: v.to(dev) for k, v in sd.items()}, {k: v.to(dev) for k, v in lora.items()}, 0.25).items()}
    sdf = {k: v.float() for k, v in sd.items()}
    lo = dict({k: v.float() for k, v in lora.items()}, scaling=0.25)
    x = torch.randn(5, 16, generator=g)
    np.testing.assert_allclose(nets.linear(merged, "lin", x).numpy(), nets.linear(sdf, "lin", x, lo).numpy(), rtol=4e-3, atol=2e-2)
    im = torch.randn(2, 8, 7, 7, generator=g)
    for stride in (1, 2):
        np.testing.assert_allclose(nets.conv2d(merged, "cv", im, stride=stride).numpy(),
                                   nets.conv2d(sdf, "cv", im, stride=stride, lora=lo).numpy(), rtol=4e-3, atol=4e-2)
