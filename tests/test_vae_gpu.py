"""GPU parity: VAE decode / encode through the HIP engine vs the CPU fp32 oracle (tiny geometry), plus the
two kernels added for it (row softmax, bottom/right-padded stride-2 conv)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def psnr(got, want):
    mse = ((got - want) ** 2).mean().item()
    return 10 * math.log10(want.abs().max().item() ** 2 / max(mse, 1e-30))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from instantir_amd import lib
    lib.load()
    return torch.device("cuda:0")


def test_softmax_rows(dev):
    from instantir_amd import ops
    g = torch.Generator().manual_seed(0)
    for rows, cols in [(7, 64), (33, 1000), (4, 16384)]:
        x = (torch.randn(rows, cols, generator=g) * 3).half()
        want = torch.softmax(x.float(), dim=-1)
        buf = torch.zeros(rows, cols + 8, dtype=torch.half, device=dev)
        buf[:, :cols] = x.to(dev)
        ops.softmax_rows(buf[:, :cols])
        torch.cuda.synchronize()
        got = buf[:, :cols].float().cpu()
        assert (got - want).abs().max().item() < 2e-3 * want.max().item() + 1e-6
        assert buf[:, cols:].abs().max().item() == 0


def test_conv_asymmetric_pad_stride2(dev):
    """F.pad(x, (0,1,0,1)) + conv3x3 stride 2 padding 0 (module/diffusers_vae/vae.py:110)."""
    from instantir_amd import ops
    from instantir_amd.packing import conv_weight_nhwc
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 16, 12, generator=g).half()
    w = (torch.randn(128, 64, 3, 3, generator=g) / 24).half()
    b = torch.randn(128, generator=g).half()
    want = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2).permute(0, 2, 3, 1).reshape(-1, 128)
    out = torch.empty(2 * 8 * 6, 128, dtype=torch.half, device=dev)
    ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dev), conv_weight_nhwc(w).to(dev), out, stride=2, pad_mode=1, bias=b.to(dev))
    torch.cuda.synchronize()
    assert (out.float().cpu() - want).abs().max().item() < 5e-3


@pytest.fixture(scope="module")
def vae_env(dev):
    from instantir_amd import weights as W
    from instantir_amd.config import VAEConfig
    from instantir_amd.vae import HipVAE
    vc = VAEConfig.tiny()
    sd = W.synth_state_dict(W.vae_decoder_specs(vc) + W.vae_encoder_specs(vc), 21)
    return vc, sd, HipVAE(vc, sd, dev)


def test_vae_decode_matches_oracle(vae_env):
    from oracle import vae as OV
    vc, sd, hv = vae_env
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 4, 8, 8, generator=g)
    want = OV.decode({k: v.float() for k, v in sd.items()}, vc, z)
    got = hv.decode(z).cpu()
    assert got.shape == want.shape == (2, 3, 64, 64)
    p = psnr(got, want)
    assert torch.isfinite(got).all() and p > 40, p
    img = hv.decode_latent(z * vc.scaling_factor, "pt")
    assert img.min().item() >= 0 and img.max().item() <= 1
    assert len(hv.decode_latent(z * vc.scaling_factor, "pil")) == 2


def test_vae_encode_matches_oracle(vae_env):
    from oracle import vae as OV
    vc, sd, hv = vae_env
    g = torch.Generator().manual_seed(4)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    eps = torch.randn(2, 4, 8, 8, generator=g)
    want = OV.encode({k: v.float() for k, v in sd.items()}, vc, img, eps)
    got = hv.encode(img, eps).cpu()
    p = psnr(got, want)
    assert torch.isfinite(got).all() and p > 40, p


def test_pipeline_pixels_in_pixels_out(vae_env, dev):
    """Whole restoration call on pixel inputs: VAE encode -> 3-step CFG loop -> VAE decode ('pt'), against the
    oracle composition of the same stages (pipelines/sdxl_instantir.py:1370-1379, :1497-1660, :1668-1704)."""
    from instantir_amd import weights as W
    from instantir_amd.config import UNetConfig
    from instantir_amd.pipeline import InstantIRPipeline
    from instantir_amd.schedulers import DDIMScheduler, LCMSingleStepScheduler
    from oracle import pipeline as OP, vae as OV
    vc, vsd, hv = vae_env
    cfg = UNetConfig.tiny()
    sd = W.synth_state_dict(W.unet_specs(cfg), 11)
    sda = W.synth_state_dict(W.aggregator_specs(cfg), 12)
    lora = W.synth_state_dict(W.lora_specs(cfg), 13)
    g = torch.Generator().manual_seed(9)
    B = 1
    img01 = torch.rand(B, 3, 128, 128, generator=g)
    pe = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g).half().float()
    pooled = torch.randn(B, cfg.pooled_dim, generator=g).half().float()
    feats = torch.randn(2, B, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g).half().float()
    eps = torch.randn(B, 4, 16, 16, generator=g)
    noise = torch.randn(B, 4, 16, 16, generator=g)
    pipe = InstantIRPipeline(cfg, sd, scheduler=DDIMScheduler(), vae=hv, device=dev)
    pipe.aggregator.load_state_dict(sda)
    pipe.prepare_previewers(lora, lora_alpha=8)
    got = pipe(image=img01, prompt_embeds=pe, pooled_prompt_embeds=pooled, ip_adapter_image_embeds=[feats], output_type="pt",
               num_inference_steps=3, guidance_scale=5.0, init_noise=noise, vae_noise=eps,
               previewer_scheduler=LCMSingleStepScheduler.from_config(pipe.scheduler.config)).images.float().cpu()
    PV = {k: v.float() for k, v in vsd.items()}
    lq = OV.encode(PV, vc, img01 * 2 - 1, eps) * vc.scaling_factor
    L = {k: v.float() for k, v in lora.items()}
    L["scaling"] = 8.0 / cfg.lora_rank
    lat = OP.denoise({k: v.float() for k, v in sd.items()}, {k: v.float() for k, v in sda.items()}, L, cfg, lq, pe, pooled, feats,
                     init_noise=noise, num_inference_steps=3, guidance_scale=5.0, sampler="ddim")
    want = (OV.decode(PV, vc, lat / vc.scaling_factor) / 2 + 0.5).clamp(0, 1)
    assert got.shape == (B, 3, 128, 128)
    mse = ((got - want) ** 2).mean().item()
    p = 10 * math.log10(1.0 / max(mse, 1e-30))
    assert p > 35, p


def test_vae_tiled_decode_matches_oracle(vae_env):
    """Tiled decode with seam blending (BASELINE configs[3] path): 3 x 3 overlapping tiles at the tiny geometry
    (tile 16 latent px = 128 px, stride 12, blend 32 px, crop 96 px; ragged 8-px last row / column)."""
    from oracle import vae as OV
    vc, sd, hv = vae_env
    g = torch.Generator().manual_seed(6)
    z = torch.randn(1, 4, 32, 32, generator=g)
    want = OV.decode_tiled({k: v.float() for k, v in sd.items()}, vc, z, sample_size=128)
    got = hv.decode_tiled(z, sample_size=128).cpu()
    assert got.shape == want.shape
    p = psnr(got, want)
    assert torch.isfinite(got).all() and p > 40, p
