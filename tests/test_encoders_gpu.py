"""GPU parity of the DINOv2 image encoder (HIP kernels) against transformers' Dinov2Model with the same random
weights -- the class the reference instantiates (module/ip_adapter/utils.py:106-118).  transformers here is 5.x, not
the pinned 4.36.2: a secondary oracle for architecture/arithmetic, not a pinned fixture."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def psnr(got, want):
    mse = ((got - want) ** 2).mean().item()
    return 10 * math.log10(want.abs().max().item() ** 2 / max(mse, 1e-30))


def _model(hidden, heads, layers, image_size, seed):
    from transformers import Dinov2Config, Dinov2Model
    torch.manual_seed(seed)
    cfg = Dinov2Config(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, patch_size=14,
                       image_size=image_size, mlp_ratio=4)
    m = Dinov2Model(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lambda1" in n:
                p.copy_(0.5 + 0.5 * torch.rand_like(p))          # LayerScale: make it matter
            elif p.ndim == 1 and "norm" in n:
                p.add_(0.05 * torch.randn_like(p))
            elif "position_embeddings" in n or "cls_token" in n:
                p.copy_(0.3 * torch.randn_like(p))
        sd = {k: v.half().float() for k, v in m.state_dict().items()}    # both sides use fp16-representable weights
        m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("size", [56, 84])
def test_dinov2_small_matches_transformers(size):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from instantir_amd.encoders import HipDinov2
    m, sd = _model(128, 2, 3, 56, 0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, size, size, generator=g)
    with torch.no_grad():
        want = m(pixel_values=x).last_hidden_state
    enc = HipDinov2(sd, "cuda:0")
    got = enc(x).float().cpu()
    assert got.shape == want.shape
    p = psnr(got, want)
    assert p > 45, p
    f, z = enc.encode_image_pair(x)
    with torch.no_grad():
        wz = m(pixel_values=torch.zeros_like(x)).last_hidden_state
    assert psnr(z.float().cpu(), wz) > 45


def test_dinov2_vit_l_224_shapes():
    """ViT-L/14 geometry of facebook/dinov2-large at 224 px: (B, 257, 1024), 304 M parameters."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from instantir_amd.encoders import HipDinov2
    m, sd = _model(1024, 16, 24, 518, 2)
    assert 300e6 < sum(p.numel() for p in m.parameters()) < 310e6
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = m(pixel_values=x).last_hidden_state
    got = HipDinov2(sd, "cuda:0")(x).float().cpu()
    assert got.shape == (1, 257, 1024)
    p = psnr(got, want)
    assert p > 40, p


@pytest.mark.parametrize("act,eos", [("quick_gelu", 2), ("gelu", 98)])
def test_clip_text_matches_transformers(act, eos):
    """CLIP text encoders of encode_prompt (pipelines/sdxl_instantir.py:516-560): hidden_states[-2] and text_embeds."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    from instantir_amd.encoders import HipCLIPText
    torch.manual_seed(5)
    c = CLIPTextConfig(vocab_size=100, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                       max_position_embeddings=77, projection_dim=96, hidden_act=act, eos_token_id=eos, bos_token_id=0, pad_token_id=1)
    m = CLIPTextModelWithProjection(c).eval()
    sd = {k: v.half().float() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    ids = torch.randint(3, 97, (2, 77))
    ids[0, 20:] = eos if eos != 2 else 99      # first sequence ends early (eot = highest id / first eos)
    ids[1, 76] = eos if eos != 2 else 99
    with torch.no_grad():
        out = m(input_ids=ids, output_hidden_states=True)
    enc = HipCLIPText(sd, "cuda:0", hidden_act=act, eos_token_id=eos)
    h, pooled = enc(ids)
    assert h.shape == (2, 77, 128) and pooled.shape == (2, 96)
    assert psnr(h.float().cpu(), out.hidden_states[-2]) > 45
    assert psnr(pooled.float().cpu(), out.text_embeds) > 40
    h1, pooled1 = enc(ids, clip_skip=1)                         # hidden_states[-(clip_skip + 2)], pooled unchanged (:524-530)
    assert psnr(h1.float().cpu(), out.hidden_states[-3]) > 45 and torch.equal(pooled1, pooled)
    with pytest.raises(ValueError):
        enc(ids, clip_skip=3)


def test_pipeline_with_encoders_attached():
    """`pipe(prompt_ids=..., ip_adapter_image=...)` (encoders run inside the call, pipelines/sdxl_instantir.py:1325-1357)
    equals `pipe(prompt_embeds=..., ip_adapter_image_embeds=...)` fed with the same encoders' outputs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    from instantir_amd import weights as W
    from instantir_amd.config import UNetConfig
    from instantir_amd.encoders import HipCLIPText, HipDinov2
    from instantir_amd.pipeline import InstantIRPipeline
    from instantir_amd.schedulers import DDIMScheduler, LCMSingleStepScheduler
    cfg = UNetConfig.tiny()
    torch.manual_seed(7)
    mk = lambda: {k: v.half().float() for k, v in CLIPTextModelWithProjection(CLIPTextConfig(
        vocab_size=100, hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=1,
        max_position_embeddings=77, projection_dim=cfg.pooled_dim, eos_token_id=99, bos_token_id=0, pad_token_id=1)).state_dict().items()}
    te1, te2 = HipCLIPText(mk(), "cuda:0", eos_token_id=99), HipCLIPText(mk(), "cuda:0", eos_token_id=99)
    dino_m, dino_sd = _model(64, 1, 2, 56, 8)
    dino = HipDinov2(dino_sd, "cuda:0")
    sd = W.synth_state_dict(W.unet_specs(cfg), 11)
    sda = W.synth_state_dict(W.aggregator_specs(cfg), 12)
    lora = W.synth_state_dict(W.lora_specs(cfg), 13)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(2, 98, (1, cfg.text_len), generator=g)
    ids[0, -1] = 99
    img = torch.randn(1, 3, 56, 56, generator=g)
    lq, noise = torch.randn(1, 4, 16, 16, generator=g) * 0.8, torch.randn(1, 4, 16, 16, generator=g)

    def make(**kw):
        pipe = InstantIRPipeline(cfg, sd, scheduler=DDIMScheduler(), device="cuda:0", **kw)
        pipe.aggregator.load_state_dict(sda)
        pipe.prepare_previewers(lora, lora_alpha=8)
        return pipe

    common = dict(image=lq, output_type="latent", num_inference_steps=2, guidance_scale=5.0, init_noise=noise)
    p1 = make(image_encoder=dino, text_encoder=te1, text_encoder_2=te2)
    a = p1(prompt_ids=ids, ip_adapter_image=img, previewer_scheduler=LCMSingleStepScheduler.from_config(p1.scheduler.config),
           **common).images
    pe, npe, pooled, npooled = p1.encode_prompt(prompt_ids=ids, do_cfg=True)
    assert pe.shape == (1, cfg.text_len, cfg.cross_attention_dim) and pooled.shape == (1, cfg.pooled_dim)
    assert npe.abs().max().item() == 0                       # force_zeros_for_empty_prompt
    feats = p1.prepare_ip_adapter_image_embeds(img, True)
    assert feats[0].shape == (2, 1, 17, 64)
    p2 = make()
    b = p2(prompt_embeds=pe, pooled_prompt_embeds=pooled, ip_adapter_image_embeds=feats,
           previewer_scheduler=LCMSingleStepScheduler.from_config(p2.scheduler.config), **common).images
    assert torch.isfinite(a).all() and torch.equal(a, b)
    # the text encodings are kept per token-id pair (a batch job uses one prompt): same ids -> no second CLIP pass, same
    # tensors; other ids or another encoder object -> a fresh pass
    calls = []
    real = te1.__class__.__call__
    te1.__class__.__call__ = lambda self_, *aa, **kk: (calls.append(1), real(self_, *aa, **kk))[1]
    try:
        pe2, _, pooled2, _ = p1.encode_prompt(prompt_ids=ids, do_cfg=True)
        assert not calls and torch.equal(pe2, pe) and torch.equal(pooled2, pooled)
        ids_b = ids.clone()
        ids_b[0, 1] = (int(ids[0, 1]) + 1) % 90 + 2
        pe3, _, _, _ = p1.encode_prompt(prompt_ids=ids_b, do_cfg=True)
        assert len(calls) == 2 and not torch.equal(pe3, pe)           # both encoders ran once
        p1.text_encoder = HipCLIPText(mk(), "cuda:0", eos_token_id=99)
        pe4, _, _, _ = p1.encode_prompt(prompt_ids=ids, do_cfg=True)
        assert len(calls) == 4 and not torch.equal(pe4, pe)
    finally:
        te1.__class__.__call__ = real


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_clip_vision_matches_transformers(act):
    """`use_clip_encoder` branch (module/ip_adapter/utils.py:106-118): `hidden_states[-2]` of the image and of a zero image
    (pipelines/sdxl_instantir.py:644-654, what a Resampler projector is fed) and `image_embeds` (:656-659), against
    transformers' CLIPVisionModelWithProjection with the same random weights (transformers 5.x here, not the pinned 4.36.2)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from instantir_amd.encoders import HipCLIPVision
    torch.manual_seed(7)
    c = CLIPVisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2, image_size=56,
                         patch_size=14, projection_dim=96, hidden_act=act)
    m = CLIPVisionModelWithProjection(c).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.add_(0.05 * torch.randn_like(p))
            elif "position_embedding" in n or "class_embedding" in n:
                p.copy_(0.3 * torch.randn_like(p))
    sd = {k: v.half().float() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        out = m(pixel_values=x, output_hidden_states=True)
        outz = m(pixel_values=torch.zeros_like(x), output_hidden_states=True)
    enc = HipCLIPVision(sd, "cuda:0", hidden_act=act)
    h, emb = enc(x, with_embeds=True)
    assert h.shape == (2, 17, 128) and emb.shape == (2, 96)
    assert psnr(h.float().cpu(), out.hidden_states[-2]) > 45
    assert psnr(emb.float().cpu(), out.image_embeds) > 40
    f, z = enc.encode_image_pair(x)
    assert psnr(f.float().cpu(), out.hidden_states[-2]) > 45 and psnr(z.float().cpu(), outz.hidden_states[-2]) > 45
    with pytest.raises(ValueError):          # fixed position table
        enc(torch.zeros(1, 3, 70, 70))
    with pytest.raises(ValueError):          # head_dim 80 towers (ViT-H/14) are refused with the reason
        HipCLIPVision(sd, "cuda:0", num_heads=1)
