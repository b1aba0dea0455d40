"""The web demo's request handler without the web UI (gradio_demo/app.py:110-156): slider conversion,
hand-built timesteps, adapter switching, preview captions."""
import pytest
import torch


def test_slider_conversion_and_timesteps():
    from instantir_amd.demo import demo_timesteps, slider_to_fraction
    assert slider_to_fraction(30, 30) == 1.0 and slider_to_fraction(21, 30) == 0.7         # ints are step counts (:121-122)
    assert slider_to_fraction(15.0, 30) == 0.5                                            # floats above 1 too (:123-124)
    assert slider_to_fraction(0.7, 30) == 0.7 and slider_to_fraction(1.0, 30) == 1.0      # fractions pass through
    assert slider_to_fraction(0, 30) == 0.0
    ts = demo_timesteps(30, 1)                                                            # :131-134
    assert ts[0] == 29 * 33 + 1 and ts[-1] == 1 and len(ts) == 30 and all(a - b == 33 for a, b in zip(ts, ts[1:]))
    assert demo_timesteps(7, 0) == [852, 710, 568, 426, 284, 142, 0]


def test_demo_resize():
    from PIL import Image
    from instantir_amd.demo import demo_resize
    im = Image.new("RGB", (640, 400))
    assert demo_resize(im, size=(512, 384)).size == (512, 384)                            # explicit size wins (:20-21)
    assert demo_resize(im).size == (1280, 768)                                            # long side 1280, floor to 64 (:25-28)


@pytest.mark.gpu
def test_instantir_restore_equals_the_direct_call():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    from PIL import Image
    from instantir_amd import weights as W
    from instantir_amd.config import UNetConfig, VAEConfig
    from instantir_amd.demo import instantir_restore
    from instantir_amd.pipeline import InstantIRPipeline
    from instantir_amd.schedulers import DDPMScheduler, LCMSingleStepScheduler
    from instantir_amd.vae import HipVAE
    cfg, vc, dev = UNetConfig.tiny(), VAEConfig.tiny(), "cuda:0"
    vae = HipVAE(vc, W.synth_state_dict(W.vae_decoder_specs(vc) + W.vae_encoder_specs(vc), 21), dev)
    pipe = InstantIRPipeline(cfg, W.synth_state_dict(W.unet_specs(cfg), 11), scheduler=DDPMScheduler(), vae=vae, device=dev)
    pipe.aggregator.load_state_dict(W.synth_state_dict(W.aggregator_specs(cfg), 12))
    pipe.prepare_previewers(W.synth_state_dict(W.lora_specs(cfg), 13), lora_alpha=16)
    pipe.prepare_previewers(W.synth_state_dict(W.lora_specs(cfg, W.LCM_LORA_MODULES), 14), use_lcm=True, lora_alpha=8)
    lcm = LCMSingleStepScheduler.from_config(pipe.scheduler.config)
    g = torch.Generator().manual_seed(3)
    emb = dict(prompt_embeds=torch.randn(1, cfg.text_len, cfg.cross_attention_dim, generator=g),
               pooled_prompt_embeds=torch.randn(1, cfg.pooled_dim, generator=g),
               negative_prompt_embeds=torch.randn(1, cfg.text_len, cfg.cross_attention_dim, generator=g),
               negative_pooled_prompt_embeds=torch.randn(1, cfg.pooled_dim, generator=g),
               ip_adapter_image_embeds=[torch.randn(2, 1, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)],
               vae_noise=torch.randn(1, 4, 16, 16, generator=g))   # `latent_dist.sample()` draws from the GLOBAL RNG in the reference (:1375), not `generator`: pinned here
    lq = Image.fromarray((np.random.default_rng(0).random((96, 160, 3)) * 255).astype(np.uint8))
    assert pipe.unet.active_adapters() == ["lcm"]
    img, row = instantir_restore(pipe, lcm, lq, steps=3, cfg_scale=5.0, guidance_end=2, seed=5, height=128, width=128, **emb)
    assert pipe.unet.active_adapters() == ["previewer"]                                   # :118-120
    # guidance_end = 2 of 3 steps: the last step runs without previewer / Aggregator, so two previews; captions per :154-155
    assert len(row) == 2 and all(r[-1] == f"preview_{i}" for i, r in enumerate(row))
    from instantir_amd.demo import demo_resize
    want = pipe(image=[demo_resize(lq, size=(128, 128))], num_inference_steps=3, guidance_scale=5.0, timesteps=[667, 334, 1],
                control_guidance_end=2 / 3, generator=torch.Generator(device=dev).manual_seed(5), previewer_scheduler=lcm, **emb).images
    assert img.size == (128, 128) and np.array_equal(np.asarray(img), np.asarray(want[0]))
    img2, _ = instantir_restore(pipe, lcm, lq, steps=3, cfg_scale=5.0, guidance_end=2, seed=5, height=128, width=128,
                                creative_restoration=True, **emb)
    assert pipe.unet.active_adapters() == ["lcm"] and not np.array_equal(np.asarray(img2), np.asarray(img))
