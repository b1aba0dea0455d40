"""CPU: host-side logic of the product (schedulers, gates, packing, inventories, LoRA merge, input
checks, sharding) against the oracle / reference-generated goldens."""
import os

import numpy as np
import pytest
import torch

from instantir_amd import weights as W
from instantir_amd.config import UNetConfig
from instantir_amd.schedulers import DDIMScheduler, DDPMScheduler, LCMSingleStepScheduler
from oracle import nets, sched


def test_timetables_bit_identical_int64():
    for cls in (DDPMScheduler, DDIMScheduler):
        for n in (4, 30, 50):
            s = cls()
            s.set_timesteps(n)
            assert s.timesteps.dtype == torch.int64
            assert np.array_equal(s.timesteps.numpy(), sched.leading_timesteps(n))
    s = DDPMScheduler()
    s.set_timesteps(30)
    assert s.timesteps[:2].tolist() == [958, 925] and s.timesteps[-2:].tolist() == [34, 1]


def test_lcm_scheduler_matches_reference_goldens(golden_dir):
    z = np.load(os.path.join(golden_dir, "lcm_scheduler.npz"))
    main = DDPMScheduler()
    s = LCMSingleStepScheduler.from_config(main.config)      # infer.py:138
    assert np.array_equal(s.alphas_cumprod.numpy(), z["alphas_cumprod"])
    for n in (1, 2, 4, 8):
        s.set_timesteps(n)
        assert s.timesteps.dtype == torch.int64 and np.array_equal(s.timesteps.numpy(), z[f"set_timesteps_{n}"])
    cs, co = s.get_scalings_for_boundary_condition_discrete(torch.from_numpy(z["t"]))
    assert np.array_equal(cs.numpy(), z["c_skip"]) and np.array_equal(co.numpy(), z["c_out"])
    # preview coefficients reproduce the reference step when applied in the kernel's op order
    x, e = torch.from_numpy(z["x"]), torch.from_numpy(z["eps"])
    for i, t in enumerate(z["t"]):
        sb, sa, c_out, c_skip = [torch.tensor(v, dtype=torch.float32) for v in s.preview_coefficients(int(t))]
        x0 = (x - sb * e) / sa
        got = c_out * x0 + c_skip * x
        np.testing.assert_allclose(got.numpy(), z["step"][i], rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        s.set_timesteps(51)                       # > original_inference_steps (schedulers/...:380-385)
    with pytest.raises(ValueError):
        s.set_timesteps(timesteps=[10, 20])       # not descending (:346-349)
    with pytest.raises(ValueError):
        s.set_timesteps(4, timesteps=[3, 2])


@pytest.mark.parametrize("n", [4, 30])
def test_step_coefficients_match_oracle_steps(n):
    acp = sched.make_alphas_cumprod()
    g = torch.Generator().manual_seed(1)
    x, e, nz = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))

    def apply(c, noise):
        sb, sa, k0, k1, k2, k3 = [torch.tensor(v, dtype=torch.float32) for v in c[1:7]]
        x0 = (x - sb * e) / sa
        pv = k0 * x0 + k1 * x
        if c[5] != 0:
            pv = pv + k2 * e
        if c[6] != 0:
            pv = pv + k3 * noise
        return pv, x0

    for cls, ofn in ((DDPMScheduler, sched.ddpm_step), (DDIMScheduler, sched.ddim_step)):
        s = cls()
        s.set_timesteps(n)
        for t in s.timesteps.tolist():
            got, got0 = apply(s.step_coefficients(t), nz)
            want, want0 = ofn(acp, e, t, x, n, noise=nz)
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(got0.numpy(), want0.numpy(), rtol=1e-6, atol=1e-6)
    assert DDPMScheduler().step_coefficients(0)[6] == 0.0 if True else None


def test_gating_tables_python_double_semantics():
    """pipelines/sdxl_instantir.py:1415-1421 -- e.g. preview_start=0.2, N=30 skips the first 6 previews."""
    keep, prev = sched.gating_tables(30, 0.0, 1.0, 0.2, 1.0)
    assert keep == [1.0] * 30 and prev == [0.0] * 6 + [1.0] * 24
    keep, _ = sched.gating_tables(50, 0.0, 0.7)
    assert keep == [1.0] * 35 + [0.0] * 15


def test_inventories_have_sdxl_parameter_counts():
    cfg = UNetConfig.sdxl()
    specs = W.unet_specs(cfg)
    count = lambda pred: sum(int(np.prod(s)) for n, s, _ in specs if pred(n))
    base = count(lambda n: "processor" not in n and not n.startswith("encoder_hid_proj"))
    assert base == 2_567_463_684                                     # SDXL-base UNet
    assert round(count(lambda n: n.startswith("encoder_hid_proj")) / 1e6, 2) == 82.70   # Resampler, SURVEY 8c "82.70 M"
    adapter = count(lambda n: "processor" in n)
    assert 760e6 < adapter < 775e6                                   # SURVEY 8a row A3 "767 M"
    agg = sum(int(np.prod(s)) for _, s, _ in W.aggregator_specs(cfg))
    assert 0.95e9 < agg < 1.05e9                                     # SURVEY 8d "1.00 B"
    names = [n for n, _, _ in specs]
    assert len(names) == len(set(names))
    assert sum(n.endswith("attn2.processor.to_k_ip.weight") for n in names) == 70      # Appendix A: 70 processors


def test_lora_targets_follow_peft_suffix_rule():
    t = W.lora_target
    assert t("down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_k")
    assert not t("down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k")        # attn2 text k/v not targeted
    assert t("down_blocks.1.attentions.0.transformer_blocks.0.attn2.processor.to_k_ip")
    assert t("encoder_hid_proj.image_projection_layers.0.layers.0.0.to_kv")
    assert t("encoder_hid_proj.image_projection_layers.0.layers.0.0.to_out")
    assert not t("encoder_hid_proj.image_projection_layers.0.layers.0.1.1")
    assert not t("conv_in") and not t("time_embedding.linear_1")
    assert t("up_blocks.0.upsamplers.0.conv") and t("mid_block.resnets.0.conv_shortcut") is True


def test_lora_merge_has_no_cpu_path():
    """The rank-r product of the LoRA merge runs on the HIP library's GEMM (tests/test_engine_gpu.py checks its algebra
    against peft's side branch); handed CPU tensors it fails loudly instead of falling back to torch."""
    from instantir_amd.engine import _merge_lora
    sd = {"lin.weight": torch.randn(24, 16)}
    lora = {"lin.lora_A.weight": torch.randn(4, 16), "lin.lora_B.weight": torch.randn(24, 4)}
    with pytest.raises(RuntimeError):
        _merge_lora(sd, lora, 0.25)


def test_pair_rows_layout():
    from instantir_amd.packing import conv_weight_nhwc, pair_rows
    v = torch.arange(32).float()[:, None].repeat(1, 3)
    g = v + 100
    p = pair_rows(v, g)
    assert p.shape == (64, 3)
    assert p[:8, 0].tolist() == list(range(8)) and p[8:16, 0].tolist() == [100 + i for i in range(8)]
    assert p[16:24, 0].tolist() == list(range(8, 16))
    w = torch.randn(5, 4, 3, 3)
    q = conv_weight_nhwc(w, 64)
    assert q.shape == (5, 3, 3, 64) and torch.equal(q[..., :4], w.permute(0, 2, 3, 1)) and q[..., 4:].abs().max() == 0


def test_pipeline_input_checks_raise_like_the_reference():
    """pipelines/sdxl_instantir.py:749-864 conditions that apply to tensor inputs (no GPU needed)."""
    from instantir_amd.pipeline import InstantIRPipeline
    cfg = UNetConfig.tiny()
    pipe = InstantIRPipeline(cfg, {}, device="cpu")
    pe, pooled = torch.zeros(1, cfg.text_len, cfg.cross_attention_dim), torch.zeros(1, cfg.pooled_dim)
    base = dict(prompt=None, prompt_embeds=pe, negative_prompt_embeds=None, pooled_prompt_embeds=pooled,
                negative_pooled_prompt_embeds=None, ip_adapter_image=None, ip_adapter_image_embeds=[torch.zeros(2, 1, 3, 4)],
                control_guidance_start=0.0, control_guidance_end=1.0, callback_on_step_end_tensor_inputs=["latents"])
    pipe.check_inputs(**base)
    for bad in (dict(prompt="a photo"), dict(prompt_embeds=None), dict(pooled_prompt_embeds=None),
                dict(control_guidance_start=0.5, control_guidance_end=0.5), dict(control_guidance_end=1.5),
                dict(control_guidance_start=-0.1), dict(ip_adapter_image_embeds=torch.zeros(2, 3)),
                dict(ip_adapter_image=object()), dict(callback_on_step_end_tensor_inputs=["nope"]),
                dict(negative_prompt_embeds=torch.zeros(1, 5, 5)), dict(ip_adapter_image_embeds=[torch.zeros(4)])):
        with pytest.raises((ValueError, NotImplementedError)):
            pipe.check_inputs(**dict(base, **bad))
    with pytest.raises(ValueError):                     # unexpected LoRA key (pipelines/sdxl_instantir.py:390-394)
        InstantIRPipeline(cfg, {"conv_in.weight": torch.zeros(1)}, device="cpu").prepare_previewers(
            {"conv_in.lora_A.weight": torch.zeros(1)})
    with pytest.raises(NotImplementedError):
        pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, image=torch.zeros(1, 4, 8, 8), multistep_restore=True)
    with pytest.raises(ValueError):                     # keys the TA-IP processors do not take are refused, never silently ignored
        pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, image=torch.zeros(1, 4, 8, 8), cross_attention_kwargs={"gligen": {}})


def test_aggregator_state_dict_is_strict():
    from instantir_amd.pipeline import InstantIRPipeline
    cfg = UNetConfig.tiny()
    pipe = InstantIRPipeline(cfg, {}, device="cpu")
    sd = {n: torch.zeros(1) for n, _, _ in W.aggregator_specs(cfg)}
    pipe.aggregator.load_state_dict(sd)
    sd.pop("ref_conv_in.weight")
    with pytest.raises(RuntimeError):
        pipe.aggregator.load_state_dict(sd)
    assert not any("attn2" in n or "norm2." in n and "resnets" not in n for n in sd)      # remove_attn2: no attn2/norm2 keys


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from instantir_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.HipLibraryError):
        lib.load()


def test_oracle_lora_zero_b_is_noop_and_residuals_are_additive():
    cfg = UNetConfig.tiny()
    P = {k: v.float() for k, v in W.synth_state_dict(W.unet_specs(cfg), 1).items()}
    L = {k: (torch.zeros_like(v.float()) if "lora_B" in k else v.float()) for k, v in W.synth_state_dict(W.lora_specs(cfg), 2).items()}
    L["scaling"] = 1.0
    g = torch.Generator().manual_seed(0)
    R, H = 1, 8
    x = torch.randn(R, 4, H, H, generator=g)
    ctx = torch.randn(R, cfg.text_len, cfg.cross_attention_dim, generator=g)
    pooled = torch.randn(R, cfg.pooled_dim, generator=g)
    tid = torch.tensor([[64.0, 64, 0, 0, 64, 64]])
    ip = torch.randn(R, cfg.num_ip_tokens, cfg.cross_attention_dim, generator=g)
    y0 = nets.unet_forward(P, cfg, x, 10, ctx, pooled, tid, ip)
    y1 = nets.unet_forward(P, cfg, x, 10, ctx, pooled, tid, ip, lora=L)
    assert torch.isfinite(y0).all() and torch.equal(y0, y1)
    hs = [8, 8, 8, 4, 4, 4, 2, 2, 2]
    zeros = [torch.zeros(R, c, h, h) for c, h in zip(W.skip_channels(cfg), hs)]
    y2 = nets.unet_forward(P, cfg, x, 10, ctx, pooled, tid, ip, zeros, torch.zeros(R, cfg.block_out_channels[-1], 2, 2))
    assert torch.equal(y0, y2)


def test_copy_state_refreshes_in_place_and_reports_mismatches():
    """`pipeline._copy_state`: a later call's hoisted state is copied INTO the tensors the step graphs were captured on; the
    device pointer table (`ada_jobs`) is left alone; any difference in structure, shape or dtype is reported, not papered over."""
    from instantir_amd.pipeline import _copy_state
    mk = lambda v: {"R": 2, "H": 4, "aug_emb": torch.full((2, 3), v), "ada_jobs": torch.tensor([int(v)]),
                    "kv": {"b0": {"tk": torch.full((4, 2), v, dtype=torch.float16), "tpad": 8}}}
    dst, src = mk(1.0), mk(5.0)
    keep = (dst["aug_emb"], dst["kv"]["b0"]["tk"])
    assert _copy_state(dst, src)
    assert dst["aug_emb"] is keep[0] and dst["kv"]["b0"]["tk"] is keep[1]                 # same storage ...
    assert float(keep[0][0, 0]) == 5.0 and float(keep[1][0, 0]) == 5.0                    # ... new values
    assert int(dst["ada_jobs"][0]) == 1                                                   # pointer table untouched
    assert _copy_state(None, None) and not _copy_state(dst, None) and not _copy_state(None, src)
    for mutate in (lambda s: s.update(R=3), lambda s: s["kv"]["b0"].update(tpad=16), lambda s: s["kv"].update(b1={}),
                   lambda s: s.update(aug_emb=torch.zeros(2, 4)), lambda s: s["kv"]["b0"].update(tk=torch.zeros(4, 2)),
                   lambda s: s.pop("H")):
        other = mk(2.0)
        mutate(other)
        assert not _copy_state(mk(1.0), other)


def test_unet_config_from_a_diffusers_config_mapping():
    """`loaders.unet_config_from_dict`: the SDXL-base `unet/config.json` values give the SDXL geometry; other head sizes are refused."""
    from instantir_amd import loaders
    from instantir_amd.config import UNetConfig
    sdxl = dict(block_out_channels=[320, 640, 1280], down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                transformer_layers_per_block=[1, 2, 10], attention_head_dim=[5, 10, 20], cross_attention_dim=2048, layers_per_block=2,
                addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
                in_channels=4, out_channels=4, norm_num_groups=32)
    got, want = loaders.unet_config_from_dict(sdxl), UNetConfig.sdxl()
    for f in ("block_out_channels", "transformer_depth", "layers_per_block", "cross_attention_dim", "addition_time_embed_dim", "pooled_dim",
              "norm_groups", "in_channels", "out_channels"):
        assert getattr(got, f) == getattr(want, f), f
    with pytest.raises(ValueError):
        loaders.unet_config_from_dict(dict(sdxl, attention_head_dim=8))              # head_dim 40 / 80 / 160
    with pytest.raises(ValueError):
        loaders.unet_config_from_dict(dict(sdxl, addition_embed_type="text"))


def test_oracle_chain_resumed_in_sittings_equals_one_chain():
    """`oracle.pipeline.denoise(resume=, on_step=)` (what tools/parity_chain30.py relies on to run 30 full-size steps across several
    GPU-box calls): a DDIM chain stopped after 3 of 6 steps and continued from the saved latents is bit-identical to the one chain."""
    from instantir_amd import weights as W
    from instantir_amd.config import UNetConfig
    from oracle import pipeline as OP
    cfg = UNetConfig.tiny()
    P = {k: v.float() for k, v in W.synth_state_dict(W.unet_specs(cfg), 11).items()}
    PA = {k: v.float() for k, v in W.synth_state_dict(W.aggregator_specs(cfg), 12).items()}
    L = {k: v.float() for k, v in W.synth_state_dict(W.lora_specs(cfg), 13).items()}
    L["scaling"] = 2.0
    g = torch.Generator().manual_seed(1)
    B, H = 1, 8
    lq = torch.randn(B, 4, H, H, generator=g)
    pe, po = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g), torch.randn(B, cfg.pooled_dim, generator=g)
    im = torch.randn(2, B, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)
    kw = dict(negative_prompt_embeds=pe * 0.5, negative_pooled=po * 0.5, init_noise=torch.randn(B, 4, H, H, generator=g),
              num_inference_steps=6, guidance_scale=7.0)
    full = OP.denoise(P, PA, L, cfg, lq, pe, po, im, **kw)
    st = {}
    OP.denoise(P, PA, L, cfg, lq, pe, po, im, resume=(None, 0, 3), on_step=lambda i, x: st.update(x=x.clone(), n=i + 1), **kw)
    assert st["n"] == 3
    assert torch.equal(OP.denoise(P, PA, L, cfg, lq, pe, po, im, resume=(st["x"], 3, 6), **kw), full)
    with pytest.raises(ValueError):
        OP.denoise(P, PA, L, cfg, lq, pe, po, im, resume=(st["x"], 3, 6), adastep_restore=True, **kw)
