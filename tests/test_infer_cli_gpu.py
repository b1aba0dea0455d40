"""GPU: the infer CLI end to end on synthetic tiny weights (infer.py:114-225 behaviour: resize_img, batching, skip of
finished outputs, resize back to the input size, one output file per input)."""
import os

import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def test_infer_cli_synthetic_tiny(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    from instantir_amd.infer import build_parser, main
    src, out = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    rng = np.random.default_rng(0)
    for n in ("a.png", "b.png", "c.png"):
        Image.fromarray(rng.integers(0, 255, (96, 96, 3), dtype=np.uint8)).save(src / n)
    args = build_parser().parse_args(["--test_path", str(src), "--out_path", str(out), "--synthetic", "tiny", "--num_inference_steps", "2",
                                      "--width", "128", "--height", "128", "--batch_size", "2", "--cfg", "5.0", "--creative_start", "0.5"])
    import instantir_amd.infer as cli
    orig = cli.resize_img
    cli.resize_img = lambda im, **kw: orig(im, max_side=128, min_side=128, **kw)      # keep the tiny nets tiny
    try:
        main(args, torch.device("cuda:0"))
        files = sorted(os.listdir(out))
        assert files == ["a.png", "b.png", "c.png"]
        for f in files:
            im = Image.open(out / f)
            assert im.size == (128, 128) and im.mode == "RGB"
        before = {f: os.path.getmtime(out / f) for f in files}
        main(args, torch.device("cuda:0"))                                        # second run: everything is skipped
        assert {f: os.path.getmtime(out / f) for f in files} == before
    finally:
        cli.resize_img = orig


def _write_checkpoint_tree(root, cfg, vc):
    """A tiny SDXL + InstantIR + DINOv2 checkpoint tree in the on-disk formats infer.py reads, from
    seeded synthetic tensors.  Returns the in-memory pieces for the directly-constructed twin pipeline."""
    import json
    from safetensors.torch import save_file
    from tokenizers import pre_tokenizers
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection, CLIPTokenizer, Dinov2Config, Dinov2Model
    from instantir_amd import loaders, weights as W
    sdxl, iir, dino_dir = root / "sdxl", root / "instantir", root / "dino"
    for d in ("unet", "vae", "text_encoder", "text_encoder_2", "tokenizer", "tokenizer_2", "scheduler"):
        (sdxl / d).mkdir(parents=True)
    iir.mkdir(); dino_dir.mkdir()
    c = lambda sd: {k: v.contiguous() for k, v in sd.items()}
    # -- SDXL UNet: base weights only; the TA-IP processors / Resampler travel in adapter.pt
    full = W.synth_state_dict(W.unet_specs(cfg), 11)
    base = {k: v for k, v in full.items() if ".processor." not in k and not k.startswith("encoder_hid_proj.")}
    save_file(c(base), str(sdxl / "unet" / "diffusion_pytorch_model.fp16.safetensors"))
    json.dump({"block_out_channels": list(cfg.block_out_channels), "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
               "transformer_layers_per_block": [1, cfg.transformer_depth[1], cfg.transformer_depth[2]], "layers_per_block": cfg.layers_per_block,
               "attention_head_dim": [ch // 64 for ch in cfg.block_out_channels], "cross_attention_dim": cfg.cross_attention_dim,
               "addition_embed_type": "text_time", "addition_time_embed_dim": cfg.addition_time_embed_dim,
               "projection_class_embeddings_input_dim": cfg.add_embed_in, "norm_num_groups": cfg.norm_groups, "in_channels": 4,
               "out_channels": 4}, open(sdxl / "unet" / "config.json", "w"))
    vae_sd = W.synth_state_dict(W.vae_decoder_specs(vc) + W.vae_encoder_specs(vc), 14)
    save_file(c(vae_sd), str(sdxl / "vae" / "diffusion_pytorch_model.safetensors"))
    json.dump({"block_out_channels": list(vc.block_out_channels), "layers_per_block": vc.layers_per_block, "latent_channels": 4,
               "in_channels": 3, "norm_num_groups": vc.norm_groups, "scaling_factor": vc.scaling_factor}, open(sdxl / "vae" / "config.json", "w"))
    # -- tokenizers: byte-level vocabulary without merges (every character is a token); <|endoftext|> has the largest id
    chars = sorted(set(pre_tokenizers.ByteLevel.alphabet()))
    vocab = {ch: i for i, ch in enumerate(chars)}
    vocab.update({ch + "</w>": len(chars) + i for i, ch in enumerate(chars)})
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    for t in ("tokenizer", "tokenizer_2"):
        CLIPTokenizer(vocab=vocab, merges=[], model_max_length=cfg.text_len).save_pretrained(str(sdxl / t))
    # -- text encoders (width 64 each -> 128-wide context = cfg.cross_attention_dim; projection = pooled_dim)
    torch.manual_seed(5)
    tes = []
    for sub, act in (("text_encoder", "quick_gelu"), ("text_encoder_2", "gelu")):
        tc = CLIPTextConfig(vocab_size=len(vocab), hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=1,
                            max_position_embeddings=cfg.text_len, projection_dim=cfg.pooled_dim, hidden_act=act, eos_token_id=2,
                            bos_token_id=vocab["<|startoftext|>"], pad_token_id=1)
        sd = {k: v.half().float().contiguous() for k, v in CLIPTextModelWithProjection(tc).state_dict().items()}
        save_file(sd, str(sdxl / sub / "model.safetensors"))
        json.dump({"hidden_act": act, "eos_token_id": 2, "layer_norm_eps": 1e-5}, open(sdxl / sub / "config.json", "w"))
        tes.append((sd, act))
    json.dump({"num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear", "steps_offset": 1,
               "timestep_spacing": "leading", "prediction_type": "epsilon", "_class_name": "EulerDiscreteScheduler"},
              open(sdxl / "scheduler" / "scheduler_config.json", "w"))
    # -- DINOv2 directory
    torch.manual_seed(8)
    dm = Dinov2Model(Dinov2Config(hidden_size=cfg.resampler.embedding_dim, num_hidden_layers=2, num_attention_heads=1, patch_size=14,
                                  image_size=224, mlp_ratio=4))
    dino_sd = {k: v.half().float().contiguous() for k, v in dm.state_dict().items()}
    save_file(dino_sd, str(dino_dir / "model.safetensors"))
    json.dump({"patch_size": 14, "num_attention_heads": 1, "layer_norm_eps": 1e-6}, open(dino_dir / "config.json", "w"))
    # -- InstantIR files: adapter.pt (dict layout), aggregator.pt, previewer LoRA in diffusers naming with a network alpha
    paths = loaders.attn_processor_paths(cfg)
    ip = {}
    for idx, p in enumerate(paths):
        if p.endswith("attn2"):
            for n in ("to_k_ip.weight", "to_v_ip.weight", "ln_k_ip.linear.weight", "ln_k_ip.linear.bias", "ln_v_ip.linear.weight",
                      "ln_v_ip.linear.bias"):
                ip[f"{idx}.{n}"] = full[f"{p}.processor.{n}"]
    pre = "encoder_hid_proj.image_projection_layers.0."
    torch.save({"image_proj": {k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}, "ip_adapter": ip}, str(iir / "adapter.pt"))
    agg = W.synth_state_dict(W.aggregator_specs(cfg), 12)
    torch.save(agg, str(iir / "aggregator.pt"))
    lora = W.synth_state_dict(W.lora_specs(cfg), 13)
    f = {}
    for k, v in lora.items():
        k = k.replace(".lora_A.weight", ".lora.down.weight").replace(".lora_B.weight", ".lora.up.weight").replace(".processor.", ".")
        f["unet." + k] = v
    f["unet.down_blocks.0.resnets.0.conv1.alpha"] = torch.tensor(4.0)
    torch.save(f, str(iir / "previewer_lora_weights.bin"))
    return dict(full=full, vae=vae_sd, tes=tes, dino=dino_sd, agg=agg, lora=lora, alpha=4.0)


def test_infer_cli_from_checkpoint_files(tmp_path):
    """infer.py:117-225 over real files: `from_pretrained(sdxl dir)` -> `load_adapter_to_pipe(adapter.pt, dino dir)` ->
    `prepare_previewers(dir)` -> `aggregator.load_state_dict(torch.load(aggregator.pt))` -> restoration with string prompts.
    The image written by the CLI must equal the one from a twin pipeline assembled directly from the same tensors."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    from transformers import CLIPTokenizer
    import instantir_amd.infer as cli
    from instantir_amd.config import UNetConfig, VAEConfig
    from instantir_amd.encoders import HipCLIPText, HipDinov2
    from instantir_amd.pipeline import InstantIRPipeline
    from instantir_amd.schedulers import DDPMScheduler, LCMSingleStepScheduler
    from instantir_amd.vae import HipVAE
    cfg0 = UNetConfig.tiny()
    import dataclasses
    cfg = dataclasses.replace(cfg0, text_len=77, resampler=dataclasses.replace(cfg0.resampler, seq_len=257))
    vc = VAEConfig.tiny()
    mem = _write_checkpoint_tree(tmp_path, cfg, vc)
    src, out = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    rng = np.random.default_rng(3)
    Image.fromarray(rng.integers(0, 255, (128, 128, 3), dtype=np.uint8)).save(src / "x.png")
    args = cli.build_parser().parse_args([
        "--sdxl_path", str(tmp_path / "sdxl"), "--instantir_path", str(tmp_path / "instantir"), "--vision_encoder_path", str(tmp_path / "dino"),
        "--test_path", str(src), "--out_path", str(out), "--num_inference_steps", "2", "--width", "128", "--height", "128", "--cfg", "5.0",
        "--seed", "7", "--adapter_tokens", str(cfg.num_ip_tokens)])
    orig = cli.resize_img
    cli.resize_img = lambda im, **kw: orig(im, max_side=128, min_side=128, **kw)
    try:
        torch.manual_seed(123)          # the VAE posterior sample draws from the GLOBAL RNG (pipelines/sdxl_instantir.py:1375-1379)
        cli.main(args, torch.device("cuda:0"))
    finally:
        cli.resize_img = orig
    got = np.asarray(Image.open(out / "x.png"))
    assert got.shape == (128, 128, 3)

    # twin: same tensors, no files
    dev = "cuda:0"
    tok = CLIPTokenizer.from_pretrained(str(tmp_path / "sdxl" / "tokenizer"))
    mk = lambda texts: tok(texts, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
    pipe = InstantIRPipeline(cfg, mem["full"], scheduler=DDPMScheduler(), vae=HipVAE(vc, mem["vae"], dev), device=dev,
                             image_encoder=HipDinov2(mem["dino"], dev, num_heads=1),
                             text_encoder=HipCLIPText(mem["tes"][0][0], dev, hidden_act="quick_gelu"),
                             text_encoder_2=HipCLIPText(mem["tes"][1][0], dev, hidden_act="gelu"), tokenizer=mk, tokenizer_2=mk)
    assert pipe.prepare_previewers(mem["lora"], lora_alpha=mem["alpha"]) == 4.0
    pipe.aggregator.load_state_dict(mem["agg"])
    lq, _ = orig(Image.open(src / "x.png").convert("RGB"), max_side=128, min_side=128, width=128, height=128)
    g = torch.Generator(device=dev).manual_seed(7)
    torch.manual_seed(123)
    img = pipe(image=[lq], prompt=[cli.DEFAULT_PROMPT], negative_prompt=[cli.DEFAULT_NEG_PROMPT], ip_adapter_image=[lq],
               num_inference_steps=2, generator=g, guidance_scale=5.0, previewer_scheduler=LCMSingleStepScheduler.from_config(pipe.scheduler.config),
               preview_start=args.preview_start, control_guidance_end=args.creative_start).images[0]
    want = np.asarray(img.resize([128, 128], Image.BILINEAR))
    assert np.array_equal(got, want)
