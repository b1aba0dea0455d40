"""CPU: on-disk format readers and the infer CLI's host helpers, on synthetic files."""
import hashlib
import os

import pytest
import torch
from PIL import Image

from instantir_amd import loaders, weights as W
from instantir_amd.config import UNetConfig


def test_attn_processor_index_layout():
    """140 processors, 70 with weights at the ODD indices; down_blocks.1 -> 1..7,
    down_blocks.2 -> 9..47, up_blocks.0 -> 49..107, up_blocks.1 -> 109..119, mid_block -> 121..139."""
    paths = loaders.attn_processor_paths(UNetConfig.sdxl())
    assert len(paths) == 140
    idx = {p: i for i, p in enumerate(paths)}
    assert idx["down_blocks.1.attentions.0.transformer_blocks.0.attn2"] == 1
    assert idx["down_blocks.1.attentions.1.transformer_blocks.1.attn2"] == 7
    assert idx["down_blocks.2.attentions.0.transformer_blocks.0.attn2"] == 9
    assert idx["down_blocks.2.attentions.1.transformer_blocks.9.attn2"] == 47
    assert idx["up_blocks.0.attentions.0.transformer_blocks.0.attn2"] == 49
    assert idx["up_blocks.0.attentions.2.transformer_blocks.9.attn2"] == 107
    assert idx["up_blocks.1.attentions.0.transformer_blocks.0.attn2"] == 109
    assert idx["up_blocks.1.attentions.2.transformer_blocks.1.attn2"] == 119
    assert idx["mid_block.attentions.0.transformer_blocks.0.attn2"] == 121
    assert idx["mid_block.attentions.0.transformer_blocks.9.attn2"] == 139
    assert all(p.endswith("attn1") for p in paths[0::2]) and all(p.endswith("attn2") for p in paths[1::2])


def _split_unet_and_adapter(cfg):
    full = W.synth_state_dict(W.unet_specs(cfg), 3)
    base = {k: v for k, v in full.items() if ".processor." not in k and not k.startswith("encoder_hid_proj.")}
    paths = loaders.attn_processor_paths(cfg)
    ip = {}
    for i, p in enumerate(paths):
        for k, v in full.items():
            if k.startswith(p + ".processor."):
                ip[f"{i}.{k[len(p + '.processor.'):]}"] = v
    proj = {k[len("encoder_hid_proj.image_projection_layers.0."):]: v for k, v in full.items() if k.startswith("encoder_hid_proj.")}
    return full, base, ip, proj


def test_adapter_three_layouts_round_trip(tmp_path):
    from safetensors.torch import save_file
    cfg = UNetConfig.tiny()
    full, base, ip, proj = _split_unet_and_adapter(cfg)
    p1 = str(tmp_path / "adapter.pt")
    torch.save({"image_proj": proj, "ip_adapter": ip}, p1)
    p2 = str(tmp_path / "legacy.pt")
    torch.save({**{"image_proj_model." + k: v for k, v in proj.items()}, **{"adapter_modules." + k: v for k, v in ip.items()}}, p2)
    p3 = str(tmp_path / "adapter.safetensors")
    save_file({**{"image_proj." + k: v.contiguous() for k, v in proj.items()}, **{"ip_adapter." + k: v.contiguous() for k, v in ip.items()}}, p3)
    for p in (p1, p2, p3, {"image_proj": proj, "ip_adapter": ip}):
        got = loaders.install_adapter(cfg, base, p)
        assert set(got) == set(full)
        assert all(torch.equal(got[k], full[k]) for k in full)


def test_adapter_missing_and_unexpected_keys():
    cfg = UNetConfig.tiny()
    full, base, ip, proj = _split_unet_and_adapter(cfg)
    ln_only_missing = {k: v for k, v in ip.items() if "ln" not in k}
    got = loaders.install_adapter(cfg, base, {"image_proj": proj, "ip_adapter": ln_only_missing})      # tolerated (utils.py:147-150)
    k = [n for n in got if n.endswith("ln_k_ip.linear.weight")][0]
    assert got[k].abs().max() == 0                                                    # AdaLayerNorm zero init
    some = next(k for k in ip if k.endswith("to_k_ip.weight"))
    with pytest.raises(ValueError):
        loaders.install_adapter(cfg, base, {"image_proj": proj, "ip_adapter": {k: v for k, v in ip.items() if k != some}})
    with pytest.raises(ValueError):
        loaders.install_adapter(cfg, base, {"image_proj": proj, "ip_adapter": {**ip, "0.to_k_ip.weight": torch.zeros(1)}})


def test_previewer_lora_file_conversion(tmp_path):
    cfg = UNetConfig.tiny()
    peft = W.synth_state_dict(W.lora_specs(cfg), 4)
    disk = {}
    for k, v in peft.items():                       # what save_lora_weights writes: no ".processor", unet. prefix
        k = k.replace(".lora_A.weight", ".lora.down.weight").replace(".lora_B.weight", ".lora.up.weight").replace("attn2.processor", "attn2")
        disk["unet." + k] = v
    disk["unet.conv_in.alpha"] = torch.tensor(16.0)
    disk["text_encoder.foo.lora.down.weight"] = torch.zeros(1)
    os.makedirs(tmp_path / "inst")
    torch.save(disk, tmp_path / "inst" / "previewer_lora_weights.bin")
    got, alpha = loaders.read_previewer_lora(str(tmp_path / "inst"))
    assert alpha == 16.0 and set(got) == set(peft) and all(torch.equal(got[k], peft[k]) for k in peft)
    from instantir_amd.pipeline import InstantIRPipeline
    pipe = InstantIRPipeline(cfg, W.synth_state_dict(W.unet_specs(cfg), 3), device="cpu")
    assert pipe.prepare_previewers(got, lora_alpha=alpha) == 16.0 and pipe._lora_scaling == 16.0 / cfg.lora_rank


def test_component_loader_and_aggregator(tmp_path):
    from safetensors.torch import save_file
    os.makedirs(tmp_path / "sdxl" / "unet")
    sd = {"conv_in.weight": torch.randn(4, 4), "conv_in.bias": torch.randn(4)}
    save_file(sd, str(tmp_path / "sdxl" / "unet" / "diffusion_pytorch_model.fp16.safetensors"))
    save_file({"conv_in.weight": torch.zeros(4, 4)}, str(tmp_path / "sdxl" / "unet" / "diffusion_pytorch_model.safetensors"))
    got = loaders.load_component(str(tmp_path / "sdxl"), "unet")
    assert torch.equal(got["conv_in.weight"], sd["conv_in.weight"])       # fp16 variant preferred (infer.py --variant fp16)
    torch.save(sd, tmp_path / "aggregator.pt")
    assert set(loaders.read_aggregator(str(tmp_path / "aggregator.pt"))) == set(sd)
    with pytest.raises(FileNotFoundError):
        loaders.load_component(str(tmp_path / "sdxl"), "vae")


def test_resize_img_and_batches(tmp_path):
    from instantir_amd.infer import DEFAULT_NEG_PROMPT, DEFAULT_PROMPT, build_parser, plan_batches, resize_img
    im = Image.new("RGB", (500, 300))
    out, size = resize_img(im)
    assert size == (500, 300) and out.size == (1024, 576)            # min side -> 768: 1280x768; max side cap: 1024x614; floor to 64: 1024x576
    out, size = resize_img(Image.new("RGB", (2000, 1000)))
    assert size == (2000, 1000) and out.size == (1024, 512)
    out, size = resize_img(Image.new("RGB", (800, 800)), width=1024)
    assert size == (1024, 1024) and out.size == (1024, 1024)
    out, _ = resize_img(Image.new("RGB", (900, 700)), pad_to_max_side=True)
    assert out.size == (1024, 1024)
    (tmp_path / "in").mkdir(); (tmp_path / "out").mkdir()
    for n in ("b.png", "a.png", "c.png", "d.png"):
        Image.new("RGB", (8, 8)).save(tmp_path / "in" / n)
    Image.new("RGB", (8, 8)).save(tmp_path / "out" / "c.png")
    assert plan_batches(str(tmp_path / "in"), str(tmp_path / "out"), 2) == [["a.png", "b.png"], ["d.png"]]
    assert plan_batches(str(tmp_path / "in" / "a.png"), str(tmp_path / "out"), 6) == [["a.png"]]
    # default prompts byte-for-byte (infer.py:192-205; the continuation indentation is part of the text)
    assert hashlib.sha256(DEFAULT_PROMPT.encode()).hexdigest() == "92b3073ceb476cd8d20ae82dea3b6c8ceaec4dcf0debe4314738deba4cb67174"
    assert hashlib.sha256(DEFAULT_NEG_PROMPT.encode()).hexdigest() == "aeb334b508e24d52307b5bc41cd8c95b61a359b717a5568c5b6793de4599e27e"
    a = build_parser().parse_args(["--test_path", "x"])
    assert (a.num_inference_steps, a.cfg, a.creative_start, a.preview_start, a.batch_size, a.seed, a.denoising_start) == \
        (30, 7.0, 1.0, 0.0, 6, 42, 1000)


def test_batches_shard_over_ranks_without_overlap(tmp_path):
    """contiguous split of infer.py's batch list; every image restored exactly once."""
    from instantir_amd.infer import plan_batches, shard_batches
    src, out = tmp_path / "in", tmp_path / "out"
    src.mkdir(); out.mkdir()
    for i in range(11):
        (src / f"{i:02d}.png").write_bytes(b"")
    (out / "03.png").write_bytes(b"")                       # already restored: skipped before sharding
    batches = plan_batches(str(src), str(out), 2)
    assert sum(len(b) for b in batches) == 10 and ["03.png"] not in batches
    for world in (1, 2, 3, 8):
        got = [b for r in range(world) for b in shard_batches(batches, r, world)]
        assert got == batches
