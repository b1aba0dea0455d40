"""Parameter inventories (names + shapes) of the networks on the denoising path and a seeded
synthetic initialiser.

No SDXL / DINOv2 / InstantIR checkpoint exists offline (SURVEY.md section 7 "Hard parts"), so parity
tests and the benchmark run on seeded synthetic weights with the exact parameter names and shapes
the reference's loaders expect:

* UNet: diffusers SDXL names (`module/min_sdxl.py:803-840`) + TA-IP processor weights
  (`...attn2.processor.{to_k_ip,to_v_ip}.weight`, `ln_{k,v}_ip.linear.{weight,bias}`,
  `module/ip_adapter/attention_processor.py:1086-1091`) + Resampler under
  `encoder_hid_proj.image_projection_layers.0` (`module/ip_adapter/utils.py:157-161`).
* Aggregator: `module/aggregator.py:304-306,386-396,414-471` after `remove_attn2`.
* Previewer LoRA: peft names `<path>.lora_A.weight` / `.lora_B.weight`, suffix-matched targets
  `pipelines/sdxl_instantir.py:141-162`.

Zero-initialised layers of the reference (zero 1x1 convs, adaLN linear, LoRA B) get N(0, 0.02)
so that every branch is live in the tests.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .config import UNetConfig, VAEConfig

Spec = Tuple[str, Tuple[int, ...], str]   # (name, shape, kind)


# kinds: w (fan-in normal), wo (fan-in normal, damped: feeds a residual sum), b (small bias),
#        g (norm gain ~1), z (reference zero-init -> N(0,.02)), lat (resampler latents)
def _lin(specs: List[Spec], path, cin, cout, bias=True, kind="w"):
    specs.append((path + ".weight", (cout, cin), kind))
    if bias:
        specs.append((path + ".bias", (cout,), "b" if kind != "z" else "z"))


def _conv(specs, path, cin, cout, k, kind="w"):
    specs.append((path + ".weight", (cout, cin, k, k), kind))
    specs.append((path + ".bias", (cout,), "b" if kind != "z" else "z"))


def _norm(specs, path, c):
    specs.append((path + ".weight", (c,), "g"))
    specs.append((path + ".bias", (c,), "b"))


def _resnet(specs, path, cin, cout, temb):
    _norm(specs, path + ".norm1", cin)
    _conv(specs, path + ".conv1", cin, cout, 3)
    _lin(specs, path + ".time_emb_proj", temb, cout)
    _norm(specs, path + ".norm2", cout)
    _conv(specs, path + ".conv2", cout, cout, 3, kind="wo")
    if cin != cout:
        _conv(specs, path + ".conv_shortcut", cin, cout, 1)


def _transformer(specs, path, c, depth, cfg: UNetConfig, cross: bool):
    _norm(specs, path + ".norm", c)
    _lin(specs, path + ".proj_in", c, c)
    for k in range(depth):
        p = f"{path}.transformer_blocks.{k}"
        _norm(specs, p + ".norm1", c)
        for n in ("to_q", "to_k", "to_v"):
            _lin(specs, f"{p}.attn1.{n}", c, c, bias=False)
        _lin(specs, p + ".attn1.to_out.0", c, c, kind="wo")
        if cross:
            _norm(specs, p + ".norm2", c)
            _lin(specs, p + ".attn2.to_q", c, c, bias=False)
            _lin(specs, p + ".attn2.to_k", cfg.cross_attention_dim, c, bias=False)
            _lin(specs, p + ".attn2.to_v", cfg.cross_attention_dim, c, bias=False)
            _lin(specs, p + ".attn2.to_out.0", c, c, kind="wo")
            pp = p + ".attn2.processor"
            _lin(specs, pp + ".to_k_ip", cfg.cross_attention_dim, c, bias=False)
            _lin(specs, pp + ".to_v_ip", cfg.cross_attention_dim, c, bias=False)
            _lin(specs, pp + ".ln_k_ip.linear", cfg.time_embed_dim, 2 * c, kind="z")
            _lin(specs, pp + ".ln_v_ip.linear", cfg.time_embed_dim, 2 * c, kind="z")
        _norm(specs, p + ".norm3", c)
        _lin(specs, p + ".ff.net.0.proj", c, 8 * c)
        _lin(specs, p + ".ff.net.2", 4 * c, c, kind="wo")
    _lin(specs, path + ".proj_out", c, c, kind="wo")


def _encoder_half(specs, cfg: UNetConfig, cross: bool):
    """conv_in, embeddings, down blocks, mid block -- shared by the UNet and the Aggregator."""
    c0 = cfg.block_out_channels[0]
    temb = cfg.time_embed_dim
    _conv(specs, "conv_in", cfg.in_channels, c0, 3)
    _lin(specs, "time_embedding.linear_1", c0, temb)
    _lin(specs, "time_embedding.linear_2", temb, temb)
    _lin(specs, "add_embedding.linear_1", cfg.add_embed_in, temb)
    _lin(specs, "add_embedding.linear_2", temb, temb)
    cin = c0
    nb = len(cfg.block_out_channels)
    for i, c in enumerate(cfg.block_out_channels):
        for j in range(cfg.layers_per_block):
            _resnet(specs, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else c, c, temb)
            if cfg.transformer_depth[i] > 0:
                _transformer(specs, f"down_blocks.{i}.attentions.{j}", c, cfg.transformer_depth[i], cfg, cross)
        if i < nb - 1:
            _conv(specs, f"down_blocks.{i}.downsamplers.0.conv", c, c, 3)
        cin = c
    c = cfg.block_out_channels[-1]
    _resnet(specs, "mid_block.resnets.0", c, c, temb)
    _transformer(specs, "mid_block.attentions.0", c, cfg.mid_depth, cfg, cross)
    _resnet(specs, "mid_block.resnets.1", c, c, temb)


def skip_channels(cfg: UNetConfig) -> List[int]:
    """Channel count of every skip tensor in push order (conv_in first)."""
    out = [cfg.block_out_channels[0]]
    nb = len(cfg.block_out_channels)
    for i, c in enumerate(cfg.block_out_channels):
        out += [c] * cfg.layers_per_block
        if i < nb - 1:
            out.append(c)
    return out


def unet_specs(cfg: UNetConfig) -> List[Spec]:
    specs: List[Spec] = []
    _encoder_half(specs, cfg, cross=True)
    temb = cfg.time_embed_dim
    skips = skip_channels(cfg)
    rev = list(reversed(cfg.block_out_channels))
    depth = list(reversed(cfg.transformer_depth))
    prev = rev[0]
    nb = len(rev)
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            sc = skips.pop()
            _resnet(specs, f"up_blocks.{i}.resnets.{j}", prev + sc, c, temb)
            prev = c
            if depth[i] > 0:
                _transformer(specs, f"up_blocks.{i}.attentions.{j}", c, depth[i], cfg, True)
        if i < nb - 1:
            _conv(specs, f"up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    _norm(specs, "conv_norm_out", cfg.block_out_channels[0])
    _conv(specs, "conv_out", cfg.block_out_channels[0], cfg.out_channels, 3)
    # Resampler (module/ip_adapter/resampler.py:81-125)
    rc = cfg.resampler
    p = "encoder_hid_proj.image_projection_layers.0"
    specs.append((p + ".latents", (1, rc.num_queries, rc.dim), "lat"))
    _lin(specs, p + ".proj_in", rc.embedding_dim, rc.dim)
    _lin(specs, p + ".proj_out", rc.dim, rc.output_dim)
    _norm(specs, p + ".norm_out", rc.output_dim)
    inner = rc.dim_head * rc.heads
    for i in range(rc.depth):
        a = f"{p}.layers.{i}.0"
        _norm(specs, a + ".norm1", rc.dim)
        _norm(specs, a + ".norm2", rc.dim)
        _lin(specs, a + ".to_q", rc.dim, inner, bias=False)
        _lin(specs, a + ".to_kv", rc.dim, 2 * inner, bias=False)
        _lin(specs, a + ".to_out", inner, rc.dim, bias=False, kind="wo")
        f = f"{p}.layers.{i}.1"
        _norm(specs, f + ".0", rc.dim)
        _lin(specs, f + ".1", rc.dim, rc.dim * rc.ff_mult, bias=False)
        _lin(specs, f + ".3", rc.dim * rc.ff_mult, rc.dim, bias=False, kind="wo")
    return specs


def aggregator_specs(cfg: UNetConfig) -> List[Spec]:
    specs: List[Spec] = []
    _encoder_half(specs, cfg, cross=False)
    c0 = cfg.block_out_channels[0]
    _conv(specs, "ref_conv_in", cfg.in_channels, c0, 3)
    for k, c in enumerate(skip_channels(cfg)):
        p = f"controlnet_down_blocks.{k}"
        _conv(specs, p + ".0.mlp_shared.0", c, cfg.sft_hidden, 3)
        _conv(specs, p + ".0.mul", cfg.sft_hidden, c, 3, kind="wo")
        _conv(specs, p + ".0.add", cfg.sft_hidden, c, 3, kind="wo")
        _conv(specs, p + ".1", c, c, 1, kind="z")
    c = cfg.block_out_channels[-1]
    p = "controlnet_mid_block"
    _conv(specs, p + ".0.mlp_shared.0", c, cfg.sft_hidden, 3)
    _conv(specs, p + ".0.mul", cfg.sft_hidden, c, 3, kind="wo")
    _conv(specs, p + ".0.add", cfg.sft_hidden, c, 3, kind="wo")
    _conv(specs, p + ".1", c, c, 1, kind="z")
    return specs


PREVIEWER_LORA_MODULES = (   # pipelines/sdxl_instantir.py:141-162
    "to_q", "to_kv", "0.to_out", "attn1.to_k", "attn1.to_v", "to_k_ip", "to_v_ip", "ln_k_ip.linear",
    "ln_v_ip.linear", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2", "conv1", "conv2",
    "conv_shortcut", "downsamplers.0.conv", "upsamplers.0.conv", "time_emb_proj",
)
LCM_LORA_MODULES = (         # pipelines/sdxl_instantir.py:125-140
    "to_q", "to_k", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2", "conv1", "conv2",
    "conv_shortcut", "downsamplers.0.conv", "upsamplers.0.conv", "time_emb_proj",
)


def lora_target(path: str, targets=PREVIEWER_LORA_MODULES) -> bool:
    """peft suffix matching: module path == target or endswith('.' + target)."""
    return any(path == t or path.endswith("." + t) for t in targets)


def lora_specs(cfg: UNetConfig, targets=PREVIEWER_LORA_MODULES) -> List[Spec]:
    r = cfg.lora_rank
    specs: List[Spec] = []
    for name, shape, _ in unet_specs(cfg):
        if not name.endswith(".weight") or len(shape) not in (2, 4):
            continue
        path = name[: -len(".weight")]
        if not lora_target(path, targets):
            continue
        if len(shape) == 2:
            specs.append((path + ".lora_A.weight", (r, shape[1]), "w"))
            specs.append((path + ".lora_B.weight", (shape[0], r), "z"))
        else:
            specs.append((path + ".lora_A.weight", (r, shape[1], shape[2], shape[3]), "w"))
            specs.append((path + ".lora_B.weight", (shape[0], r, 1, 1), "z"))
    return specs


def vae_decoder_specs(cfg: VAEConfig) -> List[Spec]:
    """AutoencoderKL post_quant_conv + Decoder (module/diffusers_vae/vae.py:197-350)."""
    specs: List[Spec] = []
    ch = list(reversed(cfg.block_out_channels))
    _conv(specs, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    _conv(specs, "decoder.conv_in", cfg.latent_channels, ch[0], 3)
    _vae_resnet(specs, "decoder.mid_block.resnets.0", ch[0], ch[0])
    a = "decoder.mid_block.attentions.0"
    _norm(specs, a + ".group_norm", ch[0])
    for n in ("to_q", "to_k", "to_v"):
        _lin(specs, f"{a}.{n}", ch[0], ch[0])
    _lin(specs, a + ".to_out.0", ch[0], ch[0], kind="wo")
    _vae_resnet(specs, "decoder.mid_block.resnets.1", ch[0], ch[0])
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block + 1):
            _vae_resnet(specs, f"decoder.up_blocks.{i}.resnets.{j}", prev, c)
            prev = c
        if i < len(ch) - 1:
            _conv(specs, f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    _norm(specs, "decoder.conv_norm_out", ch[-1])
    _conv(specs, "decoder.conv_out", ch[-1], cfg.in_channels, 3)
    return specs


def vae_encoder_specs(cfg: VAEConfig) -> List[Spec]:
    """AutoencoderKL Encoder + quant_conv (module/diffusers_vae/vae.py:46-195)."""
    specs: List[Spec] = []
    ch = list(cfg.block_out_channels)
    _conv(specs, "encoder.conv_in", cfg.in_channels, ch[0], 3)
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _vae_resnet(specs, f"encoder.down_blocks.{i}.resnets.{j}", prev, c)
            prev = c
        if i < len(ch) - 1:
            _conv(specs, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
    _vae_resnet(specs, "encoder.mid_block.resnets.0", prev, prev)
    a = "encoder.mid_block.attentions.0"
    _norm(specs, a + ".group_norm", prev)
    for n in ("to_q", "to_k", "to_v"):
        _lin(specs, f"{a}.{n}", prev, prev)
    _lin(specs, a + ".to_out.0", prev, prev, kind="wo")
    _vae_resnet(specs, "encoder.mid_block.resnets.1", prev, prev)
    _norm(specs, "encoder.conv_norm_out", prev)
    _conv(specs, "encoder.conv_out", prev, 2 * cfg.latent_channels, 3)
    _conv(specs, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return specs


def _vae_resnet(specs, path, cin, cout):
    _norm(specs, path + ".norm1", cin)
    _conv(specs, path + ".conv1", cin, cout, 3)
    _norm(specs, path + ".norm2", cout)
    _conv(specs, path + ".conv2", cout, cout, 3, kind="wo")
    if cin != cout:
        _conv(specs, path + ".conv_shortcut", cin, cout, 1)


def synth_state_dict(specs: List[Spec], seed: int, device="cpu", dtype=torch.float16) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights.  Generated on `device` (CPU for parity tests so the CPU oracle and
    the GPU path see identical bits after upload; GPU for the full-size benchmark)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = {}
    for name, shape, kind in specs:
        if kind in ("w", "wo"):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            std = (0.5 if kind == "wo" else 1.0) / fan_in ** 0.5
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std
        elif kind == "b":
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.02
        elif kind == "g":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        elif kind == "z":
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.02
        elif kind == "lat":
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) / shape[-1] ** 0.5
        else:
            raise ValueError(kind)
        sd[name] = t.to(dtype)
    return sd
