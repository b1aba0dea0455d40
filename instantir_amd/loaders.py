"""Readers for the on-disk formats `infer.py` consumes, without diffusers / peft.

Nothing here could be exercised against real checkpoints (none exist offline): the layouts follow the reference's
loader code and are covered by round-trip tests on synthetic files (`tests/test_loaders_cpu.py`).  Files are opened
only with loaders that execute nothing from the file: safetensors, or `torch.load(..., weights_only=True)`.

* SDXL directory (`infer.py:117-120`): `unet/`, `vae/`, `text_encoder(_2)/` `*.safetensors` with diffusers /
  transformers parameter names (= the names this build uses).
* `adapter.pt` (`module/ip_adapter/utils.py:84-99,164-177`): {"image_proj", "ip_adapter"} dict, legacy flat
  `image_proj_model.* / adapter_modules.*`, or `.safetensors` with `image_proj.` / `ip_adapter.` prefixes.
* `aggregator.pt` (`infer.py:142-143`): plain state dict.
* `previewer_lora_weights.bin` (`pipelines/sdxl_instantir.py:356-374`): diffusers LoRA names with `unet.` prefix.
"""
from __future__ import annotations

import dataclasses
import glob
import json
import os
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

from .config import ResamplerConfig, UNetConfig, VAEConfig


def _load_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def load_component(model_dir: str, sub: str) -> Dict[str, torch.Tensor]:
    """State dict of one pipeline component: `<model_dir>/<sub>/*.safetensors` (fp16 variant preferred), else `.bin`."""
    d = os.path.join(model_dir, sub)
    cands = sorted(glob.glob(os.path.join(d, "*.fp16.safetensors"))) or sorted(glob.glob(os.path.join(d, "*.safetensors"))) \
        or sorted(glob.glob(os.path.join(d, "*.bin")))
    if not cands:
        raise FileNotFoundError(f"no weight file under {d}")
    sd: Dict[str, torch.Tensor] = {}
    for c in cands if len(cands) > 1 and "-of-" in cands[0] else cands[:1]:
        sd.update(_load_file(c))
    return sd


def attn_processor_paths(cfg: UNetConfig) -> List[str]:
    """Module paths of `unet.attn_processors` in traversal order: down_blocks, up_blocks, mid_block (both ModuleLists
    are registered before mid_block, module/unet/unet_2d_ZeroSFT.py:393-394 vs :464); attn1 before attn2."""
    out = []

    def tr(prefix, depth):
        for k in range(depth):
            out.append(f"{prefix}.transformer_blocks.{k}.attn1")
            out.append(f"{prefix}.transformer_blocks.{k}.attn2")

    for i, d in enumerate(cfg.transformer_depth):
        if d > 0:
            for j in range(cfg.layers_per_block):
                tr(f"down_blocks.{i}.attentions.{j}", d)
    for i, d in enumerate(reversed(cfg.transformer_depth)):
        if d > 0:
            for j in range(cfg.layers_per_block + 1):
                tr(f"up_blocks.{i}.attentions.{j}", d)
    tr("mid_block.attentions.0", cfg.mid_depth)
    return out


def read_adapter(path_or_dict) -> Dict[str, Dict[str, torch.Tensor]]:
    """-> {"image_proj": sd, "ip_adapter": sd} from any of the three layouts."""
    if isinstance(path_or_dict, dict):
        sd = path_or_dict
    elif path_or_dict.endswith(".safetensors"):
        flat = _load_file(path_or_dict)
        sd = {"image_proj": {}, "ip_adapter": {}}
        for k, v in flat.items():
            if k.startswith("image_proj."):
                sd["image_proj"][k[len("image_proj."):]] = v
            elif k.startswith("ip_adapter."):
                sd["ip_adapter"][k[len("ip_adapter."):]] = v
    else:
        sd = _load_file(path_or_dict)
    if "image_proj" not in sd and "ip_adapter" not in sd:            # legacy flat layout (utils.py:164-177)
        new = {"image_proj": OrderedDict(), "ip_adapter": OrderedDict()}
        for k, v in sd.items():
            if k.startswith("image_proj_model."):
                new["image_proj"][k[len("image_proj_model."):]] = v
            elif k.startswith("adapter_modules."):
                new["ip_adapter"][k[len("adapter_modules."):]] = v
        sd = new
    return sd


def install_adapter(cfg: UNetConfig, unet_sd: Dict[str, torch.Tensor], adapter) -> Dict[str, torch.Tensor]:
    """`load_adapter_to_pipe` for the state dict (module/ip_adapter/utils.py:136-161): TA-IP processor weights land
    under `<attn2>.processor.*`, the Resampler under `encoder_hid_proj.image_projection_layers.0.*`.  `to_k_ip/to_v_ip`
    start from the UNet's own attn2 `to_k/to_v` (attention_processor.py:1395-1411); keys missing from the file are
    tolerated only if they contain "ln" (utils.py:147-150: the adaLN linears then stay zero)."""
    ad = read_adapter(adapter)
    out = dict(unet_sd)
    paths = attn_processor_paths(cfg)
    used = set()
    for idx, p in enumerate(paths):
        if not p.endswith("attn2"):
            continue
        C = unet_sd[p + ".to_k.weight"].shape[0]
        want = {
            "to_k_ip.weight": unet_sd[p + ".to_k.weight"], "to_v_ip.weight": unet_sd[p + ".to_v.weight"],
            "ln_k_ip.linear.weight": torch.zeros(2 * C, cfg.time_embed_dim), "ln_k_ip.linear.bias": torch.zeros(2 * C),
            "ln_v_ip.linear.weight": torch.zeros(2 * C, cfg.time_embed_dim), "ln_v_ip.linear.bias": torch.zeros(2 * C),
        }
        for name, init in want.items():
            key = f"{idx}.{name}"
            if key in ad["ip_adapter"]:
                out[f"{p}.processor.{name}"] = ad["ip_adapter"][key]
                used.add(key)
            elif "ln" in key:
                out[f"{p}.processor.{name}"] = init.to(unet_sd[p + ".to_k.weight"].dtype)
            else:
                raise ValueError(f"Missing keys in adapter_modules: ['{key}']")
    unexpected = sorted(set(ad["ip_adapter"]) - used)
    if unexpected:
        raise ValueError(f"Unexpected keys in adapter_modules: {unexpected[:8]}")
    for k, v in ad["image_proj"].items():
        out["encoder_hid_proj.image_projection_layers.0." + k] = v
    return out


def read_previewer_lora(path_or_dict, weight_name="previewer_lora_weights.bin") -> Tuple[Dict[str, torch.Tensor], float]:
    """-> (peft-named LoRA state dict, lora_alpha).  Follows `prepare_previewers` (pipelines/sdxl_instantir.py:356-374):
    keep `unet.` keys, diffusers `.lora.down/.lora.up` (or `_lora.down/up`) -> `.lora_A/.lora_B`, re-insert
    `.processor` for the IP keys, first `*.alpha` entry (if any) is lora_alpha, else 1."""
    if isinstance(path_or_dict, dict):
        raw = path_or_dict
    else:
        p = path_or_dict if os.path.isfile(path_or_dict) else os.path.join(path_or_dict, weight_name)
        raw = _load_file(p)
    alpha = None
    out: Dict[str, torch.Tensor] = {}
    for k, v in raw.items():
        if k.endswith(".alpha"):
            if alpha is None:
                alpha = float(v)
            continue
        if not k.startswith("unet."):
            continue
        k = k[len("unet."):]
        k = k.replace(".lora.down.weight", ".lora_A.weight").replace(".lora.up.weight", ".lora_B.weight")
        k = k.replace("_lora.down.weight", ".lora_A.weight").replace("_lora.up.weight", ".lora_B.weight")
        k = k.replace(".lora_linear_layer.down.weight", ".lora_A.weight").replace(".lora_linear_layer.up.weight", ".lora_B.weight")
        if "ip" in k and ".processor." not in k:
            k = k.replace("attn2", "attn2.processor")
        out[k] = v
    return out, (alpha if alpha is not None else 1.0)


def read_aggregator(path: str) -> Dict[str, torch.Tensor]:
    return _load_file(path)


# ---- configs from the checkpoint directories / tensors ----------------------------------------------------------------
def _read_json(path):
    with open(path) as f:
        return json.load(f)


def unet_config_from_dict(c) -> UNetConfig:
    """A diffusers `UNet2DConditionModel` config (dict, or the `.config` object of a module) -> UNetConfig.  Only SDXL-family
    layouts are accepted: DownBlock2D first, head_dim 64, text_time addition embedding."""
    c = dict(c)
    boc = tuple(c["block_out_channels"])
    types = c.get("down_block_types", ["DownBlock2D"] + ["CrossAttnDownBlock2D"] * (len(boc) - 1))
    tl = c.get("transformer_layers_per_block", 1)
    tl = [tl] * len(boc) if isinstance(tl, int) else list(tl)
    depth = tuple(tl[i] if "CrossAttn" in types[i] else 0 for i in range(len(boc)))
    ahd = c.get("attention_head_dim", 8)
    ahd = [ahd] * len(boc) if isinstance(ahd, int) else list(ahd)       # SDXL stores the HEAD COUNT per block here
    for i, d in enumerate(depth):
        if d > 0 and boc[i] // ahd[i] != 64:
            raise ValueError(f"unet config: block {i} has head_dim {boc[i] // ahd[i]}; the HIP attention kernel is head_dim 64")
    if c.get("addition_embed_type", "text_time") != "text_time":
        raise ValueError("unet config: only addition_embed_type 'text_time' (SDXL) is supported")
    ate = c.get("addition_time_embed_dim", 256)
    base = UNetConfig.sdxl()
    return dataclasses.replace(
        base, in_channels=c.get("in_channels", 4), out_channels=c.get("out_channels", 4), block_out_channels=boc,
        transformer_depth=depth, layers_per_block=c.get("layers_per_block", 2), cross_attention_dim=c.get("cross_attention_dim", 2048),
        addition_time_embed_dim=ate, pooled_dim=c.get("projection_class_embeddings_input_dim", 2816) - 6 * ate,
        norm_groups=c.get("norm_num_groups", 32))


def unet_config_from_dir(model_dir: str) -> UNetConfig:
    """`<model_dir>/unet/config.json` -> UNetConfig; SDXL-base values when the file is absent."""
    path = os.path.join(model_dir, "unet", "config.json")
    return unet_config_from_dict(_read_json(path)) if os.path.isfile(path) else UNetConfig.sdxl()


def vae_config_from_dict(c) -> VAEConfig:
    c = dict(c)
    return VAEConfig(in_channels=c.get("in_channels", 3), latent_channels=c.get("latent_channels", 4),
                     block_out_channels=tuple(c["block_out_channels"]), layers_per_block=c.get("layers_per_block", 2),
                     norm_groups=c.get("norm_num_groups", 32), scaling_factor=c.get("scaling_factor", 0.13025))


def vae_config_from_dir(model_dir: str) -> VAEConfig:
    path = os.path.join(model_dir, "vae", "config.json")
    return vae_config_from_dict(_read_json(path)) if os.path.isfile(path) else VAEConfig.sdxl()


def resampler_config_from_state(image_proj: Dict[str, torch.Tensor], seq_len: int = 257) -> ResamplerConfig:
    """Geometry of the Resampler from its own tensors (module/ip_adapter/resampler.py:81-125); dim_head is 64 there."""
    q, dim = image_proj["latents"].shape[1:]
    depth = 1 + max(int(k.split(".")[1]) for k in image_proj if k.startswith("layers."))
    inner = image_proj["layers.0.0.to_q.weight"].shape[0]
    return ResamplerConfig(dim=dim, depth=depth, dim_head=64, heads=inner // 64, num_queries=q,
                           embedding_dim=image_proj["proj_in.weight"].shape[1], output_dim=image_proj["proj_out.weight"].shape[0],
                           ff_mult=image_proj["layers.0.1.1.weight"].shape[0] // dim, seq_len=seq_len)


def load_adapter_to_pipe(pipe, pretrained_model_path_or_dict, image_encoder_or_path=None, feature_extractor_or_path=None,
                         use_clip_encoder=False, adapter_tokens=64, use_lcm=False, use_adaln=True):
    """`module/ip_adapter/utils.py:73-161` for this build's pipeline: installs the TA-IP processors + Resampler into the UNet
    state (the nets are packed lazily, at the first call) and attaches the DINOv2 image encoder (a `HipDinov2`, or a
    directory holding `model.safetensors` / `pytorch_model.bin` [+ config.json]); with `use_clip_encoder=True` the directory
    is read as a CLIP vision tower (`encoders.HipCLIPVision`, whose penultimate hidden states feed the Resampler).
    `feature_extractor_or_path` is accepted and ignored: the fixed preprocessing lives in `encoders.dinov2_preprocess` /
    `encoders.clip_preprocess`."""
    if not use_adaln:
        raise NotImplementedError("use_adaln=False (IPAttnProcessor2_0 without AdaLayerNorm) is not built")
    ad = read_adapter(pretrained_model_path_or_dict)
    rc = resampler_config_from_state(ad["image_proj"], seq_len=pipe.cfg.resampler.seq_len)
    if rc.num_queries != adapter_tokens:
        raise ValueError(f"adapter has {rc.num_queries} image tokens, adapter_tokens={adapter_tokens}")
    pipe.cfg = dataclasses.replace(pipe.cfg, resampler=rc, num_ip_tokens=rc.num_queries)
    pipe._unet_sd = install_adapter(pipe.cfg, pipe._unet_sd, ad)
    pipe._unet = pipe._unet_prev = None
    pipe._prev_nets = {}
    pipe._unet_prev8 = None
    if image_encoder_or_path is not None:
        if isinstance(image_encoder_or_path, str):
            from .encoders import HipCLIPVision, HipDinov2
            cands = [os.path.join(image_encoder_or_path, n) for n in ("model.safetensors", "pytorch_model.bin")]
            found = [c for c in cands if os.path.exists(c)]
            if not found:
                raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {image_encoder_or_path}")
            kw = {}
            cj = os.path.join(image_encoder_or_path, "config.json")
            c = _read_json(cj) if os.path.isfile(cj) else {}
            if use_clip_encoder:          # CLIPVisionModelWithProjection.from_pretrained (module/ip_adapter/utils.py:106-112)
                c = c.get("vision_config", c)
                pipe.image_encoder = HipCLIPVision(_load_file(found[0]), pipe.device, patch_size=c.get("patch_size", 14),
                                                   num_heads=c.get("num_attention_heads"), eps=c.get("layer_norm_eps", 1e-5),
                                                   hidden_act=c.get("hidden_act", "quick_gelu"))
            else:
                if c:
                    kw = {"patch_size": c.get("patch_size", 14), "num_heads": c.get("num_attention_heads"), "eps": c.get("layer_norm_eps", 1e-6)}
                pipe.image_encoder = HipDinov2(_load_file(found[0]), pipe.device, **kw)
        else:
            pipe.image_encoder = image_encoder_or_path
    return pipe
