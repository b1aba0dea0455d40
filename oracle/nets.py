"""Oracle (test infrastructure): fp32 CPU restatement of the networks on the InstantIR step.

Functional style: every network is a function of (params dict, config, inputs).  `P` maps
diffusers-style parameter names to float32 CPU tensors; `lora` (optional)
maps `<module path>.lora_A.weight` / `.lora_B.weight` to tensors plus the key `"scaling"`.
Layout is NCHW / (B, T, C) like the reference.  Citations are to /root/reference.

Pinned against reference-generated goldens: `resampler`, `ada_layer_norm`, `attn_self`,
`attn_ta_ip`, `image_projection`.  Everything that restates diffusers-0.28.1 block arithmetic is
PARITY UNPINNED (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# leaves
# ----------------------------------------------------------------------------------------------
def sinusoid(t, dim):
    """module/min_sdxl.py:205-224 -- [cos, sin] order, exponent -ln(1e4) * k / (dim/2)."""
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32)
    freq = torch.exp(-math.log(10000) * k / half)
    ang = t.reshape(-1, 1).float() * freq[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _lora_delta_linear(lora, path, x):
    if lora is None or (path + ".lora_A.weight") not in lora:
        return None
    a, b = lora[path + ".lora_A.weight"], lora[path + ".lora_B.weight"]
    return lora["scaling"] * F.linear(F.linear(x, a), b)


def linear(P, path, x, lora=None):
    """nn.Linear, plus the peft side branch W x + s * B(A x)."""
    y = F.linear(x, P[path + ".weight"], P.get(path + ".bias"))
    d = _lora_delta_linear(lora, path, x)
    return y if d is None else y + d


def conv2d(P, path, x, stride=1, padding=1, lora=None):
    """nn.Conv2d; peft Conv2d LoRA: A is k x k (same stride/padding) Cin->r, B is 1x1 r->Cout."""
    y = F.conv2d(x, P[path + ".weight"], P.get(path + ".bias"), stride=stride, padding=padding)
    if lora is not None and (path + ".lora_A.weight") in lora:
        a, b = lora[path + ".lora_A.weight"], lora[path + ".lora_B.weight"]
        y = y + lora["scaling"] * F.conv2d(F.conv2d(x, a, None, stride=stride, padding=padding), b)
    return y


def group_norm(P, path, x, groups, eps):
    return F.group_norm(x, groups, P[path + ".weight"], P[path + ".bias"], eps)


def layer_norm(P, path, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P.get(path + ".weight"), P.get(path + ".bias"), eps)


def _heads(x, h):
    b, t, c = x.shape
    return x.reshape(b, t, h, c // h).transpose(1, 2)


def sdpa(q, k, v, heads):
    """F.scaled_dot_product_attention, non-causal, scale 1/sqrt(head_dim), written out."""
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(qh.shape[-1])
    o = torch.softmax(s, dim=-1) @ vh
    b, h, t, d = o.shape
    return o.transpose(1, 2).reshape(b, t, h * d)


# ----------------------------------------------------------------------------------------------
# attention processors (module/ip_adapter/attention_processor.py)
# ----------------------------------------------------------------------------------------------
def attn_self(P, path, x, heads, lora=None):
    """AttnProcessor2_0.__call__, attention_processor.py:337-414 (self-attention use)."""
    q = linear(P, path + ".to_q", x, lora)
    k = linear(P, path + ".to_k", x, lora)
    v = linear(P, path + ".to_v", x, lora)
    return linear(P, path + ".to_out.0", sdpa(q, k, v, heads), lora)


def ada_layer_norm(P, path, x, temb, lora=None):
    """AdaLayerNorm.forward, attention_processor.py:20-26: shift first, LN eps 1e-6 no affine."""
    emb = linear(P, path + ".linear", F.silu(temb), lora)
    shift, scale = emb.reshape(len(x), 1, -1).chunk(2, dim=-1)
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6) * (1 + scale) + shift


def attn_ta_ip(P, path, x, ctx, ip_tokens, temb, heads, lora=None, scale=1.0):
    """TA_IPAttnProcessor2_0.__call__, attention_processor.py:1093-1207 (tuple input branch)."""
    q = linear(P, path + ".to_q", x, lora)
    k = linear(P, path + ".to_k", ctx, lora)
    v = linear(P, path + ".to_v", ctx, lora)
    text = sdpa(q, k, v, heads)
    pp = path + ".processor"
    ipk = ada_layer_norm(P, pp + ".ln_k_ip", linear(P, pp + ".to_k_ip", ip_tokens, lora), temb, lora)
    ipv = ada_layer_norm(P, pp + ".ln_v_ip", linear(P, pp + ".to_v_ip", ip_tokens, lora), temb, lora)
    ip = sdpa(q, ipk, ipv, heads)
    return linear(P, path + ".to_out.0", text + scale * ip, lora)


# ----------------------------------------------------------------------------------------------
# Resampler (module/ip_adapter/resampler.py) and MultiIPAdapterImageProjection
# ----------------------------------------------------------------------------------------------
def _perceiver_attention(P, path, x, latents, heads, dim_head, lora):
    # resampler.py:50-78
    x = layer_norm(P, path + ".norm1", x)
    latents = layer_norm(P, path + ".norm2", latents)
    b, l, _ = latents.shape
    q = linear(P, path + ".to_q", latents, lora)
    kv = linear(P, path + ".to_kv", torch.cat((x, latents), dim=-2), lora)
    k, v = kv.chunk(2, dim=-1)
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    s = 1 / math.sqrt(math.sqrt(dim_head))
    w = (q * s) @ (k * s).transpose(-2, -1)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
    return linear(P, path + ".to_out", out, lora)


def resampler(P, path, x, rc, lora=None):
    """Resampler.forward, resampler.py:127-147 (no pos_emb, no mean-pooled latents)."""
    latents = P[path + ".latents"].repeat(x.size(0), 1, 1)
    x = linear(P, path + ".proj_in", x, lora)
    for i in range(rc.depth):
        latents = _perceiver_attention(P, f"{path}.layers.{i}.0", x, latents, rc.heads, rc.dim_head, lora) + latents
        ff = f"{path}.layers.{i}.1"
        h = layer_norm(P, ff + ".0", latents)
        h = F.linear(F.gelu(F.linear(h, P[ff + ".1.weight"])), P[ff + ".3.weight"])   # resampler.py:13-20
        latents = h + latents
    latents = linear(P, path + ".proj_out", latents, lora)
    return layer_norm(P, path + ".norm_out", latents)


def image_projection(P, image_embeds, rc, lora=None):
    """MultiIPAdapterImageProjection.forward, ip_adapter.py:68-90: (n, B, S, E) -> (n*B, Q, D)."""
    out = []
    for i, e in enumerate(image_embeds):
        e = e.reshape((e.shape[0] * e.shape[1],) + e.shape[2:])
        out.append(resampler(P, f"encoder_hid_proj.image_projection_layers.{i}", e, rc, lora))
    return out


# ----------------------------------------------------------------------------------------------
# SDXL blocks (spec text: module/min_sdxl.py)
# ----------------------------------------------------------------------------------------------
def resnet(P, path, x, temb, groups, lora=None, eps=1e-5):
    """ResnetBlock2D, module/min_sdxl.py:242-283."""
    h = F.silu(group_norm(P, path + ".norm1", x, groups, eps))
    h = conv2d(P, path + ".conv1", h, lora=lora)
    h = h + linear(P, path + ".time_emb_proj", F.silu(temb), lora)[:, :, None, None]
    h = F.silu(group_norm(P, path + ".norm2", h, groups, eps))
    h = conv2d(P, path + ".conv2", h, lora=lora)
    if (path + ".conv_shortcut.weight") in P:
        x = conv2d(P, path + ".conv_shortcut", x, padding=0, lora=lora)
    return x + h


def attn_cross(P, path, x, ctx, heads, lora=None):
    """AttnProcessor2_0.__call__ with encoder_hidden_states (attention_processor.py:337-414): the plain
    SDXL cross-attention of module/min_sdxl.py:550-554 (no IP branch)."""
    q = linear(P, path + ".to_q", x, lora)
    k = linear(P, path + ".to_k", ctx, lora)
    v = linear(P, path + ".to_v", ctx, lora)
    return linear(P, path + ".to_out.0", sdpa(q, k, v, heads), lora)


def geglu(P, path, x, lora=None):
    """GEGLU, module/min_sdxl.py:502-510: x1 * gelu(x2), exact-erf GELU."""
    a, g = linear(P, path + ".proj", x, lora).chunk(2, dim=-1)
    return a * F.gelu(g)


def feed_forward(P, path, x, lora=None):
    """FeedForward, module/min_sdxl.py:513-528: GEGLU -> Dropout(0) -> Linear."""
    return linear(P, path + ".net.2", geglu(P, path + ".net.0", x, lora), lora)


def transformer_block(P, path, x, ctx, ip_tokens, temb, heads, lora=None):
    """BasicTransformerBlock, module/min_sdxl.py:531-562.  attn2 is skipped when absent (Aggregator after
    remove_attn2, pipelines/sdxl_instantir.py:165-177), is the TA-IP processor when its weights are present
    (module/ip_adapter/attention_processor.py:1093-1207) and plain cross-attention otherwise (stock SDXL)."""
    x = attn_self(P, path + ".attn1", layer_norm(P, path + ".norm1", x), heads, lora) + x
    if (path + ".attn2.to_q.weight") in P:
        h = layer_norm(P, path + ".norm2", x)
        if (path + ".attn2.processor.to_k_ip.weight") in P:
            x = attn_ta_ip(P, path + ".attn2", h, ctx, ip_tokens, temb, heads, lora) + x
        else:
            x = attn_cross(P, path + ".attn2", h, ctx, heads, lora) + x
    return feed_forward(P, path + ".ff", layer_norm(P, path + ".norm3", x), lora) + x


def transformer2d(P, path, x, depth, ctx, ip_tokens, temb, heads, groups, lora=None):
    """Transformer2DModel (linear projection form), module/min_sdxl.py:565-595; outer GN eps 1e-6."""
    b, c, hh, ww = x.shape
    h = group_norm(P, path + ".norm", x, groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    h = linear(P, path + ".proj_in", h, lora)
    for k in range(depth):
        h = transformer_block(P, f"{path}.transformer_blocks.{k}", h, ctx, ip_tokens, temb, heads, lora)
    h = linear(P, path + ".proj_out", h, lora)
    return h.reshape(b, hh, ww, c).permute(0, 3, 1, 2) + x


def time_embedding(P, cfg, t, text_embeds, time_ids, lora=None):
    """emb = time_embedding(sincos(t)) + add_embedding(cat(pooled, sincos(time_ids)))
    (module/unet/unet_2d_ZeroSFT.py:998-1022,1056-1072; pipelines/sdxl_instantir.py:1516-1531).
    time_embedding / add_embedding are not LoRA targets (pipelines/sdxl_instantir.py:141-162)."""
    n = text_embeds.shape[0]
    t = torch.as_tensor(t).reshape(-1).expand(n)
    e = sinusoid(t, cfg.block_out_channels[0])
    e = linear(P, "time_embedding.linear_2", F.silu(linear(P, "time_embedding.linear_1", e)))
    tid = sinusoid(time_ids.flatten(), cfg.addition_time_embed_dim).reshape(n, -1)
    a = torch.cat([text_embeds, tid], dim=-1)
    a = linear(P, "add_embedding.linear_2", F.silu(linear(P, "add_embedding.linear_1", a)))
    return e + a


def down_block(P, path, x, emb, ctx, ip_tokens, n_res, depth, heads, groups, has_down, lora=None):
    """DownBlock2D / CrossAttnDownBlock2D, module/min_sdxl.py:621-679: returns (x, [outputs])."""
    outs = []
    for j in range(n_res):
        x = resnet(P, f"{path}.resnets.{j}", x, emb, groups, lora)
        if depth > 0:
            x = transformer2d(P, f"{path}.attentions.{j}", x, depth, ctx, ip_tokens, emb, heads, groups, lora)
        outs.append(x)
    if has_down:
        x = conv2d(P, f"{path}.downsamplers.0.conv", x, stride=2, padding=1, lora=lora)   # Downsample2D :598-606
        outs.append(x)
    return x, outs


def up_block(P, path, x, skips, emb, ctx, ip_tokens, depth, heads, groups, has_up, lora=None):
    """CrossAttnUpBlock2D / UpBlock2D, module/min_sdxl.py:682-761: `skips` is consumed from its END
    (res_hidden_states_tuple[-1] first), one per resnet."""
    skips = list(skips)
    for j in range(len(skips)):
        x = torch.cat([x, skips.pop()], dim=1)                           # min_sdxl.py:706-717
        x = resnet(P, f"{path}.resnets.{j}", x, emb, groups, lora)
        if depth > 0:
            x = transformer2d(P, f"{path}.attentions.{j}", x, depth, ctx, ip_tokens, emb, heads, groups, lora)
    if has_up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")          # Upsample2D, min_sdxl.py:609-618
        x = conv2d(P, f"{path}.upsamplers.0.conv", x, lora=lora)
    return x


def mid_block(P, path, x, emb, ctx, ip_tokens, depth, heads, groups, lora=None):
    """UNetMidBlock2DCrossAttn, module/min_sdxl.py:764-786."""
    x = resnet(P, path + ".resnets.0", x, emb, groups, lora)
    x = transformer2d(P, path + ".attentions.0", x, depth, ctx, ip_tokens, emb, heads, groups, lora)
    return resnet(P, path + ".resnets.1", x, emb, groups, lora)


def _down_blocks(P, cfg, x, emb, ctx, ip_tokens, lora):
    """conv_in output -> list of skip tensors (module/min_sdxl.py:869-889)."""
    skips = [x]
    nb = len(cfg.block_out_channels)
    for i, c in enumerate(cfg.block_out_channels):
        x, outs = down_block(P, f"down_blocks.{i}", x, emb, ctx, ip_tokens, cfg.layers_per_block,
                             cfg.transformer_depth[i], c // cfg.head_dim, cfg.norm_groups, i < nb - 1, lora)
        skips += outs
    return x, skips


def _mid_block(P, cfg, x, emb, ctx, ip_tokens, lora):
    c = cfg.block_out_channels[-1]
    return mid_block(P, "mid_block", x, emb, ctx, ip_tokens, cfg.mid_depth, c // cfg.head_dim, cfg.norm_groups, lora)


def unet_forward(P, cfg, sample, t, ctx, text_embeds, time_ids, ip_tokens, down_res=None, mid_res=None, lora=None,
                 emb=None):
    """UNet2DConditionModel.forward with additive ControlNet residuals.

    Structure: module/unet/unet_2d_ZeroSFT.py:1226-1388; leaves module/min_sdxl.py.  Deviation of
    that in-tree copy NOT followed: stock diffusers adds
    down_block_additional_residuals[k] to skip k and mid_block_additional_residual to the mid output.
    `ip_tokens` is the Resampler output (R, Q, D) -- the reference evaluates it inside every forward
    (`…ZeroSFT.py:1114-1121`); it is an argument here because it is step invariant.
    """
    if emb is None:
        emb = time_embedding(P, cfg, t, text_embeds, time_ids)
    x = conv2d(P, "conv_in", sample)
    x, skips = _down_blocks(P, cfg, x, emb, ctx, ip_tokens, lora)
    if down_res is not None:
        skips = [s + r for s, r in zip(skips, down_res)]
    x = _mid_block(P, cfg, x, emb, ctx, ip_tokens, lora)
    if mid_res is not None:
        x = x + mid_res
    rev = list(reversed(cfg.block_out_channels))
    nb = len(rev)
    n = cfg.layers_per_block + 1
    for i, c in enumerate(rev):
        depth = list(reversed(cfg.transformer_depth))[i]
        take, skips = skips[-n:], skips[:-n]
        x = up_block(P, f"up_blocks.{i}", x, take, emb, ctx, ip_tokens, depth, c // cfg.head_dim, cfg.norm_groups,
                     i < nb - 1, lora)
    x = F.silu(group_norm(P, "conv_norm_out", x, cfg.norm_groups, 1e-5))
    return conv2d(P, "conv_out", x)


# ----------------------------------------------------------------------------------------------
# Aggregator (module/aggregator.py)
# ----------------------------------------------------------------------------------------------
def sft(P, path, cond, h):
    """SFT.forward, module/aggregator.py:70-90, followed by the zero-initialised 1x1
    (`nn.Sequential(SFT, zero_module(Conv2d 1x1))`, :414-417)."""
    actv = F.silu(conv2d(P, path + ".0.mlp_shared.0", cond))
    gamma = conv2d(P, path + ".0.mul", actv)
    beta = conv2d(P, path + ".0.add", actv)
    h = h * (gamma + 1) + beta
    return conv2d(P, path + ".1", h, padding=0)


def aggregator_forward(P, cfg, sample, t, cond, text_embeds, time_ids):
    """Aggregator.forward, module/aggregator.py:758-977 (cat_dim=-2, pad_concat=False,
    conditioning_scale=1).  `sample` = LQ latent, `cond` = preview latent.  Returns (9 down, mid)."""
    emb = time_embedding(P, cfg, t, text_embeds, time_ids)                # :823-882
    x = torch.cat([conv2d(P, "conv_in", sample), conv2d(P, "ref_conv_in", cond)], dim=-2)   # :889-902
    x, skips = _down_blocks(P, cfg, x, emb, None, None, None)             # :906-928
    x = _mid_block(P, cfg, x, emb, None, None, None)                      # :931-936
    outs = []
    for k, s in enumerate(skips):                                         # :940-950
        hh = s.shape[2]
        outs.append(sft(P, f"controlnet_down_blocks.{k}", s[:, :, :hh // 2, :], s[:, :, -(hh // 2):, :]))
    hh = x.shape[2]
    mid = sft(P, "controlnet_mid_block", x[:, :, :hh // 2, :], x[:, :, -(hh // 2):, :])   # :953-960
    return outs, mid
