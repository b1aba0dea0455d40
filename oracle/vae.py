"""Oracle (test infrastructure): fp32 CPU restatement of the SDXL AutoencoderKL decode / encode used by
the pipeline (pipelines/sdxl_instantir.py:1370-1379 encode, :1668-1695 decode).

Spec text: module/diffusers_vae/vae.py:46-195 (Encoder), :197-350 (Decoder), :771-793 (posterior
sampling); block wrappers module/unet/unet_2d_ZeroSFT_blocks.py:679-831 (UNetMidBlock2D), :1498-1574
(DownEncoderBlock2D), :2804-2874 (UpDecoderBlock2D); attention module/ip_adapter/attention_processor.py:337-414
with group_norm, bias, 1 head of dim C, residual connection.  PARITY UNPINNED (diffusers absent;
no reference fixture).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .nets import conv2d, group_norm, linear


def vae_resnet(P, path, x, groups):
    """ResnetBlock2D without time embedding, eps 1e-6 (vae.py:241-252)."""
    h = F.silu(group_norm(P, path + ".norm1", x, groups, 1e-6))
    h = conv2d(P, path + ".conv1", h)
    h = F.silu(group_norm(P, path + ".norm2", h, groups, 1e-6))
    h = conv2d(P, path + ".conv2", h)
    if (path + ".conv_shortcut.weight") in P:
        x = conv2d(P, path + ".conv_shortcut", x, padding=0)
    return x + h


def vae_attention(P, path, x, groups):
    """Attention(heads=1, dim_head=C, norm_num_groups, bias=True, residual_connection=True) run by
    AttnProcessor2_0 on a 4-D input (attention_processor.py:346-412)."""
    b, c, hh, ww = x.shape
    h = x.reshape(b, c, hh * ww).transpose(1, 2)
    h = F.group_norm(h.transpose(1, 2), groups, P[path + ".group_norm.weight"], P[path + ".group_norm.bias"], 1e-6).transpose(1, 2)
    q, k, v = (linear(P, f"{path}.{n}", h) for n in ("to_q", "to_k", "to_v"))
    s = torch.softmax(q @ k.transpose(-1, -2) / c ** 0.5, dim=-1)
    o = linear(P, path + ".to_out.0", s @ v)
    return o.transpose(-1, -2).reshape(b, c, hh, ww) + x


def _mid(P, path, x, groups):
    x = vae_resnet(P, path + ".resnets.0", x, groups)
    x = vae_attention(P, path + ".attentions.0", x, groups)
    return vae_resnet(P, path + ".resnets.1", x, groups)


def decode(P, vc, z):
    """AutoencoderKL.decode: post_quant_conv then Decoder.forward (vae.py:285-350).  `z` already divided
    by the scaling factor (pipelines/sdxl_instantir.py:1689)."""
    g = vc.norm_groups
    x = conv2d(P, "post_quant_conv", z, padding=0)
    x = conv2d(P, "decoder.conv_in", x)
    x = _mid(P, "decoder.mid_block", x, g)
    ch = list(reversed(vc.block_out_channels))
    for i in range(len(ch)):
        for j in range(vc.layers_per_block + 1):
            x = vae_resnet(P, f"decoder.up_blocks.{i}.resnets.{j}", x, g)
        if i < len(ch) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv2d(P, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(group_norm(P, "decoder.conv_norm_out", x, g, 1e-6))
    return conv2d(P, "decoder.conv_out", x)


def encode(P, vc, img, eps):
    """AutoencoderKL.encode(...).latent_dist.sample() with the N(0,1) draw passed explicitly
    (vae.py:137-195 Encoder.forward; :771-793 posterior; the reference draws from the global RNG,
    SURVEY Appendix B item 1).  Returns the UNscaled latent."""
    return sample_moments(moments(P, vc, img), eps)


def sample_moments(mom, eps):
    mean, logvar = torch.chunk(mom, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return mean + std * eps


def encode_tiled(P, vc, img, eps, sample_size=1024, overlap=0.25):
    """AutoencoderKL.tiled_encode, module/diffusers_vae/autoencoder_kl.py:323-375: overlapping PIXEL tiles of `sample_size`
    (stride sample_size * (1 - overlap)), each through encoder + quant_conv on its own; the 8-channel MOMENTS are blended over
    tile_latent * overlap rows / columns against the upper / left neighbour, cropped to tile_latent * (1 - overlap) and
    concatenated; the posterior is sampled from the blended moments."""
    tl = sample_size // 8
    stride = int(sample_size * (1 - overlap))
    ext = int(tl * overlap)
    limit = tl - ext
    rows = [[moments(P, vc, img[:, :, i:i + sample_size, j:j + sample_size]) for j in range(0, img.shape[3], stride)]
            for i in range(0, img.shape[2], stride)]
    return sample_moments(_blend_rows(rows, ext, limit), eps)


def moments(P, vc, img):
    g = vc.norm_groups
    x = conv2d(P, "encoder.conv_in", img)
    n = len(vc.block_out_channels)
    for i in range(n):
        for j in range(vc.layers_per_block):
            x = vae_resnet(P, f"encoder.down_blocks.{i}.resnets.{j}", x, g)
        if i < n - 1:
            x = F.pad(x, (0, 1, 0, 1))                       # Downsample2D(padding=0): vae.py:110 / blocks :1560
            x = conv2d(P, f"encoder.down_blocks.{i}.downsamplers.0.conv", x, stride=2, padding=0)
    x = _mid(P, "encoder.mid_block", x, g)
    x = F.silu(group_norm(P, "encoder.conv_norm_out", x, g, 1e-6))
    x = conv2d(P, "encoder.conv_out", x)
    return conv2d(P, "quant_conv", x, padding=0)


def decode_tiled(P, vc, z, sample_size=1024, overlap=0.25):
    """AutoencoderKL.tiled_decode + blend_v / blend_h, module/diffusers_vae/autoencoder_kl.py:311-321,377-423."""
    tl = sample_size // 8
    stride, ext = int(tl * (1 - overlap)), int(sample_size * overlap)
    limit = sample_size - ext
    rows = [[decode(P, vc, z[:, :, i:i + tl, j:j + tl]) for j in range(0, z.shape[3], stride)] for i in range(0, z.shape[2], stride)]
    return _blend_rows(rows, ext, limit)


def _blend_rows(rows, ext, limit):
    """blend_v / blend_h against the already blended upper / left tile, crop, concatenate (autoencoder_kl.py:311-321,359-371,406-418)."""
    def blend_v(a, b, e):
        e = min(a.shape[2], b.shape[2], e)
        for y in range(e):
            b[:, :, y, :] = a[:, :, -e + y, :] * (1 - y / e) + b[:, :, y, :] * (y / e)
        return b

    def blend_h(a, b, e):
        e = min(a.shape[3], b.shape[3], e)
        for x in range(e):
            b[:, :, :, x] = a[:, :, :, -e + x] * (1 - x / e) + b[:, :, :, x] * (x / e)
        return b

    out = []
    for i, row in enumerate(rows):
        parts = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = blend_v(rows[i - 1][j], tile, ext)
            if j > 0:
                tile = blend_h(row[j - 1], tile, ext)
            parts.append(tile[:, :, :limit, :limit])
        out.append(torch.cat(parts, dim=3))
    return torch.cat(out, dim=2)
