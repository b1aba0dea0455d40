"""Static shape descriptions of the networks on the InstantIR denoising path.

The reference hard-wires these numbers in third-party configs (SDXL-base `unet/config.json`)
and in `module/min_sdxl.py:803-840` (320/640/1280 channels, depths 0/2/10, mid 10, head_dim 64),
`module/ip_adapter/utils.py:138-152` (Resampler 1024 -> 1280 x4 layers -> 2048, 64 queries) and
`module/aggregator.py:229-270`.  `tiny()` keeps every structural feature (concat skips, 3 levels,
cross-attn with ragged KV lengths, IP tokens, adaLN, SFT heads) at sizes the CPU oracle finishes
in seconds; it is the parity-test geometry, never a bench geometry.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class ResamplerConfig:
    dim: int = 1280            # module/ip_adapter/utils.py:139
    depth: int = 4
    dim_head: int = 64
    heads: int = 20
    num_queries: int = 64
    embedding_dim: int = 1024  # DINOv2-L hidden size
    output_dim: int = 2048     # = unet cross_attention_dim
    ff_mult: int = 4
    seq_len: int = 257         # DINOv2 tokens @224^2


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    transformer_depth: Tuple[int, ...] = (0, 2, 10)   # per down block; mid uses the last entry
    layers_per_block: int = 2
    head_dim: int = 64
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    pooled_dim: int = 1280          # text_encoder_2 projection dim
    text_len: int = 77
    num_ip_tokens: int = 64
    norm_groups: int = 32
    resampler: ResamplerConfig = field(default_factory=ResamplerConfig)
    lora_rank: int = 64             # pipelines/sdxl_instantir.py:376-381
    sft_hidden: int = 128           # module/aggregator.py:60

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def add_embed_in(self) -> int:
        # text_embeds (pooled) ++ 6 sinusoid-embedded time ids (pipelines/sdxl_instantir.py:965-981)
        return self.pooled_dim + 6 * self.addition_time_embed_dim

    @property
    def mid_depth(self) -> int:
        return self.transformer_depth[-1]

    @staticmethod
    def sdxl() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def tiny() -> "UNetConfig":
        return UNetConfig(
            block_out_channels=(64, 128, 256),
            transformer_depth=(0, 1, 2),
            cross_attention_dim=128,
            addition_time_embed_dim=32,
            pooled_dim=64,
            text_len=13,
            num_ip_tokens=16,
            resampler=ResamplerConfig(dim=128, depth=2, dim_head=64, heads=2, num_queries=16,
                                      embedding_dim=64, output_dim=128, ff_mult=4, seq_len=21),
            lora_rank=8,
            sft_hidden=64,
        )


@dataclass(frozen=True)
class VAEConfig:
    """SDXL AutoencoderKL decoder/encoder geometry (spec: module/diffusers_vae/vae.py:46-350)."""
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.13025

    @staticmethod
    def sdxl() -> "VAEConfig":
        return VAEConfig()

    @staticmethod
    def tiny() -> "VAEConfig":
        return VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1)
