"""SDXL AutoencoderKL on MI355X: decode (latents -> image) and encode (image -> LQ latent).

Reference call sites: `pipelines/sdxl_instantir.py:1370-1379` (encode + `latent_dist.sample()` x
scaling_factor) and `:1668-1695` (`latents / scaling_factor` -> `vae.decode`), run there in fp32 after
`upcast_vae` (:984-1001) because the SDXL VAE overflows fp16.  Spec text: `module/diffusers_vae/vae.py:46-350`,
`:771-793`.  Here the activations and weights are **bf16** by default (fp32 exponent range: nothing overflows where
the reference's fp32 does not), with fp32 accumulation, fp32 GroupNorm statistics (merged mean / M2, no
E[x^2] - mean^2) and fp32 attention scores through the softmax.  `dtype=torch.float16` builds the fp16 variant of
the same kernels (3 more mantissa bits, but activations above 65504 become inf: `decode_latent` / `encode` then
raise FloatingPointError instead of returning a black image).  Tolerance against the fp32 oracle is stated in
tests/test_vae_gpu.py; no real SDXL-VAE checkpoint exists offline, so real-weight behaviour is unpinned.

Kernels shared with the UNet path (NHWC): implicit-GEMM conv (3x3, 1x1, nearest-2x folded, the
encoder's bottom/right-padded stride-2 conv), GroupNorm+SiLU, GEMM.  The single-head d = C attention of
the mid block is three GEMMs around a row softmax: S = QK^T/sqrt(C) written as FP32 (`iir_gemm_desc.c_f32`),
P = softmax(S) (`iir_softmax_rows_f32`, fp32 in, 16-bit out), O = P V + b_v (rows of P sum to 1, so the V bias
is added after the product).  T x T scores limit the untiled decode to 16384 latent pixels (1024^2 images); larger
images decode tile by tile (`enable_tiling()`), as BASELINE configs[3] asks.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .config import VAEConfig
from .engine import CPAD, Arena, _NullOps
from .packing import conv_weight_nhwc


class HipVAE:
    def __init__(self, cfg: VAEConfig, sd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16):
        if dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("HipVAE: dtype must be torch.bfloat16 (default) or torch.float16")
        self.cfg, self.device = cfg, torch.device(device)
        self.dtype = dtype
        self.w: Dict[str, torch.Tensor] = {}
        self.arena = Arena(self.device, dtype)
        self.o = ops
        self._gnws = ops.gn_workspace(self.device, 64, cfg.norm_groups)
        self._sized = {}
        self.has_decoder = "decoder.conv_in.weight" in sd
        self.has_encoder = "encoder.conv_in.weight" in sd
        self.tile_sample_size = 1024         # AutoencoderKL config.sample_size (autoencoder_kl.py:117-124)
        self.tile_overlap_factor = 0.25      # autoencoder_kl.py:124
        self.use_tiling = False              # `vae.enable_tiling()` (autoencoder_kl.py:130-143)
        self.use_slicing = False             # `vae.enable_slicing()` (autoencoder_kl.py:145-157)
        self.dtype_name = "bf16" if dtype == torch.bfloat16 else "fp16"
        F16 = dtype          # (the rest of this class allocates its 16-bit tensors as `F16`: the VAE's element type)
        self._E = dtype
        t = lambda n: sd[n].to(device=self.device, dtype=F16)
        for name in sd:
            if not name.endswith(".weight"):
                continue
            p = name[: -len(".weight")]
            w = t(name)
            if w.dim() == 4:
                cin_pad = CPAD if w.shape[1] < CPAD else None          # 3/4/8-channel inputs are zero-padded to one K tile
                if w.shape[2] == 1:
                    k = w.reshape(w.shape[0], w.shape[1])
                    if cin_pad:
                        k = torch.nn.functional.pad(k, (0, cin_pad - k.shape[1]))
                    self.w[p + ".w"] = k.contiguous()
                else:
                    self.w[p + ".w"] = conv_weight_nhwc(w, cin_pad)
                b = t(p + ".bias")
                if w.shape[0] % 4:                                         # conv_out: 3 output channels -> 4
                    padn = 4 - w.shape[0] % 4
                    self.w[p + ".w"] = torch.cat([self.w[p + ".w"], torch.zeros(padn, *self.w[p + ".w"].shape[1:], dtype=F16, device=self.device)]).contiguous()
                    b = torch.cat([b, torch.zeros(padn, dtype=F16, device=self.device)])
                self.w[p + ".b"] = b.contiguous()
            elif w.dim() == 2:
                self.w[p + ".w"] = w.contiguous()
                self.w[p + ".b"] = t(p + ".bias").contiguous()
            else:
                self.w[p + ".g"] = w.contiguous()
                self.w[p + ".b"] = t(p + ".bias").contiguous()

    def enable_tiling(self, use_tiling: bool = True):
        """autoencoder_kl.py:130-136: decode latents larger than `tile_sample_size / 8` per side tile by tile."""
        self.use_tiling = use_tiling

    def disable_tiling(self):
        self.enable_tiling(False)

    def enable_slicing(self):
        """autoencoder_kl.py:145-157: batches are encoded / decoded one image at a time (activation memory of ONE image)."""
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    @property
    def tile_latent_min_size(self):
        return self.tile_sample_size // 8       # sample_size / 2^(len(block_out_channels) - 1), autoencoder_kl.py:123

    # ---- blocks -----------------------------------------------------------------------------------
    def _resnet(self, path, x, R, H, W):
        """ResnetBlock2D without temb, GN eps 1e-6 (module/diffusers_vae/vae.py:241-252 via blocks :2804-2874)."""
        o, w, A, g = self.o, self.w, self.arena, self.cfg.norm_groups
        HW, cin = H * W, x.shape[1]
        cout = w[path + ".conv1.w"].shape[0]
        out = A.alloc(R * HW, cout)
        m = A.mark()
        h = A.alloc(R * HW, cin)
        o.groupnorm(x, h, R, HW, w[path + ".norm1.g"], w[path + ".norm1.b"], 1e-6, True, g, self._gnws)
        h2 = A.alloc(R * HW, cout)
        o.conv2d(h.view(R, H, W, cin), w[path + ".conv1.w"], h2, bias=w[path + ".conv1.b"])
        h3 = A.alloc(R * HW, cout)
        o.groupnorm(h2, h3, R, HW, w[path + ".norm2.g"], w[path + ".norm2.b"], 1e-6, True, g, self._gnws)
        if (path + ".conv_shortcut.w") in w:
            sc = A.alloc(R * HW, cout)
            o.gemm(x, w[path + ".conv_shortcut.w"], sc, bias=w[path + ".conv_shortcut.b"])
        else:
            sc = x
        o.conv2d(h3.view(R, H, W, cout), w[path + ".conv2.w"], out, bias=w[path + ".conv2.b"], res=sc)
        A.release(m)
        return out

    def _attention(self, path, x, R, H, W):
        """Attention(heads=1, dim_head=C, group_norm, bias, residual) -- blocks :776-790, processor :346-412."""
        o, w, A = self.o, self.w, self.arena
        T, C = H * W, x.shape[1]
        out = A.alloc(R * T, C)
        m = A.mark()
        n = A.alloc(R * T, C)
        o.groupnorm(x, n, R, T, w[path + ".group_norm.g"], w[path + ".group_norm.b"], 1e-6, False, self.cfg.norm_groups, self._gnws)
        q, k = A.alloc(R * T, C), A.alloc(R * T, C)
        o.gemm(n, w[path + ".to_q.w"], q, bias=w[path + ".to_q.b"])
        o.gemm(n, w[path + ".to_k.w"], k, bias=w[path + ".to_k.b"])
        a = A.alloc(R * T, C)
        if T > 16384 or T % 4:
            raise ValueError(f"VAE mid-block attention over {T} latent pixels: the untiled path handles up to 16384 (a 1024x1024 "
                             "image); call vae.enable_tiling() for larger images (autoencoder_kl.py:130-136)")
        s32 = A.alloc(T, 2 * T).view(torch.float32)                               # fp32 scores (T, T)
        pr = A.alloc(T, T)                                                        # probabilities, 16-bit
        vt = A.alloc(C, T)
        for r in range(R):
            rows = slice(r * T, (r + 1) * T)
            o.gemm(w[path + ".to_v.w"], n[rows], vt)                               # V^T (bias added after P V)
            o.gemm(q[rows], k[rows], s32, out_scale=C ** -0.5)                    # S = Q K^T / sqrt(C), kept in fp32
            o.softmax_rows_f32(s32, pr)
            o.gemm(pr, vt, a[rows], bias=w[path + ".to_v.b"])                     # O = P V + b_v
        o.gemm(a, w[path + ".to_out.0.w"], out, bias=w[path + ".to_out.0.b"], res=x)
        A.release(m)
        return out

    def _mid(self, path, x, R, H, W):
        x = self._resnet(path + ".resnets.0", x, R, H, W)
        x = self._attention(path + ".attentions.0", x, R, H, W)
        return self._resnet(path + ".resnets.1", x, R, H, W)

    def _run(self, key, fn):
        if key not in self._sized:
            self.arena.buf, self.arena.off, self.arena.high = None, 0, 0
            self.o = _NullOps()
            try:
                fn()
            finally:
                self.o = ops
            self._sized[key] = self.arena.high + 1024
        need = max(self._sized.values())
        if self.arena.buf is None or self.arena.buf.numel() < need:
            self.arena.reserve(need)
        self.arena.reset()
        return fn()

    # ---- decode -----------------------------------------------------------------------------------
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z: (B,4,h,w) fp32 latents ALREADY divided by the scaling factor.  Returns (B,3,8h,8w) fp32."""
        z = z.to(self.device, torch.float32).contiguous()
        B, _, h, wd = z.shape
        out = torch.empty(B, 4, 8 * h, 8 * wd, dtype=torch.float32, device=self.device)
        zin = torch.zeros(B * h * wd, CPAD, dtype=self._E, device=self.device)
        ops.pack_latent(z, zin)
        self._run(("dec", B, h, wd), lambda: self._decode(zin, B, h, wd, out))
        return out[:, :3]

    def _decode(self, zin, R, H, W, out):
        o, w, A, cfg = self.o, self.w, self.arena, self.cfg
        ch = list(reversed(cfg.block_out_channels))
        x0 = A.alloc(R * H * W, CPAD)
        if self.o is ops:
            x0.zero_()
        pq = x0[:, :4]
        o.gemm(zin, w["post_quant_conv.w"], pq, bias=w["post_quant_conv.b"])                   # 1x1, 4 -> 4
        x = A.alloc(R * H * W, ch[0])
        o.conv2d(x0.view(R, H, W, CPAD), w["decoder.conv_in.w"], x, bias=w["decoder.conv_in.b"])
        x = self._mid("decoder.mid_block", x, R, H, W)
        for i, c in enumerate(ch):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, R, H, W)
            if i < len(ch) - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                y = A.alloc(R * 4 * H * W, c)
                o.conv2d(x.view(R, H, W, c), w[p + ".w"], y, upsample=True, bias=w[p + ".b"])
                x, H, W = y, 2 * H, 2 * W
        g = A.alloc(R * H * W, ch[-1])
        o.groupnorm(x, g, R, H * W, w["decoder.conv_norm_out.g"], w["decoder.conv_norm_out.b"], 1e-6, True, cfg.norm_groups, self._gnws)
        img = A.alloc(R * H * W, 4)
        o.conv2d(g.view(R, H, W, ch[-1]), w["decoder.conv_out.w"], img, bias=w["decoder.conv_out.b"])
        o.unpack_latent(img, out)
        return out

    def decode_tiled(self, z: torch.Tensor, sample_size: int = 1024, overlap: float = 0.25) -> torch.Tensor:
        """`AutoencoderKL.tiled_decode` (module/diffusers_vae/autoencoder_kl.py:377-423): overlapping latent tiles of
        sample_size/8 (stride = tile * (1 - overlap)), each decoded on its own, seams blended over sample_size * overlap
        pixels against the already blended upper / left neighbour, tiles cropped to sample_size * (1 - overlap) and
        concatenated.  1024-px tiles / 256-px blends / 768-px crops for SDXL."""
        z = z.to(self.device, torch.float32)
        tl = sample_size // 8
        stride = int(tl * (1 - overlap))
        ext = int(sample_size * overlap)
        limit = sample_size - ext
        rows = []
        for i in range(0, z.shape[2], stride):
            rows.append([self.decode(z[:, :, i:i + tl, j:j + tl].contiguous()).contiguous() for j in range(0, z.shape[3], stride)])
        out_rows = []
        for i, row in enumerate(rows):
            parts = []
            for j, tile in enumerate(row):
                if i > 0:
                    up = rows[i - 1][j]
                    ops.blend_tiles(up, tile, min(up.shape[2], tile.shape[2], ext), True)
                if j > 0:
                    left = row[j - 1]
                    ops.blend_tiles(left, tile, min(left.shape[3], tile.shape[3], ext), False)
                parts.append(tile[:, :, :limit, :limit])
            out_rows.append(torch.cat(parts, dim=3))
        return torch.cat(out_rows, dim=2)

    def decode_latent(self, latents: torch.Tensor, output_type: str = "pt", tiled=None):
        """pipelines/sdxl_instantir.py:1689-1704: latents / scaling_factor -> decode -> postprocess
        (VaeImageProcessor: (x / 2 + 0.5).clamp(0, 1); 'pt' tensor, 'np' NHWC array, 'pil' images).  Tiled when
        `enable_tiling()` is on (or `tiled=True`) and a latent side exceeds the tile (autoencoder_kl.py:270-272)."""
        z = latents.to(self.device, torch.float32) / self.cfg.scaling_factor
        tiled = self.use_tiling if tiled is None else tiled
        big = z.shape[-1] > self.tile_latent_min_size or z.shape[-2] > self.tile_latent_min_size
        one = lambda zz: self.decode_tiled(zz, self.tile_sample_size, self.tile_overlap_factor) if tiled and big else self.decode(zz)
        if self.use_slicing and z.shape[0] > 1:          # autoencoder_kl.py:300-302: every slice takes the (tiled or plain) path on its own
            img = torch.cat([one(zz) for zz in z.split(1)])
        else:
            img = one(z)
        if not torch.isfinite(img).all():
            raise FloatingPointError("VAE decode produced non-finite pixels (activation overflow): this build stores VAE "
                                     f"activations as {self.dtype_name}; see DESIGN.md section 7")
        img = (img / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return img
        arr = img.permute(0, 2, 3, 1).cpu().numpy()
        if output_type == "np":
            return arr
        if output_type == "pil":
            from PIL import Image
            return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
        raise ValueError(f"unknown output_type {output_type}")

    # ---- encode -----------------------------------------------------------------------------------
    def encode(self, image: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        """image (B,3,H,W) in [-1,1]; eps (B,4,H/8,W/8) the N(0,1) draw of `latent_dist.sample()`.
        Returns the UNscaled latent mean + std * eps (fp32)."""
        image = image.to(self.device, torch.float32).contiguous()
        big = image.shape[-1] > self.tile_sample_size or image.shape[-2] > self.tile_sample_size
        if self.use_tiling and big:
            mom = self.moments_tiled(image, self.tile_sample_size, self.tile_overlap_factor)
        elif self.use_slicing and image.shape[0] > 1:    # autoencoder_kl.py:256-258
            mom = torch.cat([self.moments(x1) for x1 in image.split(1)])
        else:
            mom = self.moments(image)
        mean, logvar = mom[:, :4], mom[:, 4:]
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))           # module/diffusers_vae/vae.py:774-777,792
        return mean + std * eps.to(self.device, torch.float32)

    def moments(self, image: torch.Tensor) -> torch.Tensor:
        """encoder + quant_conv: the 8-channel posterior moments (B, 8, H/8, W/8), fp32."""
        image = image.to(self.device, torch.float32).contiguous()
        B, _, H, W = image.shape
        xin = torch.zeros(B * H * W, CPAD, dtype=self._E, device=self.device)
        ops.pack_latent(image, xin)
        mom = torch.empty(B, 8, H // 8, W // 8, dtype=torch.float32, device=self.device)
        self._run(("enc", B, H, W), lambda: self._encode(xin, B, H, W, mom))
        if not torch.isfinite(mom).all():
            raise FloatingPointError(f"VAE encode produced non-finite moments (activation overflow in the {self.dtype_name} build)")
        return mom

    def moments_tiled(self, image: torch.Tensor, sample_size: int = 1024, overlap: float = 0.25) -> torch.Tensor:
        """`AutoencoderKL.tiled_encode` (module/diffusers_vae/autoencoder_kl.py:323-375): overlapping pixel tiles of `sample_size`
        (stride 768 px for SDXL), each encoded on its own; the MOMENTS are blended over tile_latent * overlap (32) rows /
        columns against the already blended upper / left neighbour, cropped to 96 and concatenated."""
        tl = sample_size // 8
        stride = int(sample_size * (1 - overlap))
        ext = int(tl * overlap)
        limit = tl - ext
        rows = []
        for i in range(0, image.shape[2], stride):
            rows.append([self.moments(image[:, :, i:i + sample_size, j:j + sample_size].contiguous()).contiguous()
                         for j in range(0, image.shape[3], stride)])
        out_rows = []
        for i, row in enumerate(rows):
            parts = []
            for j, tile in enumerate(row):
                if i > 0:
                    up = rows[i - 1][j]
                    ops.blend_tiles(up, tile, min(up.shape[2], tile.shape[2], ext), True)
                if j > 0:
                    left = row[j - 1]
                    ops.blend_tiles(left, tile, min(left.shape[3], tile.shape[3], ext), False)
                parts.append(tile[:, :, :limit, :limit])
            out_rows.append(torch.cat(parts, dim=3))
        return torch.cat(out_rows, dim=2)

    def encode_to_latent(self, image, eps=None, generator=None):
        """`vae.encode(image).latent_dist.sample() * scaling_factor` (pipelines/sdxl_instantir.py:1375-1376)."""
        B, _, H, W = image.shape
        if eps is None:
            gdev = generator.device if generator is not None else self.device
            eps = torch.randn(B, 4, H // 8, W // 8, generator=generator, device=gdev, dtype=torch.float32)
        return self.encode(image, eps) * self.cfg.scaling_factor

    def _encode(self, xin, R, H, W, mom):
        o, w, A, cfg = self.o, self.w, self.arena, self.cfg
        ch = list(cfg.block_out_channels)
        x = A.alloc(R * H * W, ch[0])
        o.conv2d(xin.view(R, H, W, CPAD), w["encoder.conv_in.w"], x, bias=w["encoder.conv_in.b"])
        for i, c in enumerate(ch):
            for j in range(cfg.layers_per_block):
                x = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", x, R, H, W)
            if i < len(ch) - 1:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                y = A.alloc(R * (H // 2) * (W // 2), c)
                o.conv2d(x.view(R, H, W, c), w[p + ".w"], y, stride=2, pad_mode=1, bias=w[p + ".b"])   # F.pad(0,1,0,1) + conv s2 p0
                x, H, W = y, H // 2, W // 2
        x = self._mid("encoder.mid_block", x, R, H, W)
        g = A.alloc(R * H * W, ch[-1])
        o.groupnorm(x, g, R, H * W, w["encoder.conv_norm_out.g"], w["encoder.conv_norm_out.b"], 1e-6, True, cfg.norm_groups, self._gnws)
        m8 = A.alloc(R * H * W, CPAD)
        if self.o is ops:
            m8.zero_()
        o.conv2d(g.view(R, H, W, ch[-1]), w["encoder.conv_out.w"], m8[:, :8], bias=w["encoder.conv_out.b"])
        q8 = A.alloc(R * H * W, 8)
        o.gemm(m8, w["quant_conv.w"], q8, bias=w["quant_conv.b"])                                  # 1x1, 8 -> 8
        o.unpack_latent(q8, mom)
        return mom
