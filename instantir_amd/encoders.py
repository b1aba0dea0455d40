"""Image / text encoders that feed the denoising loop once per batch, on the same HIP kernels.

* `HipDinov2`: the DINOv2 ViT the reference loads with `AutoModel.from_pretrained(dinov2)`
  (`module/ip_adapter/utils.py:106-118`) and runs in `encode_image` on the image AND on `zeros_like(image)`
  (`pipelines/sdxl_instantir.py:660-667`), returning `last_hidden_state` (B, 257, 1024 @ 224 px).
  Architecture = transformers `Dinov2Model` (third-party, `transformers==4.36.2` pinned by requirements.txt:2):
  14x14/14 patch conv + CLS + bicubic-resized position table, pre-LN blocks (MHA with biases, LayerScale, GELU
  MLP), final LayerNorm.  Parameter names are the Hugging Face ones.

MI355X mapping: tokens of an image are padded to a multiple of 8 rows (257 -> 264) so every matrix keeps the
alignment the kernels want; pad rows are computed and ignored (keys beyond 257 are masked in the attention
kernel).  LayerScale is folded into the output projections (W' = diag(lambda) W), the value bias into the
projection bias (softmax rows sum to 1: (P(V + 1 b^T)) Wo^T = (P V) Wo^T + Wo b), Q|K are one GEMM, V is produced
transposed by swapping GEMM operands.  The zero-image branch is input independent and cached per geometry
(SURVEY.md Appendix C Q7).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops

F16 = torch.float16


class HipDinov2:
    def __init__(self, sd: Dict[str, torch.Tensor], device, patch_size=14, num_heads=None, eps=1e-6):
        self.device = torch.device(device)
        dev = self.device
        f32 = lambda n: sd[n].to(dev, torch.float32)
        self.patch = patch_size
        wp = f32("embeddings.patch_embeddings.projection.weight")            # (D, 3, p, p)
        self.D = D = wp.shape[0]
        self.heads = num_heads if num_heads is not None else D // 64
        if D // self.heads != 64:
            raise ValueError("HipDinov2 uses the head_dim-64 attention kernel (ViT-S/B/L/g all have 64)")
        kp = wp[0].numel()
        self.kpad = (kp + 63) // 64 * 64
        self.w = {}
        self.w["patch.w"] = torch.nn.functional.pad(wp.reshape(D, kp), (0, self.kpad - kp)).to(F16).contiguous()
        self.w["patch.b"] = f32("embeddings.patch_embeddings.projection.bias").to(F16)
        self.cls = f32("embeddings.cls_token").reshape(D)
        self.pos = f32("embeddings.position_embeddings")                      # (1, 1 + n^2, D)
        self.eps = eps
        self.depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layer."))
        for i in range(self.depth):
            p = f"encoder.layer.{i}"
            a = p + ".attention.attention"
            self.w[f"{i}.n1.g"], self.w[f"{i}.n1.b"] = f32(p + ".norm1.weight").to(F16), f32(p + ".norm1.bias").to(F16)
            self.w[f"{i}.n2.g"], self.w[f"{i}.n2.b"] = f32(p + ".norm2.weight").to(F16), f32(p + ".norm2.bias").to(F16)
            self.w[f"{i}.qk.w"] = torch.cat([f32(a + ".query.weight"), f32(a + ".key.weight")]).to(F16).contiguous()
            self.w[f"{i}.qk.b"] = torch.cat([f32(a + ".query.bias"), f32(a + ".key.bias")]).to(F16).contiguous()
            self.w[f"{i}.v.w"] = f32(a + ".value.weight").to(F16).contiguous()
            ls1, ls2 = f32(p + ".layer_scale1.lambda1"), f32(p + ".layer_scale2.lambda1")
            wo, bo = f32(p + ".attention.output.dense.weight"), f32(p + ".attention.output.dense.bias")
            self.w[f"{i}.o.w"] = (ls1[:, None] * wo).to(F16).contiguous()
            self.w[f"{i}.o.b"] = (ls1 * (wo @ f32(a + ".value.bias") + bo)).to(F16).contiguous()
            self.w[f"{i}.fc1.w"], self.w[f"{i}.fc1.b"] = f32(p + ".mlp.fc1.weight").to(F16).contiguous(), f32(p + ".mlp.fc1.bias").to(F16)
            self.w[f"{i}.fc2.w"] = (ls2[:, None] * f32(p + ".mlp.fc2.weight")).to(F16).contiguous()
            self.w[f"{i}.fc2.b"] = (ls2 * f32(p + ".mlp.fc2.bias")).to(F16).contiguous()
        self.w["ln.g"], self.w["ln.b"] = f32("layernorm.weight").to(F16), f32("layernorm.bias").to(F16)
        self._pos_cache = {}
        self._zero_cache = {}

    def _pos_table(self, gh, gw):
        """Position table for a gh x gw patch grid: bicubic resize of the stored square table (one-time weight
        preprocessing, same call as transformers' Dinov2Embeddings.interpolate_pos_encoding)."""
        key = (gh, gw)
        if key not in self._pos_cache:
            n = self.pos.shape[1] - 1
            side = int(round(n ** 0.5))
            if gh * gw == n and gh == gw:
                table = self.pos[0]
            else:
                pp = self.pos[:, 1:].reshape(1, side, side, self.D).permute(0, 3, 1, 2)
                pp = torch.nn.functional.interpolate(pp, size=(gh, gw), mode="bicubic", align_corners=False)
                table = torch.cat([self.pos[0, :1], pp.permute(0, 2, 3, 1).reshape(gh * gw, self.D)])
            self._pos_cache[key] = (table[1:].to(F16).contiguous(), (self.cls + table[0]).to(F16))
        return self._pos_cache[key]

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values (B, 3, H, W), already normalised (AutoImageProcessor).  Returns last_hidden_state
        (B, 1 + (H/14)(W/14), D) fp16."""
        dev, D, ps, w = self.device, self.D, self.patch, self.w
        x = pixel_values.to(dev, torch.float32)
        B, _, H, W = x.shape
        gh, gw = H // ps, W // ps
        T = 1 + gh * gw
        Tp = (T + 7) // 8 * 8
        M = B * Tp
        # patchify (pure data movement): (B, 3, gh, p, gw, p) -> (B*gh*gw, 3*p*p) zero-padded to a K tile multiple
        pt = x[:, :, :gh * ps, :gw * ps].reshape(B, 3, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, 3 * ps * ps)
        patches = torch.zeros(B * gh * gw, self.kpad, dtype=F16, device=dev)
        patches[:, :3 * ps * ps] = pt.to(F16)
        pos_patch, cls_row = self._pos_table(gh, gw)
        h = torch.zeros(M, D, dtype=F16, device=dev)
        h3 = h.view(B, Tp, D)
        h3[:, 0] = cls_row                                               # CLS token + its position row (constants)
        for b in range(B):                                              # patch projection + bias + position rows
            ops.gemm(patches[b * gh * gw:(b + 1) * gh * gw], w["patch.w"], h3[b, 1:T], bias=w["patch.b"], res=pos_patch)
        n = torch.empty(M, D, dtype=F16, device=dev)
        qk = torch.empty(M, 2 * D, dtype=F16, device=dev)
        vt = torch.empty(D, M, dtype=F16, device=dev)
        a = torch.empty(M, D, dtype=F16, device=dev)
        f = torch.empty(M, w["0.fc1.w"].shape[0], dtype=F16, device=dev)
        for i in range(self.depth):
            ops.layernorm(h, n, w[f"{i}.n1.g"], w[f"{i}.n1.b"], self.eps)
            ops.gemm(n, w[f"{i}.qk.w"], qk, bias=w[f"{i}.qk.b"])
            ops.gemm(w[f"{i}.v.w"], n, vt)                               # V^T (value bias folded into o.b)
            ops.attention(qk[:, :D], a, [(qk[:, D:], Tp, vt, Tp, T)], B, self.heads, Tp)
            ops.gemm(a, w[f"{i}.o.w"], h, bias=w[f"{i}.o.b"], res=h)      # LayerScale folded
            ops.layernorm(h, n, w[f"{i}.n2.g"], w[f"{i}.n2.b"], self.eps)
            ops.gemm(n, w[f"{i}.fc1.w"], f, bias=w[f"{i}.fc1.b"], act=ops.ACT_GELU)
            ops.gemm(f, w[f"{i}.fc2.w"], h, bias=w[f"{i}.fc2.b"], res=h)
        ops.layernorm(h, n, w["ln.g"], w["ln.b"], self.eps)
        return n.view(B, Tp, D)[:, :T]

    __call__ = forward

    def encode_image_pair(self, pixel_values: torch.Tensor):
        """`encode_image` for the DINO branch (pipelines/sdxl_instantir.py:660-667): features of the image and of
        `zeros_like(image)`; the latter depends only on the geometry and is cached."""
        feats = self.forward(pixel_values)
        key = tuple(pixel_values.shape[1:])
        if key not in self._zero_cache:
            self._zero_cache[key] = self.forward(torch.zeros(1, *key))
        zero = self._zero_cache[key].expand(pixel_values.shape[0], -1, -1)
        return feats, zero


def dinov2_preprocess(images, size=256, crop=224, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """AutoImageProcessor of facebook/dinov2-* (BitImageProcessor): bicubic resize of the shortest edge to 256,
    centre crop 224, rescale 1/255, ImageNet normalise (SURVEY.md Appendix A "Image I/O").  PIL in, fp32 NCHW out."""
    import numpy as np
    from PIL import Image
    out = []
    for im in images if isinstance(images, (list, tuple)) else [images]:
        im = im.convert("RGB")
        w0, h0 = im.size
        s = size / min(w0, h0)
        im = im.resize((max(crop, round(w0 * s)), max(crop, round(h0 * s))), Image.BICUBIC)
        l, t = (im.size[0] - crop) // 2, (im.size[1] - crop) // 2
        arr = np.asarray(im.crop((l, t, l + crop, t + crop)), dtype=np.float32) / 255.0
        out.append((arr - np.array(mean, dtype=np.float32)) / np.array(std, dtype=np.float32))
    return torch.from_numpy(np.stack(out)).permute(0, 3, 1, 2).contiguous()


def clip_preprocess(images, size=224, mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)):
    """`CLIPImageProcessor()` with its defaults (module/ip_adapter/utils.py:114-118): bicubic resize of the shortest edge to
    224, centre crop 224, rescale 1/255, CLIP normalise.  PIL in, fp32 NCHW out."""
    return dinov2_preprocess(images, size=size, crop=size, mean=mean, std=std)


class HipCLIPVision:
    """CLIP vision tower (transformers `CLIPVisionModelWithProjection`, third-party) for the reference's `use_clip_encoder`
    branch (`module/ip_adapter/utils.py:106-118`).  With a Resampler as the image projector the pipeline asks for
    `hidden_states[-2]` of the image and of `zeros_like(image)` (`pipelines/sdxl_instantir.py:696-699,644-654`); `image_embeds`
    (post-LayerNorm CLS feature through `visual_projection`) is provided as well (:656-659).

    Patch conv without bias + class embedding + learned position table (fixed geometry), `pre_layrnorm` (sic, the HF
    parameter name), pre-LN blocks with unmasked attention, quick-GELU or GELU MLP.  Token rows are padded to a multiple
    of 8 (257 -> 264, pad keys masked).  Heads of dim 64 only (ViT-B/16, ViT-B/32, ViT-L/14): the attention kernel is a
    d = 64 kernel, so ViT-H/14 (d = 80) and bigG (d = 104) are refused with that reason."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, patch_size=None, num_heads=None, eps=1e-5, hidden_act="quick_gelu"):
        self.device = torch.device(device)
        dev = self.device
        sd = {(k[len("vision_model."):] if k.startswith("vision_model.") else k): v for k, v in sd.items()}
        f32 = lambda n: sd[n].to(dev, torch.float32)
        wp = f32("embeddings.patch_embedding.weight")                    # (D, 3, p, p), no bias
        self.D = D = wp.shape[0]
        self.patch = wp.shape[-1] if patch_size is None else patch_size
        self.heads = num_heads if num_heads is not None else D // 64
        if D // self.heads != 64:
            raise ValueError(f"HipCLIPVision: head_dim {D // self.heads} -- the attention kernel is built for head_dim 64 "
                             "(CLIP ViT-B / ViT-L); ViT-H/14 and bigG towers are not supported")
        kp = wp[0].numel()
        self.kpad = (kp + 63) // 64 * 64
        self.act = {"quick_gelu": ops.ACT_QUICKGELU, "gelu": ops.ACT_GELU}[hidden_act]
        self.eps = eps
        w = self.w = {}
        w["patch.w"] = torch.nn.functional.pad(wp.reshape(D, kp), (0, self.kpad - kp)).to(F16).contiguous()
        pos = f32("embeddings.position_embedding.weight")                # (1 + n^2, D)
        self.n_pos = pos.shape[0]
        w["pos"] = pos[1:].to(F16).contiguous()
        w["cls"] = (f32("embeddings.class_embedding").reshape(D) + pos[0]).to(F16)
        w["pre.g"], w["pre.b"] = f32("pre_layrnorm.weight").to(F16), f32("pre_layrnorm.bias").to(F16)
        self.depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
        for i in range(self.depth):
            p = f"encoder.layers.{i}"
            a = p + ".self_attn"
            w[f"{i}.n1.g"], w[f"{i}.n1.b"] = f32(p + ".layer_norm1.weight").to(F16), f32(p + ".layer_norm1.bias").to(F16)
            w[f"{i}.n2.g"], w[f"{i}.n2.b"] = f32(p + ".layer_norm2.weight").to(F16), f32(p + ".layer_norm2.bias").to(F16)
            w[f"{i}.qk.w"] = torch.cat([f32(a + ".q_proj.weight"), f32(a + ".k_proj.weight")]).to(F16).contiguous()
            w[f"{i}.qk.b"] = torch.cat([f32(a + ".q_proj.bias"), f32(a + ".k_proj.bias")]).to(F16).contiguous()
            w[f"{i}.v.w"] = f32(a + ".v_proj.weight").to(F16).contiguous()
            wo = f32(a + ".out_proj.weight")
            w[f"{i}.o.w"] = wo.to(F16).contiguous()
            w[f"{i}.o.b"] = (wo @ f32(a + ".v_proj.bias") + f32(a + ".out_proj.bias")).to(F16).contiguous()   # value bias folded
            w[f"{i}.fc1.w"], w[f"{i}.fc1.b"] = f32(p + ".mlp.fc1.weight").to(F16).contiguous(), f32(p + ".mlp.fc1.bias").to(F16)
            w[f"{i}.fc2.w"], w[f"{i}.fc2.b"] = f32(p + ".mlp.fc2.weight").to(F16).contiguous(), f32(p + ".mlp.fc2.bias").to(F16)
        w["post.g"], w["post.b"] = f32("post_layernorm.weight").to(F16), f32("post_layernorm.bias").to(F16)
        self.proj = sd["visual_projection.weight"].to(dev, F16).contiguous() if "visual_projection.weight" in sd else None
        self._zero_cache = {}

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, with_embeds: bool = False):
        """pixel_values (B, 3, H, W) normalised (CLIPImageProcessor).  Returns hidden_states[-2] (B, 1 + (H/p)(W/p), D) fp16
        [and image_embeds (B, P) when `with_embeds`]."""
        dev, D, ps, w = self.device, self.D, self.patch, self.w
        x = pixel_values.to(dev, torch.float32)
        B, _, H, W = x.shape
        gh, gw = H // ps, W // ps
        T = 1 + gh * gw
        if T != self.n_pos:
            raise ValueError(f"CLIP vision tower has a fixed position table of {self.n_pos} tokens; a {H}x{W} input gives {T}")
        Tp = (T + 7) // 8 * 8
        M = B * Tp
        pt = x[:, :, :gh * ps, :gw * ps].reshape(B, 3, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, 3 * ps * ps)
        patches = torch.zeros(B * gh * gw, self.kpad, dtype=F16, device=dev)
        patches[:, :3 * ps * ps] = pt.to(F16)
        e = torch.zeros(M, D, dtype=F16, device=dev)
        e3 = e.view(B, Tp, D)
        e3[:, 0] = w["cls"]
        for b in range(B):                                               # patch projection (no bias) + position rows
            ops.gemm(patches[b * gh * gw:(b + 1) * gh * gw], w["patch.w"], e3[b, 1:T], res=w["pos"])
        h = torch.empty(M, D, dtype=F16, device=dev)
        ops.layernorm(e, h, w["pre.g"], w["pre.b"], self.eps)             # hidden_states[0]
        n = torch.empty(M, D, dtype=F16, device=dev)
        qk = torch.empty(M, 2 * D, dtype=F16, device=dev)
        vt = torch.empty(D, M, dtype=F16, device=dev)
        a = torch.empty(M, D, dtype=F16, device=dev)
        f = torch.empty(M, w["0.fc1.w"].shape[0], dtype=F16, device=dev)
        penult = None
        for i in range(self.depth if with_embeds else self.depth - 1):   # hidden_states[-2] is the INPUT of the last layer
            if i == self.depth - 1:
                penult = h.clone()
            ops.layernorm(h, n, w[f"{i}.n1.g"], w[f"{i}.n1.b"], self.eps)
            ops.gemm(n, w[f"{i}.qk.w"], qk, bias=w[f"{i}.qk.b"])
            ops.gemm(w[f"{i}.v.w"], n, vt)
            ops.attention(qk[:, :D], a, [(qk[:, D:], Tp, vt, Tp, T)], B, self.heads, Tp)
            ops.gemm(a, w[f"{i}.o.w"], h, bias=w[f"{i}.o.b"], res=h)
            ops.layernorm(h, n, w[f"{i}.n2.g"], w[f"{i}.n2.b"], self.eps)
            ops.gemm(n, w[f"{i}.fc1.w"], f, bias=w[f"{i}.fc1.b"], act=self.act)
            ops.gemm(f, w[f"{i}.fc2.w"], h, bias=w[f"{i}.fc2.b"], res=h)
        if not with_embeds:
            return h.view(B, Tp, D)[:, :T]
        if self.proj is None:
            raise ValueError("image_embeds need `visual_projection.weight` (a CLIPVisionModelWithProjection state dict)")
        rows8 = torch.zeros((B + 7) // 8 * 8, D, dtype=F16, device=dev)
        rows8[:B] = h.view(B, Tp, D)[:, 0]                                # pooled = CLS row
        pn = torch.empty_like(rows8)
        ops.layernorm(rows8, pn, w["post.g"], w["post.b"], self.eps)
        out = torch.empty(rows8.shape[0], self.proj.shape[0], dtype=F16, device=dev)
        ops.gemm(pn, self.proj, out)
        return penult.view(B, Tp, D)[:, :T], out[:B]

    __call__ = forward

    def encode_image_pair(self, pixel_values: torch.Tensor):
        """`encode_image(..., output_hidden_states=True)` (pipelines/sdxl_instantir.py:644-654): penultimate hidden states of
        the image and of `zeros_like(image)` (input independent, cached per geometry)."""
        feats = self.forward(pixel_values)
        key = tuple(pixel_values.shape[1:])
        if key not in self._zero_cache:
            self._zero_cache[key] = self.forward(torch.zeros(1, *key))
        return feats, self._zero_cache[key].expand(pixel_values.shape[0], -1, -1)


class HipCLIPText:
    """CLIP text transformer (transformers `CLIPTextModel` / `CLIPTextModelWithProjection`, third-party) as used by
    `encode_prompt` (pipelines/sdxl_instantir.py:400-632): SDXL takes `hidden_states[-2]` of both encoders (768 + 1280
    features, concatenated to the 2048-wide context) and the projected EOS feature of the second one (`text_embeds`).

    Pre-LN blocks with causal self-attention (heads of dim 64: 12 for CLIP-L, 20 for OpenCLIP-bigG), quick-GELU or GELU
    MLP.  Sequences are padded from 77 to 80 rows; the causal mask also hides the pad keys from every real query."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, hidden_act="quick_gelu", eos_token_id=2, eps=1e-5):
        self.device = torch.device(device)
        dev = self.device
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}
        f32 = lambda n: sd[n].to(dev, torch.float32)
        self.tok = f32("embeddings.token_embedding.weight").to(F16)
        self.pos = f32("embeddings.position_embedding.weight").to(F16)
        self.D = D = self.tok.shape[1]
        self.heads = D // 64
        self.act = {"quick_gelu": ops.ACT_QUICKGELU, "gelu": ops.ACT_GELU}[hidden_act]
        self.eos_token_id, self.eps = eos_token_id, eps
        self.depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
        self.w = {}
        for i in range(self.depth):
            p = f"encoder.layers.{i}"
            a = p + ".self_attn"
            self.w[f"{i}.n1.g"], self.w[f"{i}.n1.b"] = f32(p + ".layer_norm1.weight").to(F16), f32(p + ".layer_norm1.bias").to(F16)
            self.w[f"{i}.n2.g"], self.w[f"{i}.n2.b"] = f32(p + ".layer_norm2.weight").to(F16), f32(p + ".layer_norm2.bias").to(F16)
            self.w[f"{i}.qk.w"] = torch.cat([f32(a + ".q_proj.weight"), f32(a + ".k_proj.weight")]).to(F16).contiguous()
            self.w[f"{i}.qk.b"] = torch.cat([f32(a + ".q_proj.bias"), f32(a + ".k_proj.bias")]).to(F16).contiguous()
            self.w[f"{i}.v.w"] = f32(a + ".v_proj.weight").to(F16).contiguous()
            wo = f32(a + ".out_proj.weight")
            self.w[f"{i}.o.w"] = wo.to(F16).contiguous()
            self.w[f"{i}.o.b"] = (wo @ f32(a + ".v_proj.bias") + f32(a + ".out_proj.bias")).to(F16).contiguous()   # value bias folded
            self.w[f"{i}.fc1.w"], self.w[f"{i}.fc1.b"] = f32(p + ".mlp.fc1.weight").to(F16).contiguous(), f32(p + ".mlp.fc1.bias").to(F16)
            self.w[f"{i}.fc2.w"], self.w[f"{i}.fc2.b"] = f32(p + ".mlp.fc2.weight").to(F16).contiguous(), f32(p + ".mlp.fc2.bias").to(F16)
        self.w["ln.g"], self.w["ln.b"] = f32("final_layer_norm.weight").to(F16), f32("final_layer_norm.bias").to(F16)
        self.proj = sd["text_projection.weight"].to(dev, F16).contiguous() if "text_projection.weight" in sd else None

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, clip_skip=None):
        """input_ids (B, T<=77) int64.  Returns (hidden_states[-2] (B, T, D) fp16 -- hidden_states[-(clip_skip + 2)] when
        `clip_skip` is given, pipelines/sdxl_instantir.py:524-530 -- and text_embeds (B, P) fp16 or None)."""
        take = self.depth - 1 - (clip_skip or 0)            # the kept state is the INPUT of layer `take`
        if not 0 <= take < self.depth:
            raise ValueError(f"clip_skip={clip_skip} out of range for a {self.depth}-layer text encoder")
        dev, D, w = self.device, self.D, self.w
        ids = input_ids.to(dev)
        B, T = ids.shape
        Tp = (T + 7) // 8 * 8
        M = B * Tp
        emb = torch.zeros(M, D, dtype=F16, device=dev)
        emb.view(B, Tp, D)[:, :T] = self.tok[ids]                        # gather (data movement)
        pos = torch.zeros(M, D, dtype=F16, device=dev)
        pos.view(B, Tp, D)[:, :T] = self.pos[:T]
        h = torch.empty(M, D, dtype=F16, device=dev)
        ops.copy_add(emb, h, 0, add=pos)                                  # token + position embeddings
        n = torch.empty(M, D, dtype=F16, device=dev)
        qk = torch.empty(M, 2 * D, dtype=F16, device=dev)
        vt = torch.empty(D, M, dtype=F16, device=dev)
        a = torch.empty(M, D, dtype=F16, device=dev)
        f = torch.empty(M, w["0.fc1.w"].shape[0], dtype=F16, device=dev)
        penult = None
        for i in range(self.depth):
            if i == take:
                penult = h.clone()                                        # hidden_states[-2]: input of the last layer
            ops.layernorm(h, n, w[f"{i}.n1.g"], w[f"{i}.n1.b"], self.eps)
            ops.gemm(n, w[f"{i}.qk.w"], qk, bias=w[f"{i}.qk.b"])
            ops.gemm(w[f"{i}.v.w"], n, vt)
            ops.attention(qk[:, :D], a, [(qk[:, D:], Tp, vt, Tp, T)], B, self.heads, Tp, causal=True)
            ops.gemm(a, w[f"{i}.o.w"], h, bias=w[f"{i}.o.b"], res=h)
            ops.layernorm(h, n, w[f"{i}.n2.g"], w[f"{i}.n2.b"], self.eps)
            ops.gemm(n, w[f"{i}.fc1.w"], f, bias=w[f"{i}.fc1.b"], act=self.act)
            ops.gemm(f, w[f"{i}.fc2.w"], h, bias=w[f"{i}.fc2.b"], res=h)
        pooled = None
        if self.proj is not None:
            ops.layernorm(h, n, w["ln.g"], w["ln.b"], self.eps)            # final_layer_norm
            if self.eos_token_id == 2:                                    # legacy configs: eot = highest id in the sequence
                eos = ids.to(torch.int).argmax(dim=-1)
            else:
                eos = (ids == self.eos_token_id).int().argmax(dim=-1)
            rows = n.view(B, Tp, D)[torch.arange(B, device=dev), eos]      # (B, D) gather
            rows8 = torch.zeros((B + 7) // 8 * 8, D, dtype=F16, device=dev)
            rows8[:B] = rows
            out = torch.empty(rows8.shape[0], self.proj.shape[0], dtype=F16, device=dev)
            ops.gemm(rows8, self.proj, out)                               # text_projection (no bias)
            pooled = out[:B]
        return penult.view(B, Tp, D)[:, :T], pooled

    __call__ = forward


def encode_prompt_ids(enc1: HipCLIPText, enc2: HipCLIPText, ids1: torch.Tensor, ids2: torch.Tensor, clip_skip=None):
    """The tensor part of `encode_prompt` (pipelines/sdxl_instantir.py:516-560): concat the two penultimate hidden
    states -> (B, 77, 2048) prompt_embeds; pooled = projected EOS feature of the second encoder -> (B, 1280)."""
    h1, _ = enc1(ids1, clip_skip)
    h2, pooled = enc2(ids2, clip_skip)
    return torch.cat([h1, h2], dim=-1), pooled
