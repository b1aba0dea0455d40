"""`InstantIRPipeline` for MI355X: the reference's pipeline API over the HIP engine.

Mirrors `pipelines/sdxl_instantir.py::InstantIRPipeline` (constructor :303-348, `prepare_previewers`
:350-397, `__call__` :1065-1739) for the denoising path: same keyword arguments and defaults, same
error conventions for the checks it implements, same outputs.  The hot loop (:1497-1660) runs as
HIP kernel launches, optionally replayed from a hipGraph per loop phase.

Observable deviations, all documented in SURVEY.md Appendix C and DESIGN.md:
  Q2  a step whose cond_scale is <= 0.1 everywhere skips previewer and Aggregator and, as the reference's statements do
      (:1602-1603 run unconditionally), re-scales the PREVIOUS step's already scaled residuals (loop phase `unet_res`); the
      adds are skipped only when that compound scale is zero for every image (same result).  On step 0 there are no previous
      residuals: a clear error instead of the reference's NameError.
  callback_on_step_end may return `latents` and `prompt_embeds` (:1651-1659); a replaced context re-hoists the text K / V of the
      cross-attention blocks and re-captures the step.  `negative_prompt_embeds` is accepted and, as in the reference (the name
      is never read after the CFG concat at :1465), has no effect.
  Q3  preview latents are copied to the host only when `save_preview_row=True`.
  Q5  `multistep_restore=True` raises NotImplementedError (unusable with the shipped scheduler).
  Q8/Q13  Resampler / embeddings hoisted out of the step.
  Q15 the reference defaults `ip_adapter_image` to `[image]` BEFORE `check_inputs` (:1278-1279), so passing
      `ip_adapter_image_embeds` always trips its own "provide either ... or ..." check (:850-853); here the default is
      applied only when no embeds are given, so the embeds path (the parity hook) is usable.
Latents are carried in fp32 between steps (the reference carries fp16); UNet inputs are fp16.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional

import torch

from . import ops
from .config import UNetConfig, VAEConfig
from .engine import CPAD, F16, HipAggregator, HipUNet
from .schedulers import DDIMScheduler, DDPMScheduler, LCMSingleStepScheduler  # noqa: F401
from .weights import LCM_LORA_MODULES, PREVIEWER_LORA_MODULES, lora_target


class StableDiffusionXLPipelineOutput(SimpleNamespace):
    """`.images` like diffusers' output class (pipelines/sdxl_instantir.py:1739)."""


class _AggregatorHandle:
    """`pipe.aggregator.load_state_dict(sd)` / `.to(...)` surface of infer.py:142-144."""

    def __init__(self, pipe):
        self._pipe = pipe

    def load_state_dict(self, sd, strict=True):
        from .weights import aggregator_specs
        k = "controlnet_mid_block.0.mlp_shared.0.weight"            # SFT hidden width comes from the file (module/aggregator.py:60)
        if k in sd and sd[k].shape[0] != self._pipe.cfg.sft_hidden:
            import dataclasses
            self._pipe.cfg = dataclasses.replace(self._pipe.cfg, sft_hidden=sd[k].shape[0])
        want = {n for n, _, _ in aggregator_specs(self._pipe.cfg)}
        missing, unexpected = sorted(want - set(sd)), sorted(set(sd) - want)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Aggregator: missing {missing[:5]} unexpected {unexpected[:5]}")
        self._pipe._agg_sd = dict(sd)
        self._pipe._agg = None

    def to(self, *a, **k):
        return self

    def from_unet(self):
        """`Aggregator.from_unet(unet)` + `remove_attn2` (module/aggregator.py:503-578, pipelines/sdxl_instantir.py:165-177,
        320-322) -- what the reference pipeline holds when no aggregator is passed in: `conv_in` (also as `ref_conv_in`),
        time / add embeddings, down blocks and mid block copied from the UNet (cross-attention and its norm dropped), the
        SFT heads freshly initialised behind zero 1x1 convolutions, so every residual it emits is exactly zero.  (The
        reference draws the SFT 3x3 weights from PyTorch's default init; behind the zero convs their values are unobservable,
        zeros are used.)"""
        from .weights import aggregator_specs
        usd = self._pipe._unet_sd
        out = {}
        for name, shape, _ in aggregator_specs(self._pipe.cfg):
            if name.startswith("ref_conv_in."):
                out[name] = usd["conv_in." + name[len("ref_conv_in."):]]
            elif name.startswith("controlnet_"):
                out[name] = torch.zeros(shape, dtype=torch.float32)
            else:
                out[name] = usd[name]
        return out


class _UNetHandle:
    """The slice of `pipe.unet`'s peft surface the reference's callers touch: `set_adapter` / `active_adapters`
    (gradio_demo/app.py:116-120 switches between the `previewer` and `lcm` LoRAs per request), `enable_adapters` /
    `disable_adapters` (pipelines/sdxl_instantir.py:396, :1543, :1556).  Each adapter is a merged second weight copy built on
    first use and kept (5 GB at SDXL size); switching adapters swaps which copy the previewer pass launches."""

    def __init__(self, pipe):
        self._pipe = pipe

    def active_adapters(self):
        return [self._pipe._active_adapter] if self._pipe._active_adapter is not None else []

    def set_adapter(self, name):
        if isinstance(name, (list, tuple)):
            if len(name) != 1:
                raise ValueError("one active LoRA adapter at a time (the reference's callers never mix them)")
            name = name[0]
        if name not in self._pipe._adapters:
            raise ValueError(f"Adapter {name} not found. Available adapters: {list(self._pipe._adapters)}")
        self._pipe._active_adapter = name
        self._pipe._unet_prev = self._pipe._prev_nets.get((name, 1.0))

    def enable_adapters(self):       # the loop itself decides which pass runs with the LoRA (:1543-1556)
        pass

    def disable_adapters(self):
        pass


class InstantIRPipeline:
    vae_scale_factor = 8

    def __init__(self, cfg: UNetConfig, unet_state_dict: Dict[str, torch.Tensor], aggregator_state_dict=None,
                 scheduler=None, vae=None, device="cuda:0", image_encoder=None, text_encoder=None, text_encoder_2=None,
                 tokenizer=None, tokenizer_2=None):
        """unet_state_dict: diffusers SDXL names + TA-IP processor + Resampler weights (what
        `load_adapter_to_pipe` leaves in `pipe.unet`, module/ip_adapter/utils.py:136-161)."""
        self.cfg = cfg
        self.device = torch.device(device)
        self._unet_sd = unet_state_dict
        self._agg_sd = aggregator_state_dict
        self._adapters = {}                          # name -> (peft-named LoRA tensors, alpha / r)
        self._active_adapter = None
        self._prev_nets = {}                         # (adapter, cross_attention_kwargs["scale"]) -> merged previewer copy
        self.unet = _UNetHandle(self)
        self.scheduler = scheduler if scheduler is not None else DDPMScheduler()
        self.aggregator = _AggregatorHandle(self)
        self.vae = vae
        self.image_encoder = image_encoder          # encoders.HipDinov2 (module/ip_adapter/utils.py:106-118)
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2     # encoders.HipCLIPText
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2               # callables: list[str] -> (B,77) int64 ids
        self._unet = self._unet_prev = self._agg = self._unet_prev8 = None
        self._graphs = {}
        self._loop_cache = None                     # (key, _DenoiseLoop) of the last call: see _loop_for
        self._prompt_cache = {}                     # (token ids, encoders) -> (prompt_embeds, pooled): see encode_prompt
        self.use_graphs = True
        self.overlap_streams = True
        self.overlap_sft = os.environ.get("IIR_OVERLAP_SFT", "1") != "0"     # shallow SFT heads beside the decoder's first up block
        self._guidance_scale = 7.0

    # ---- reference surface ----------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, device="cuda:0", **kwargs):
        """`InstantIRPipeline.from_pretrained(sdxl_dir, torch_dtype=torch.float16)` (infer.py:117-120): reads the SDXL
        directory layout with this build's own readers -- `unet/`, `vae/`, `text_encoder(_2)/`
        safetensors (+ config.json), `tokenizer(_2)/`.  Local directories only: there is no hub access.  The TA-IP
        adapter, previewer LoRA and Aggregator are attached afterwards exactly as infer.py does
        (`loaders.load_adapter_to_pipe`, `prepare_previewers`, `aggregator.load_state_dict`).  Compute is fp16 with
        fp32 accumulation whatever `torch_dtype` says."""
        from . import loaders
        from .encoders import HipCLIPText
        from .vae import HipVAE
        d = pretrained_model_name_or_path
        if not os.path.isdir(d):
            raise FileNotFoundError(f"{d!r} is not a local directory (hub ids cannot be fetched: no network)")
        cfg, vc = loaders.unet_config_from_dir(d), loaders.vae_config_from_dir(d)
        unet_sd = loaders.load_component(d, "unet")
        vae = HipVAE(vc, loaders.load_component(d, "vae"), device) if os.path.isdir(os.path.join(d, "vae")) else None
        enc, tok = [], []
        for sub, tsub, default_act in (("text_encoder", "tokenizer", "quick_gelu"), ("text_encoder_2", "tokenizer_2", "gelu")):
            e = t = None
            if os.path.isdir(os.path.join(d, sub)):
                cj = os.path.join(d, sub, "config.json")
                c = loaders._read_json(cj) if os.path.isfile(cj) else {}
                e = HipCLIPText(loaders.load_component(d, sub), device, hidden_act=c.get("hidden_act", default_act),
                                eos_token_id=c.get("eos_token_id", 2), eps=c.get("layer_norm_eps", 1e-5))
            if os.path.isdir(os.path.join(d, tsub)):
                from transformers import CLIPTokenizer
                tk = CLIPTokenizer.from_pretrained(os.path.join(d, tsub))
                t = (lambda tk_: (lambda texts: tk_(texts, padding="max_length", max_length=tk_.model_max_length, truncation=True,
                                                    return_tensors="pt").input_ids))(tk)
            enc.append(e)
            tok.append(t)
        return cls(cfg, unet_sd, scheduler=kwargs.get("scheduler"), vae=vae, device=device, text_encoder=enc[0],
                   text_encoder_2=enc[1], tokenizer=tok[0], tokenizer_2=tok[1])

    @classmethod
    def from_modules(cls, vae=None, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, unet=None, scheduler=None,
                     aggregator=None, force_zeros_for_empty_prompt=True, add_watermarker=None, feature_extractor=None,
                     image_encoder=None, device="cuda:0", unet_config=None, vae_config=None):
        """The reference's own constructor signature (pipelines/sdxl_instantir.py:303-322): build the pipeline from module OBJECTS
        somebody else loaded -- anything with `.state_dict()` (diffusers / transformers modules, or plain dicts of their
        tensors) and, where geometry is not implied by the tensors, a `.config` (mapping or attribute object; `unet_config=` /
        `vae_config=` override it).  `aggregator=None` = `Aggregator.from_unet(unet)` (:319-320); tokenizers may be transformers
        tokenizers or callables `list[str] -> (B, 77) ids`; `image_encoder` a `HipDinov2` / `HipCLIPVision` or a module of one
        of those architectures (DINOv2 unless its tensors say CLIP).  `add_watermarker` / `feature_extractor` are accepted and
        unused (no watermarking; image pre-processing is `encoders.preprocess_*`), `force_zeros_for_empty_prompt` must stay True
        (the only value the reference's loaders produce)."""
        from . import loaders
        from .encoders import HipCLIPText, HipCLIPVision, HipDinov2
        from .vae import HipVAE
        if unet is None:
            raise ValueError("from_modules: `unet` is required")
        if not force_zeros_for_empty_prompt:
            raise NotImplementedError("force_zeros_for_empty_prompt=False is not supported")

        def tensors(m):
            return dict(m) if isinstance(m, dict) else dict(m.state_dict())

        def conf(m):
            c = getattr(m, "config", None)
            if c is None:
                return None
            return dict(c) if hasattr(c, "keys") else {k: getattr(c, k) for k in dir(c) if not k.startswith("_") and not callable(getattr(c, k))}

        cfg = unet_config or (loaders.unet_config_from_dict(conf(unet)) if conf(unet) else UNetConfig.sdxl())
        hv = vae
        if vae is not None and not isinstance(vae, HipVAE):
            vc = vae_config or (loaders.vae_config_from_dict(conf(vae)) if conf(vae) else VAEConfig.sdxl())
            hv = HipVAE(vc, tensors(vae), device)
        encs = []
        for m, default_act in ((text_encoder, "quick_gelu"), (text_encoder_2, "gelu")):
            if m is None or isinstance(m, HipCLIPText):
                encs.append(m)
                continue
            c = conf(m) or {}
            encs.append(HipCLIPText(tensors(m), device, hidden_act=c.get("hidden_act", default_act), eos_token_id=c.get("eos_token_id", 2),
                                    eps=c.get("layer_norm_eps", 1e-5)))
        toks = []
        for t in (tokenizer, tokenizer_2):
            if t is not None and hasattr(t, "model_max_length"):          # a transformers tokenizer
                t = (lambda tk_: (lambda texts: tk_(texts, padding="max_length", max_length=tk_.model_max_length, truncation=True,
                                                    return_tensors="pt").input_ids))(t)
            toks.append(t)
        ie = image_encoder
        if ie is not None and not isinstance(ie, (HipDinov2, HipCLIPVision)):
            sd_ie = tensors(ie)
            ie = HipCLIPVision(sd_ie, device) if any("vision_model" in k for k in sd_ie) else HipDinov2(sd_ie, device)
        pipe = cls(cfg, tensors(unet), scheduler=scheduler, vae=hv, device=device, image_encoder=ie, text_encoder=encs[0],
                   text_encoder_2=encs[1], tokenizer=toks[0], tokenizer_2=toks[1])
        if aggregator is not None:
            pipe.aggregator.load_state_dict(tensors(aggregator))
        return pipe

    def to(self, *a, **k):
        return self

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def _lora(self):
        return self._adapters[self._active_adapter][0] if self._active_adapter is not None else None

    @_lora.setter
    def _lora(self, v):            # bench.py drops the host copies once the device copies exist
        if v is None:
            self._adapters = {k: (None, sc) for k, (_, sc) in self._adapters.items()}

    @property
    def _lora_scaling(self):
        return self._adapters[self._active_adapter][1] if self._active_adapter is not None else 1.0

    @property
    def do_classifier_free_guidance(self):
        # pipelines/sdxl_instantir.py:1050-1051 (time_cond_proj_dim is None for SDXL)
        return self._guidance_scale > 1

    def prepare_previewers(self, lora_state_dict: Dict[str, torch.Tensor], use_lcm=False, lora_alpha=None):
        """pipelines/sdxl_instantir.py:350-397.  `lora_state_dict` holds peft-named tensors
        (`<module>.lora_A.weight` / `.lora_B.weight`) -- the form the reference has after
        `convert_unet_state_dict_to_peft` and the `attn2` -> `attn2.processor` rename (:364-370).
        Raises ValueError on keys that match no LoRA target (:390-394); missing keys are ignored."""
        targets = LCM_LORA_MODULES if use_lcm else PREVIEWER_LORA_MODULES
        if isinstance(lora_state_dict, (str, os.PathLike)):          # the reference's own call form: a directory / file path
            from .loaders import read_previewer_lora
            lora_state_dict, file_alpha = read_previewer_lora(os.fspath(lora_state_dict))
            lora_alpha = file_alpha if lora_alpha is None else lora_alpha
        ranks = {v.shape[0] for k, v in lora_state_dict.items() if k.endswith(".lora_A.weight")}
        if len(ranks) == 1 and next(iter(ranks)) != self.cfg.lora_rank:
            import dataclasses
            self.cfg = dataclasses.replace(self.cfg, lora_rank=next(iter(ranks)))
        unexpected = []
        for k in lora_state_dict:
            for suf in (".lora_A.weight", ".lora_B.weight"):
                if k.endswith(suf):
                    path = k[: -len(suf)]
                    if not lora_target(path, targets) or (path + ".weight") not in self._unet_sd:
                        unexpected.append(k)
                    break
            else:
                unexpected.append(k)
        if unexpected:
            raise ValueError("Loading adapter weights from state_dict led to unexpected keys not found in the model: "
                             f" {unexpected}. ")
        lora_alpha = 1 if lora_alpha is None else lora_alpha
        name = "lcm" if use_lcm else "previewer"                       # :383
        self._adapters[name] = (dict(lora_state_dict), lora_alpha / self.cfg.lora_rank)   # peft: alpha / r with r = 64 (:376-381)
        for k in [k for k in self._prev_nets if k[0] == name]:
            del self._prev_nets[k]
        self._unet_prev8 = None
        self.unet.set_adapter(name)        # diffusers' `add_adapter` ends in `set_adapter(adapter_name)`: the newest one is active
        return lora_alpha

    # ---- engine construction ----------------------------------------------------------------------
    def _build(self, lora_mult: float = 1.0):
        """`lora_mult`: `cross_attention_kwargs["scale"]`, which diffusers' UNet forward turns into `scale_lora_layers(unet, scale)`
        for the pass that runs with the adapters on -- i.e. the previewer copy is W + scale * (alpha / r) * B A."""
        if self._unet is None:
            self._unet = HipUNet(self.cfg, self._unet_sd, self.device)
        if self._active_adapter is not None:
            key = (self._active_adapter, float(lora_mult))
            if key not in self._prev_nets and float(lora_mult) != 1.0:
                # a merged copy is a whole UNet (~5 GB at SDXL size): keep ONE non-default scale per adapter, evict the previous
                for k_old in [k for k in self._prev_nets if k[0] == self._active_adapter and k[1] != 1.0 and len(k) == 2]:
                    del self._prev_nets[k_old]
            if key not in self._prev_nets:
                if self._lora is None:
                    raise RuntimeError(f"LoRA adapter {self._active_adapter!r}: the host copy was released; call prepare_previewers again")
                self._prev_nets[key] = HipUNet(self.cfg, self._unet_sd, self.device, lora=self._lora,
                                               lora_scaling=self._lora_scaling * float(lora_mult))
            self._unet_prev = self._prev_nets[key]
        if self._agg is None:
            if self._agg_sd is None:
                self._agg_sd = self.aggregator.from_unet()
            self._agg = HipAggregator(self.cfg, self._agg_sd, self.device)

    def _loop_for(self, B, rep, Hl, Wl, st, st_prev, st_agg, lq, reference_latents, previewer_scheduler, guidance_rescale):
        """The step's buffers and captured hipGraphs are kept from one call to the next: a second image of the same geometry,
        through the same engines, re-uses them (its hoisted K / V, embeddings and LQ latent are copied into the captured
        tensors) instead of paying the warm-up step, the capture and the graph instantiation again (~0.1 s of a 1.9 s call
        at 1024^2).  Anything the captured launches depend on is part of the key; `IIR_LOOP_CACHE=0` switches it off."""
        nets = (self._unet, self._unet_prev, self._agg)
        key = (B, rep, Hl, Wl, reference_latents is not None, float(guidance_rescale or 0.0), self.use_graphs, self.overlap_streams,
               self.overlap_sft, tuple(None if n is None else (id(n), n.arena_gen, n.inkernel_prefetch, n.gn_fuse) for n in nets))
        cached = self._loop_cache
        if cached is not None and cached[0] == key and os.environ.get("IIR_LOOP_CACHE", "1") != "0":
            if cached[1].adopt(st, st_prev, st_agg, lq, reference_latents, previewer_scheduler):
                return cached[1]
        self._loop_cache = None                      # drop the old graphs before building the new ones
        loop = _DenoiseLoop(self, B, rep, Hl, Wl, st, st_prev, st_agg, lq, reference_latents, previewer_scheduler,
                            guidance_rescale=guidance_rescale)
        # the entry keeps the engines alive: `id()` in the key can then not be re-issued to a NEW engine (adapter switch, LoRA
        # scale change) while graphs captured on the old one's arena and weights are still cached
        self._loop_cache = (key, loop, nets)
        return loop

    # ---- input checks (pipelines/sdxl_instantir.py:749-864, the conditions that apply to tensor inputs) ----
    def check_inputs(self, prompt, prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds,
                     ip_adapter_image, ip_adapter_image_embeds, control_guidance_start, control_guidance_end,
                     callback_on_step_end_tensor_inputs):
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in ("latents", "prompt_embeds", "negative_prompt_embeds") for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in ['latents', 'prompt_embeds', "
                             f"'negative_prompt_embeds'], but found {callback_on_step_end_tensor_inputs}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                             " only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and (self.text_encoder is None or self.text_encoder_2 is None or self.tokenizer is None):
            raise NotImplementedError("`prompt` strings need text encoders + tokenizers attached to the pipeline "
                                      "(tokenizer vocabularies are not available offline): pass `prompt_embeds` / "
                                      "`pooled_prompt_embeds`, or `prompt_ids` / `prompt_ids_2`")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                             f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")
        if prompt is None and prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed. Make sure to "
                             "generate `pooled_prompt_embeds` from the same text encoder that was used to generate `prompt_embeds`.")
        if negative_prompt_embeds is not None and negative_pooled_prompt_embeds is None:
            raise ValueError("If `negative_prompt_embeds` are provided, `negative_pooled_prompt_embeds` also have to be passed.")
        for s, e in [(control_guidance_start, control_guidance_end)]:
            if s >= e:
                raise ValueError(f"control guidance start: {s} cannot be larger or equal to control guidance end: {e}.")
            if s < 0.0:
                raise ValueError(f"control guidance start: {s} can't be smaller than 0.")
            if e > 1.0:
                raise ValueError(f"control guidance end: {e} can't be larger than 1.0.")
        if ip_adapter_image is not None and ip_adapter_image_embeds is not None:
            raise ValueError("Provide either `ip_adapter_image` or `ip_adapter_image_embeds`. Cannot leave both "
                             "`ip_adapter_image` and `ip_adapter_image_embeds` defined.")
        if ip_adapter_image_embeds is not None:
            if not isinstance(ip_adapter_image_embeds, list):
                raise ValueError(f"`ip_adapter_image_embeds` has to be of type `list` but is {type(ip_adapter_image_embeds)}")
            if ip_adapter_image_embeds[0].ndim not in [3, 4]:
                raise ValueError("`ip_adapter_image_embeds` has to be a list of 3D or 4D tensors but is "
                                 f"{ip_adapter_image_embeds[0].ndim}D")

    def encode_prompt(self, prompt=None, prompt_2=None, negative_prompt=None, negative_prompt_2=None, prompt_ids=None,
                      prompt_ids_2=None, negative_prompt_ids=None, negative_prompt_ids_2=None, do_cfg=True, clip_skip=None):
        """`encode_prompt` (pipelines/sdxl_instantir.py:400-632) on the HIP text encoders: both CLIP encoders'
        hidden_states[-2] concatenated (B,77,2048) + the second encoder's projected EOS feature (B,1280).  Strings
        go through the attached tokenizers (77 tokens, padded / truncated); ids may be passed directly.  With no
        negative prompt the negative embeddings are zeros (`force_zeros_for_empty_prompt`, :567-571)."""
        from .encoders import encode_prompt_ids

        def ids_of(texts, tok):
            texts = [texts] if isinstance(texts, str) else list(texts)
            return tok(texts)

        if prompt_ids is None:
            prompt_ids = ids_of(prompt, self.tokenizer)
            prompt_ids_2 = ids_of(prompt_2 if prompt_2 is not None else prompt, self.tokenizer_2 or self.tokenizer)
        if prompt_ids_2 is None:
            prompt_ids_2 = prompt_ids
        def encode(ids, ids2, skip):
            # A batch job restores every image under the SAME prompt (infer.py:211-222): the two CLIP passes (~20 ms per prompt at
            # SDXL size) are kept per token-id pair -- the encoders are deterministic, the cached tensors are never written to.
            key = (ids.cpu().numpy().tobytes(), ids2.cpu().numpy().tobytes(), tuple(ids.shape), skip, id(self.text_encoder), id(self.text_encoder_2))
            hit = self._prompt_cache.get(key)
            if hit is None or hit[2] is not self.text_encoder or hit[3] is not self.text_encoder_2:
                hit = (*encode_prompt_ids(self.text_encoder, self.text_encoder_2, ids, ids2, skip), self.text_encoder, self.text_encoder_2)
                if len(self._prompt_cache) >= 8:
                    self._prompt_cache.pop(next(iter(self._prompt_cache)))
                self._prompt_cache[key] = hit
            return hit[0], hit[1]

        pe, pooled = encode(prompt_ids, prompt_ids_2, clip_skip)
        npe = npooled = None
        if do_cfg:
            if negative_prompt_ids is None and negative_prompt is not None:
                negative_prompt_ids = ids_of(negative_prompt, self.tokenizer)
                negative_prompt_ids_2 = ids_of(negative_prompt_2 if negative_prompt_2 is not None else negative_prompt,
                                               self.tokenizer_2 or self.tokenizer)
            if negative_prompt_ids is None:
                npe, npooled = torch.zeros_like(pe), torch.zeros_like(pooled)
            else:
                # (the reference always takes hidden_states[-2] for the negative prompt, :586: clip_skip is not applied)
                npe, npooled = encode(negative_prompt_ids, negative_prompt_ids_2 if negative_prompt_ids_2 is not None else negative_prompt_ids, None)
        return pe, npe, pooled, npooled

    def encode_image(self, image):
        """`encode_image` (pipelines/sdxl_instantir.py:636-670): (features, zero-image features).  DINO branch (:660-667):
        `last_hidden_state`; CLIP branch as `prepare_ip_adapter_image_embeds` calls it for a Resampler projector (:696-699,
        :644-654): `hidden_states[-2]` of the image and of `zeros_like(image)`.
        `image`: PIL image(s) (pre-processed like the encoder's own image processor) or a normalised tensor."""
        from .encoders import HipCLIPVision, clip_preprocess, dinov2_preprocess
        if self.image_encoder is None:
            raise NotImplementedError("pass `ip_adapter_image_embeds` or attach an image encoder (encoders.HipDinov2)")
        pre = clip_preprocess if isinstance(self.image_encoder, HipCLIPVision) else dinov2_preprocess
        px = image if torch.is_tensor(image) else pre(image)
        return self.image_encoder.encode_image_pair(px)

    def prepare_ip_adapter_image_embeds(self, ip_adapter_image, do_cfg):
        """:672-707 for one IP adapter: [(2, B, S, E)] = cat([zero-image features, image features]) under CFG."""
        f, z = self.encode_image(ip_adapter_image)
        f, z = f.unsqueeze(0), z.unsqueeze(0)
        return [torch.cat([z, f]) if do_cfg else f]

    @staticmethod
    def _prepare_image(image):
        """`prepare_image` (:905-929) via VaeImageProcessor(do_normalize=True): PIL / numpy / [0,1] tensors
        become fp32 NCHW in [-1,1]; 4-channel tensors (latents) and tensors already below 0 pass through."""
        if torch.is_tensor(image):
            if image.shape[1] == 4 or image.min() < 0:
                return image
            return image * 2.0 - 1.0
        import numpy as np
        if not isinstance(image, (list, tuple)):
            image = [image]
        arrs = [np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.float32) / 255.0 for im in image]
        x = torch.from_numpy(np.stack(arrs)).permute(0, 3, 1, 2)
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"image size {tuple(x.shape[2:])} must be a multiple of 8 (infer.py:31-66 resizes to multiples of 64)")
        return x * 2.0 - 1.0

    # ---- single-step previewer restoration (BASELINE configs[4]; spec train_previewer_lora.py:118-145) ------------
    @torch.no_grad()
    def restore_single_step(self, image, prompt_embeds, pooled_prompt_embeds, ip_adapter_image_embeds=None,
                            ip_adapter_image=None, timestep: int = 999, previewer_scheduler=None, generator=None,
                            init_noise=None, output_type: str = "pil", fp8: bool = False, **kwargs):
        """LCM one-step restoration with the previewer LoRA, no CFG (guidance 1.0): noise the LQ latent to `timestep`
        (`prepare_latents` there = scheduler.add_noise), ONE UNet pass with the LoRA enabled, `LCMSingleStepScheduler.step`
        (schedulers/lcm_single_step_scheduler.py:421-489), VAE decode.  `fp8=True` (BASELINE configs[4]): the transformer
        blocks' linear layers of that pass run on fp8-E4M3 weights (per-output-channel scales) AND fp8 activations, stored as
        such by the LayerNorm / attention / GEGLU launch that produces them (`ops.gemm_fp8`, `v_mfma_f32_16x16x32_fp8_fp8`; round 3) --
        a third weight set, built on first use; its tolerance is its own (45.8 dB against the fp32 oracle at SDXL shapes)."""
        from .engine import CPAD, F16
        if self._lora is None:
            raise RuntimeError("restore_single_step needs the previewer LoRA: call prepare_previewers(...)")
        if fp8:
            key8 = (self._active_adapter, 1.0, "fp8")            # one fp8 copy per adapter: `set_adapter` must not leave a stale one
            if key8 not in self._prev_nets:
                self._prev_nets[key8] = HipUNet(self.cfg, self._unet_sd, self.device, lora=self._lora, lora_scaling=self._lora_scaling,
                                                fp8_linear=True)
            net = self._unet_prev8 = self._prev_nets[key8]
        else:
            self._build()
            net = self._unet_prev
        sched = previewer_scheduler if previewer_scheduler is not None else LCMSingleStepScheduler.from_config(self.scheduler.config)
        dev, cfg = self.device, self.cfg
        image = self._prepare_image(image)
        if image.shape[1] != 4:
            if self.vae is None:
                raise NotImplementedError("pixel-space `image` needs a VAE; pass the LQ latent (B,4,h,w)")
            image = self.vae.encode_to_latent(image, eps=kwargs.get("vae_noise"))
        lq = image.to(dev, torch.float32).contiguous()
        B, _, Hl, Wl = lq.shape
        if ip_adapter_image_embeds is None:
            ip_adapter_image_embeds = self.prepare_ip_adapter_image_embeds(ip_adapter_image, False)
        px = (Hl * self.vae_scale_factor, Wl * self.vae_scale_factor)
        time_ids = torch.tensor([[px[0], px[1], 0, 0, px[0], px[1]]], dtype=torch.float32).repeat(B, 1)
        st = net.prepare(prompt_embeds, pooled_prompt_embeds, time_ids, net.resampler(ip_adapter_image_embeds[0]), Hl, Wl)
        if init_noise is None:
            gdev = generator.device if generator is not None else dev
            init_noise = torch.randn(lq.shape, generator=generator, device=gdev, dtype=torch.float32)
        x = sched.add_noise(lq, init_noise.to(dev, torch.float32), torch.tensor([timestep] * B)).contiguous()
        lat16 = torch.zeros(B * Hl * Wl, CPAD, dtype=F16, device=dev)
        ops.pack_latent(x, lat16)
        t_dev = torch.full((B, 1), float(timestep), dtype=torch.float32, device=dev)
        eps = net.forward(lat16, t_dev, st)
        out16 = torch.zeros(B * Hl * Wl, CPAD, dtype=F16, device=dev)
        out = torch.empty(B, 4, Hl, Wl, dtype=torch.float32, device=dev)
        coef = torch.tensor(sched.preview_coefficients(timestep), dtype=torch.float32).to(dev)
        ops.lcm_step(eps, B, 1, coef, x, out16, out)
        if output_type == "latent":
            return StableDiffusionXLPipelineOutput(images=out)
        if self.vae is None:
            raise NotImplementedError("output_type other than 'latent' needs a VAE attached to the pipeline")
        return StableDiffusionXLPipelineOutput(images=self.vae.decode_latent(out, output_type))

    # ---- the call ---------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, image=None, height=None, width=None, num_inference_steps: int = 30,
                 timesteps: List[int] = None, denoising_end: Optional[float] = None, guidance_scale: float = 7.0,
                 negative_prompt=None, negative_prompt_2=None, num_images_per_prompt: Optional[int] = 1, eta: float = 0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, ip_adapter_image=None, ip_adapter_image_embeds=None,
                 output_type: Optional[str] = "pil", return_dict: bool = True, save_preview_row: bool = False,
                 init_latents_with_lq: bool = True, multistep_restore: bool = False, adastep_restore: bool = False,
                 cross_attention_kwargs=None, guidance_rescale: float = 0.0, controlnet_conditioning_scale=1.0,
                 control_guidance_start: float = 0.0, control_guidance_end: float = 1.0, preview_start: float = 0.0,
                 preview_end: float = 1.0, original_size=None, crops_coords_top_left=(0, 0), target_size=None,
                 negative_original_size=None, negative_crops_coords_top_left=(0, 0), negative_target_size=None,
                 clip_skip=None, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], previewer_scheduler=None,
                 reference_latents=None, init_noise=None, step_noises=None, **kwargs):
        """Keyword arguments and defaults of pipelines/sdxl_instantir.py:1067-1115.  Two additions for
        bit-reproducible parity runs: `init_noise` (the randn of init_latents) and
        `step_noises` (list of per-step DDPM noises) replace draws from `generator` when given.
        `image` must be the LQ *latent* (B,4,h,w) here unless a VAE is attached (`image.shape[1] == 4` branch of :1369-1382)."""
        # :1531-1535 merges the caller's dict over {"temb": emb} and hands it to both UNet passes.  What a key can do there:
        # "scale" is popped by diffusers' UNet forward and scales every LoRA layer for that pass (it also sets the text-encoder
        # LoRA scale, :1324-1326 -- no text-encoder LoRA exists on this path); "temb" / "external_kv" are the processors' own
        # arguments (module/ip_adapter/attention_processor.py:1093-1100): the loop owns `temb`, and no caller passes `external_kv`;
        # any other key is a TypeError inside the reference's processors.  "scale" is honoured by building the previewer's
        # merged copy with scale * alpha / r; the rest is refused with the reason.
        lora_mult = 1.0
        if cross_attention_kwargs:
            extra = set(cross_attention_kwargs) - {"scale"}
            if extra:
                raise ValueError(f"cross_attention_kwargs keys {sorted(extra)} are not accepted: the TA-IP attention processors take "
                                 "`temb` (set by the loop) and nothing a caller may override; only 'scale' (LoRA scale) is honoured")
            lora_mult = float(cross_attention_kwargs["scale"])
        if multistep_restore:
            raise NotImplementedError("multistep_restore passes kwargs the shipped DDPM scheduler does not accept "
                                      "(SURVEY.md Appendix C Q5)")
        if prompt is None and prompt_embeds is None and kwargs.get("prompt_ids") is not None and self.text_encoder is not None:
            prompt_embeds_chk = kwargs["prompt_ids"]          # ids stand in for the prompt in the exclusivity checks
        else:
            prompt_embeds_chk = prompt_embeds
        self.check_inputs(prompt, prompt_embeds_chk, negative_prompt_embeds,
                          pooled_prompt_embeds if prompt_embeds_chk is prompt_embeds else torch.zeros(1), negative_pooled_prompt_embeds,
                          ip_adapter_image, ip_adapter_image_embeds, control_guidance_start, control_guidance_end,
                          callback_on_step_end_tensor_inputs)
        self._guidance_scale = guidance_scale
        cfg, dev = self.cfg, self.device
        do_cfg = self.do_classifier_free_guidance
        self._build(lora_mult)

        if prompt_embeds is None:                                                   # :1325-1348
            prompt_embeds, ne, pooled_prompt_embeds, npool = self.encode_prompt(
                prompt, prompt_2, negative_prompt, negative_prompt_2, kwargs.get("prompt_ids"), kwargs.get("prompt_ids_2"),
                kwargs.get("negative_prompt_ids"), kwargs.get("negative_prompt_ids_2"), do_cfg, clip_skip)
            if negative_prompt_embeds is None:
                negative_prompt_embeds, negative_pooled_prompt_embeds = ne, npool
        nipp = int(num_images_per_prompt or 1)
        pb = prompt_embeds.shape[0]                                                # `batch_size` of :1304-1315
        n_img = 1 if hasattr(image, "size") and not torch.is_tensor(image) and not isinstance(image, (list, tuple)) else len(image)
        assert pb == n_img or n_img == 1                                          # :1310-1315
        if ip_adapter_image is None and ip_adapter_image_embeds is None:           # :1278-1279 (see Q15 in the header)
            if torch.is_tensor(image) and image.dim() == 4 and image.shape[1] == 4:
                raise ValueError("`image` is an LQ latent (B,4,h,w): it cannot stand in for `ip_adapter_image` (the image encoder takes "
                                 "pixels); pass `ip_adapter_image` or `ip_adapter_image_embeds`")
            ip_adapter_image = image
        if nipp > 1:            # diffusers encode_prompt: embeds.repeat(1, n, 1).view(bs * n, ...) == repeat_interleave
            prompt_embeds = prompt_embeds.repeat_interleave(nipp, 0)
            pooled_prompt_embeds = pooled_prompt_embeds.repeat_interleave(nipp, 0)
            if negative_prompt_embeds is not None:
                negative_prompt_embeds = negative_prompt_embeds.repeat_interleave(nipp, 0)
                negative_pooled_prompt_embeds = negative_pooled_prompt_embeds.repeat_interleave(nipp, 0)
        B = pb * nipp
        image = self._prepare_image(image)
        if image.shape[1] != 4:                                                        # :1369-1379
            if self.vae is None:
                raise NotImplementedError("pixel-space `image` needs a VAE; pass the LQ latent (B,4,h,w)")
            image = self.vae.encode_to_latent(image, eps=kwargs.get("vae_noise"), generator=None)
        lq = image.to(dev, torch.float32)
        # prepare_image :919-925: one image serves the whole batch, otherwise each image is repeated per prompt copy
        lq = lq.repeat(B, 1, 1, 1) if lq.shape[0] == 1 else lq.repeat_interleave(nipp, 0)
        lq = lq.contiguous()
        Hl, Wl = lq.shape[2], lq.shape[3]
        height, width = Hl * self.vae_scale_factor, Wl * self.vae_scale_factor

        # -- embeddings, CFG order [negative; positive] (:1456-1464)
        if do_cfg:
            if negative_prompt_embeds is None:
                negative_prompt_embeds = torch.zeros_like(prompt_embeds)           # force_zeros_for_empty_prompt
                negative_pooled_prompt_embeds = torch.zeros_like(pooled_prompt_embeds)
            ctx = torch.cat([negative_prompt_embeds, prompt_embeds], 0)
            pooled = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], 0)
        else:
            ctx, pooled = prompt_embeds, pooled_prompt_embeds
        rep = 2 if do_cfg else 1
        R = rep * B
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        ids = list(original_size) + list(crops_coords_top_left) + list(target_size)     # :965-981
        if cfg.addition_time_embed_dim * len(ids) + cfg.pooled_dim != cfg.add_embed_in:
            raise ValueError("Model expects an added time embedding vector of length "
                             f"{cfg.add_embed_in}, but a vector of {cfg.addition_time_embed_dim * len(ids) + cfg.pooled_dim} was created.")
        neg_ids = ids
        if negative_original_size is not None and negative_target_size is not None:   # :1445-1454
            neg_ids = list(negative_original_size) + list(negative_crops_coords_top_left) + list(negative_target_size)
        if do_cfg:
            # :1459,:1464 verbatim: cat([neg, pos]) then .repeat(B, 1) -- for B > 1 the rows alternate neg, pos, neg, ... while
            # the prompt rows are [neg x B; pos x B]; only observable when negative sizes differ from the positive ones
            time_ids = torch.tensor([neg_ids, ids], dtype=torch.float32).repeat(B, 1)
        else:
            time_ids = torch.tensor([ids], dtype=torch.float32).repeat(R, 1)
        if ip_adapter_image_embeds is None:                                         # :1350-1357, :672-707
            f, z = self.encode_image(ip_adapter_image)                             # (Bimg, S, E) features, zero-image features
            reps = max(B // f.shape[0], 1)                                         # torch.stack([e] * (B // e.shape[0]), dim=0)
            f, z = torch.stack([f] * reps, 0), torch.stack([z] * reps, 0)
            img = torch.cat([z, f]) if do_cfg else f
        else:
            img = ip_adapter_image_embeds[0]
            if do_cfg:                                                              # :709-722
                neg, pos = img.chunk(2)
                img = torch.cat([neg.repeat(nipp, *([1] * (neg.dim() - 1))), pos.repeat(nipp, *([1] * (pos.dim() - 1)))])
            else:
                img = img.repeat(nipp, *([1] * (img.dim() - 1)))
        if img.dim() == 3:
            img = img.unsqueeze(0)
        n_rows = img.shape[0] * img.shape[1]
        if n_rows != R:
            raise ValueError(f"image embeds give {n_rows} rows, the batch has {R} (prompts {pb} x images-per-prompt {nipp}"
                             f"{' x 2 (CFG)' if do_cfg else ''})")

        # -- timetable and gates (:1385, :1415-1425)
        if timesteps is not None:                                                   # retrieve_timesteps, :195-237
            self.scheduler.set_timesteps(timesteps=list(timesteps), device=None)
        else:
            self.scheduler.set_timesteps(num_inference_steps, device=None)
        ts = [int(t) for t in self.scheduler.timesteps]
        n = len(ts)
        keep, previewing = [], []
        for i in range(n):
            keep.append(1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end))
            previewing.append(1.0 - float(i / n < preview_start or (i + 1) / n > preview_end))
        if isinstance(controlnet_conditioning_scale, list):
            assert len(controlnet_conditioning_scale) == n, \
                f"{len(controlnet_conditioning_scale)} controlnet scales do not match number of sampling steps {n}"
            ccs = controlnet_conditioning_scale
        else:
            ccs = [controlnet_conditioning_scale] * n
        if denoising_end is not None and isinstance(denoising_end, float) and 0 < denoising_end < 1:      # :1470-1483
            cutoff = int(round(self.scheduler.config.num_train_timesteps - denoising_end * self.scheduler.config.num_train_timesteps))
            ts = [t for t in ts if t >= cutoff]

        # -- step-invariant device state
        st = self._unet.prepare(ctx, pooled, time_ids, self._unet.resampler(img), Hl, Wl)
        st_prev = None
        if self._unet_prev is not None:
            st_prev = self._unet_prev.prepare(ctx, pooled, time_ids, self._unet_prev.resampler(img), Hl, Wl)
        st_agg = self._agg.prepare(pooled, time_ids, Hl, Wl)

        # -- initial latents (:1388-1403)
        if init_latents_with_lq:
            if init_noise is None:
                gdev = generator.device if generator is not None else dev
                init_noise = torch.randn(lq.shape, generator=generator, device=gdev, dtype=torch.float32)
            x = self.scheduler.add_noise(lq, init_noise.to(dev, torch.float32), torch.tensor([ts[0]] * B))
        else:
            if latents is None:
                gdev = generator.device if generator is not None else dev
                latents = torch.randn(lq.shape, generator=generator, device=gdev, dtype=torch.float32)
            x = latents.to(dev, torch.float32) * self.scheduler.init_noise_sigma
        x = x.contiguous()

        loop = self._loop_for(B, rep, Hl, Wl, st, st_prev, st_agg, lq, reference_latents, previewer_scheduler, guidance_rescale)
        preview_row = []
        preview_factor = torch.ones(B)
        compound = None            # per-image scale the Aggregator's (persistent, raw) outputs currently carry
        pv = None                  # positive half of `preview_latent`: what conditioned the Aggregator last (:1545-1582)
        if adastep_restore and not do_cfg:
            raise ValueError("adastep_restore slices preview_latent[B:], which is empty without classifier-free guidance "
                             "(pipelines/sdxl_instantir.py:1638; SURVEY.md Appendix C Q6)")
        for i, t in enumerate(ts):
            scale_rows = torch.clamp(preview_factor, 0.0, ccs[i]) * keep[i]            # :1538-1540
            use_agg = bool((scale_rows > 0.1).sum().item() > 0)                      # :1542
            if use_agg:
                mode = "preview" if (previewing[i] > 0 and st_prev is not None and previewer_scheduler is not None) else "agg"
                if previewing[i] > 0 and (st_prev is None or previewer_scheduler is None):
                    raise RuntimeError("previewing requested but no previewer: call prepare_previewers(...) and pass "
                                       "previewer_scheduler=LCMSingleStepScheduler")
                compound = scale_rows.clone()
            elif compound is None:
                raise RuntimeError("control_guidance_start > 0 leaves no aggregator residuals for step 0 "
                                   "(the reference fails with NameError here, SURVEY.md Appendix C Q2)")
            else:
                # :1602-1603 run unconditionally: the PREVIOUS step's already scaled residuals are scaled again (Q2).
                # Zero everywhere (the creative phase, keep = 0) -> the adds are skipped, which is the same result.
                compound = compound * scale_rows
                mode = "unet_res" if bool((compound != 0).any()) else "unet"
            noise = None
            if step_noises is not None:
                noise = step_noises[i]
            x0 = loop.step(mode, t, x, (compound if mode != "unet" else scale_rows).repeat(rep), guidance_scale, eta, noise,
                           generator, want_x0=adastep_restore, want_preview=save_preview_row or adastep_restore)
            if mode == "preview":
                pv = loop.preview_f32[B * (rep - 1):]
                if save_preview_row:
                    preview_row.append(pv.clone().cpu())
            elif mode == "agg":
                pv = reference_latents.to(dev, torch.float32) if reference_latents is not None else lq     # :1579-1582
            if adastep_restore:                                                    # :1636-1644
                pv = pv.float().clone()
                pred_x0_l2 = (pv - x0).pow(2).sum(dim=(1, 2, 3))
                prev_l2 = (pv - loop.previewer_mean).pow(2).sum(dim=(1, 2, 3))
                loop.previewer_mean = pv
                preview_factor = (pred_x0_l2 / prev_l2).cpu()
            if callback_on_step_end is not None:                                   # :1646-1659
                cb_in = {"latents": x}
                for name in (callback_on_step_end_tensor_inputs or []):            # (`prompt_embeds` here is the CFG-concatenated context)
                    if name == "prompt_embeds":
                        cb_in[name] = ctx
                    elif name == "negative_prompt_embeds":
                        cb_in[name] = negative_prompt_embeds
                cb = callback_on_step_end(self, i, t, cb_in)
                x = cb.pop("latents", x)
                new_ctx = cb.pop("prompt_embeds", None)
                cb.pop("negative_prompt_embeds", None)       # the reference rebinds a name it never reads again (:1465, :1655): no effect
                if new_ctx is not None and new_ctx is not ctx:
                    # the text K / V^T of all 70 cross-attention blocks are hoisted per call: a replaced context means hoisting
                    # again (and a new captured step), after which the loop continues on the same latents
                    if tuple(new_ctx.shape) != tuple(ctx.shape):
                        raise ValueError(f"callback_on_step_end returned prompt_embeds of shape {tuple(new_ctx.shape)}, expected {tuple(ctx.shape)}")
                    ctx = new_ctx
                    st = self._unet.prepare(ctx, pooled, time_ids, self._unet.resampler(img), Hl, Wl)
                    if self._unet_prev is not None:
                        st_prev = self._unet_prev.prepare(ctx, pooled, time_ids, self._unet_prev.resampler(img), Hl, Wl)
                    mean_keep = loop.previewer_mean
                    loop = _DenoiseLoop(self, B, rep, Hl, Wl, st, st_prev, st_agg, lq, reference_latents, previewer_scheduler,
                                        guidance_rescale=guidance_rescale)
                    loop.previewer_mean = mean_keep
        latents_out = x
        if output_type == "latent":
            image_out = latents_out
        else:
            if self.vae is None:
                raise NotImplementedError("output_type other than 'latent' needs a VAE attached to the pipeline")
            image_out = self.vae.decode_latent(latents_out, output_type)          # tiles when `vae.enable_tiling()` is on
        if save_preview_row and self.vae is not None and output_type != "latent":     # :1706-1729 (decoded independently, Q4)
            preview_row = [self.vae.decode_latent(pl, output_type) for pl in preview_row]
        if not return_dict:
            return (image_out, preview_row) if save_preview_row else (image_out,)
        return StableDiffusionXLPipelineOutput(images=image_out)


def _copy_state(dst, src):
    """Refresh a `prepare()` state in place (same structure, shapes and dtypes) so that launch sequences captured on `dst`'s
    tensors see `src`'s values.  `ada_jobs` is a device table of pointers into `dst`'s own tensors and stays.  False = the
    two states differ in structure: the caller builds a new loop."""
    if dst is None or src is None:
        return dst is None and src is None
    if len(dst) != len(src):
        return False
    for k, v in src.items():
        if k not in dst:
            return False
        d = dst[k]
        if k == "ada_jobs":
            continue
        if isinstance(v, dict):
            if not isinstance(d, dict) or not _copy_state(d, v):
                return False
        elif isinstance(v, torch.Tensor):
            if not isinstance(d, torch.Tensor) or d.shape != v.shape or d.dtype != v.dtype or d.device != v.device:
                return False
            d.copy_(v)
        elif d != v:
            return False
    return True


class _DenoiseLoop:
    """Device buffers + (optionally hipGraph-captured) launch sequences of one denoising step.
    Three phases exist (pipelines/sdxl_instantir.py:1542-1616): "preview" (UNet+LoRA -> LCM preview
    -> Aggregator -> UNet), "agg" (Aggregator on the LQ / reference latent -> UNet), "unet"."""

    def __init__(self, pipe, B, rep, H, W, st, st_prev, st_agg, lq, reference_latents, previewer_scheduler, guidance_rescale=0.0):
        dev = pipe.device
        self.guidance_rescale = float(guidance_rescale or 0.0)
        self.cfg_factor = torch.ones(B, dtype=torch.float32, device=dev)
        self.p, self.B, self.rep, self.H, self.W = pipe, B, rep, H, W
        self.st, self.st_prev, self.st_agg = st, st_prev, st_agg
        self.prev_sched = previewer_scheduler
        R, HW = B * rep, H * W
        self.lat16 = torch.zeros(R * HW, CPAD, dtype=F16, device=dev)
        self.prev16 = torch.zeros(R * HW, CPAD, dtype=F16, device=dev)
        self.lq16 = torch.zeros(R * HW, CPAD, dtype=F16, device=dev)
        ops.pack_latent(lq, self.lq16, rep=rep)
        self.ref16 = None
        if reference_latents is not None:
            self.ref16 = torch.zeros(R * HW, CPAD, dtype=F16, device=dev)
            ops.pack_latent(reference_latents.to(dev, torch.float32).contiguous(), self.ref16, rep=rep)
        self.x_in = torch.empty(B, 4, H, W, dtype=torch.float32, device=dev)
        self.x_out = torch.empty_like(self.x_in)
        self.x0 = torch.empty_like(self.x_in)
        self.noise = torch.zeros_like(self.x_in)
        self.preview_f32 = torch.zeros(R, 4, H, W, dtype=torch.float32, device=dev)
        self.previewer_mean = torch.zeros_like(self.x_in)
        # per-step scalars: [t x R | lcm coef x4 | sched coef x8 | res scale x R]
        self.n_sc = R + 4 + 8 + R
        # ring of pinned staging rows: a row is rewritten only after the H2D copy that read it has completed
        self.sc_ring = [torch.zeros(self.n_sc, dtype=torch.float32).pin_memory() for _ in range(8)]
        self.sc_events = [None] * 8
        self.sc_idx = 0
        self.sc_dev = torch.zeros(self.n_sc, dtype=torch.float32, device=dev)
        self.t_dev = self.sc_dev[:R].view(R, 1)
        self.lcm_coef = self.sc_dev[R:R + 4]
        self.sched_coef = self.sc_dev[R + 4:R + 12]
        self.res_scale = self.sc_dev[R + 12:]
        self.graphs = {}
        self.side = None

    def adopt(self, st, st_prev, st_agg, lq, reference_latents, previewer_scheduler):
        """Re-use this loop -- its buffers and captured graphs -- for another call of the same geometry: the new call's hoisted
        state is copied into the tensors the graphs were captured on.  False when the states do not line up."""
        if (self.ref16 is None) != (reference_latents is None):
            return False
        if not (_copy_state(self.st, st) and _copy_state(self.st_prev, st_prev) and _copy_state(self.st_agg, st_agg)):
            return False
        self.prev_sched = previewer_scheduler
        ops.pack_latent(lq, self.lq16, rep=self.rep)
        if self.ref16 is not None:
            ops.pack_latent(reference_latents.to(self.x_in.device, torch.float32).contiguous(), self.ref16, rep=self.rep)
        self.previewer_mean = torch.zeros_like(self.x_in)
        self.cfg_factor.fill_(1.0)
        return True

    def _launch(self, mode, use_noise, want_x0, want_preview):
        p, B, rep = self.p, self.B, self.rep
        ops.pack_latent(self.x_in, self.lat16, rep=rep)                      # cat([latents]*2), :1503
        down = mid = None
        if mode == "unet_res":       # stale residuals of the last Aggregator pass, re-scaled (see __call__)
            eps = p._unet.forward(self.lat16, self.t_dev, self.st, p._agg._out, p._agg._out_mid, self.res_scale)
            self._sched(eps, use_noise, want_x0)
            return
        if mode != "unet" and p.overlap_streams:
            # The main UNet's encoder half does not depend on the previewer / Aggregator: run it on a side
            # stream so its (CU-underfilling) launches overlap theirs; join before the residual adds.
            main = torch.cuda.current_stream()
            if self.side is None:
                self.side = torch.cuda.Stream(device=self.x_in.device)
            fork, join = torch.cuda.Event(), torch.cuda.Event()
            fork.record(main)
            self.side.wait_event(fork)
            with torch.cuda.stream(self.side):
                enc = p._unet.encode(self.lat16, self.t_dev, self.st)
                join.record(self.side)
            if mode == "preview":
                eps1 = p._unet_prev.forward(self.lat16, self.t_dev, self.st_prev)
                ops.lcm_step(eps1, B, rep, self.lcm_coef, self.x_in, self.prev16, self.preview_f32 if want_preview else None)
                cond = self.prev16
            else:
                cond = self.ref16 if self.ref16 is not None else self.lq16
            p._agg.defer_shallow = p.overlap_sft
            down, mid = p._agg.forward(self.lq16, cond, self.t_dev, self.st_agg)
            main.wait_event(join)
            late = None
            if p.overlap_sft:
                # the SFT heads of the shallow skips (consumed by the last up blocks) run on the side stream beside the
                # decoder's first up block
                f2, late = torch.cuda.Event(), torch.cuda.Event()
                f2.record(main)
                self.side.wait_event(f2)
                with torch.cuda.stream(self.side):
                    p._agg.late_heads()
                    late.record(self.side)
            eps = p._unet.decode(enc, self.st, down, mid, self.res_scale, late_event=late)
            self._sched(eps, use_noise, want_x0)
            return
        if mode != "unet":
            if mode == "preview":
                eps1 = p._unet_prev.forward(self.lat16, self.t_dev, self.st_prev)          # :1545-1554
                ops.lcm_step(eps1, B, rep, self.lcm_coef, self.x_in, self.prev16,
                             self.preview_f32 if want_preview else None)                 # :1555-1561
                cond = self.prev16
            else:
                cond = self.ref16 if self.ref16 is not None else self.lq16               # :1579-1582
            p._agg.defer_shallow = False
            down, mid = p._agg.forward(self.lq16, cond, self.t_dev, self.st_agg)           # :1591-1599
        eps = p._unet.forward(self.lat16, self.t_dev, self.st, down, mid, self.res_scale if down is not None else None)
        self._sched(eps, use_noise, want_x0)

    def _sched(self, eps, use_noise, want_x0):
        """CFG (+ rescale_noise_cfg when guidance_rescale > 0, :181-192) + scheduler step, :1619-1633."""
        B, rep = self.B, self.rep
        fac = None
        if rep == 2 and self.guidance_rescale > 0.0:
            fac = ops.cfg_rescale_factor(eps, B, self.sched_coef, self.x_in, self.guidance_rescale, self.cfg_factor)
        ops.sched_step(eps, B, self.sched_coef, self.x_in, self.x_out, noise=self.noise if use_noise else None,
                       cfg=rep == 2, x0_out=self.x0 if want_x0 else None, eps_factor=fac)

    def step(self, mode, t, x, res_scale_rows, guidance, eta, noise, generator, want_x0=False, want_preview=False):
        p, R = self.p, self.B * self.rep
        slot = self.sc_idx % len(self.sc_ring)
        self.sc_idx += 1
        if self.sc_events[slot] is not None:
            self.sc_events[slot].synchronize()
        sc = self.sc_ring[slot]
        sc[:R] = float(t)
        if mode == "preview":
            sc[R:R + 4] = torch.tensor(self.prev_sched.preview_coefficients(t))
        coef = p.scheduler.step_coefficients(t, eta=eta)
        coef[0] = float(guidance)
        sc[R + 4:R + 12] = torch.tensor(coef)
        sc[R + 12:] = res_scale_rows.float()
        use_noise = coef[6] != 0.0
        if use_noise:
            if noise is None:
                gdev = generator.device if generator is not None else self.x_in.device
                noise = torch.randn(self.x_in.shape, generator=generator, device=gdev, dtype=torch.float32)
            self.noise.copy_(noise.to(self.noise.device, torch.float32), non_blocking=True)
        self.sc_dev.copy_(sc, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.sc_events[slot] = ev
        self.x_in.copy_(x)
        key = (mode, use_noise, want_x0, want_preview)
        if p.use_graphs:
            g = self.graphs.get(key)
            if g is None:
                self._launch(*key)                        # warm-up: one-time attribute / workspace setup
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch(*key)
                self.graphs[key] = g
            g.replay()
        else:
            self._launch(*key)
        x.copy_(self.x_out)
        return self.x0 if want_x0 else None
