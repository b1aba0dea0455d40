"""Generate golden input/output vectors by RUNNING the reference's own importable modules.

Run in the build container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_goldens.py

Outputs (committed, small): tests/golden/ip_adapter.npz, lcm_scheduler.npz, min_sdxl.npz, sft.npz.
Nothing here travels except those data files: inputs, the reference modules' own randomly
initialised parameters (or the seed + inventory that re-creates them, tests/golden/seeded.py), and
the reference's outputs.

Importable as-is: module.ip_adapter.{resampler,attention_processor,ip_adapter}.
`schedulers.lcm_single_step_scheduler` imports four names from the third-party `diffusers`
(absent here).  Following SURVEY.md section 8c it is imported behind a shim that supplies ONLY config
plumbing (register_to_config / ConfigMixin / SchedulerMixin / BaseOutput / logger /
randn_tensor) and no arithmetic, so every number in lcm_scheduler.npz is computed by the
reference file itself.  The shim exists only inside this generator process.

`module.min_sdxl` (the in-tree hard-coded SDXL UNet and its leaf blocks) imports two names from
`diffusers.models.attention_processor`: AttnProcessor and AttnProcessor2_0.  The reference ships
both classes itself in module/ip_adapter/attention_processor.py, so the shim RE-EXPORTS those
reference classes under the diffusers module name (zero arithmetic of its own) and every number in
min_sdxl.npz is computed by reference code.  `module.aggregator` imports a list of diffusers names
at module level; `SFT` and `zero_module` in that file are plain torch.  The shim supplies name-only
placeholders (empty classes, never called) so the file imports and `SFT` can be run; the Aggregator
class itself needs diffusers' blocks and is NOT built.
"""
import os
import sys
import types
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def np_sd(mod, prefix=""):
    return {prefix + k: v.detach().numpy() for k, v in mod.state_dict().items()}


class _AttnStub(torch.nn.Module):
    """The `attn` argument the reference processors expect (a diffusers `Attention` module):
    only holds the four projections and the attributes the processors read."""

    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = torch.nn.Linear(dim, dim, bias=False)
        self.to_k = torch.nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = torch.nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(dim, dim), torch.nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0


def gen_ip_adapter():
    from module.ip_adapter.attention_processor import AdaLayerNorm, AttnProcessor2_0, TA_IPAttnProcessor2_0
    from module.ip_adapter.ip_adapter import MultiIPAdapterImageProjection
    from module.ip_adapter.resampler import Resampler

    torch.manual_seed(1234)
    out = {}
    with torch.no_grad():
        # --- Resampler + MultiIPAdapterImageProjection (small geometry, same code path)
        rs = Resampler(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=64, output_dim=128,
                       ff_mult=4)
        for p in rs.parameters():          # LayerNorm affine defaults are 1/0: perturb so they matter
            if p.ndim == 1:
                p.add_(0.1 * torch.randn_like(p))
        proj = MultiIPAdapterImageProjection([rs])
        x = torch.randn(2, 3, 21, 64)      # (n=2 [neg;pos], B=3, S, E)
        y = proj([x])[0]
        out.update(np_sd(rs, "rs."))
        out["rs_in"] = x.numpy()
        out["rs_out"] = y.numpy()

        # --- AdaLayerNorm (zero-init linear perturbed)
        aln = AdaLayerNorm(128, 256)
        aln.linear.weight.normal_(0, 0.05)
        aln.linear.bias.normal_(0, 0.05)
        ax, at = torch.randn(2, 16, 128), torch.randn(2, 256)
        out.update(np_sd(aln, "aln."))
        out["aln_x"], out["aln_t"], out["aln_out"] = ax.numpy(), at.numpy(), aln(ax, at).numpy()

        # --- AttnProcessor2_0 (self-attention)
        attn = _AttnStub(128, 128, heads=2)
        hs = torch.randn(2, 40, 128)
        out.update(np_sd(attn, "sa."))
        out["sa_x"] = hs.numpy()
        out["sa_out"] = AttnProcessor2_0()(attn, hs, temb=torch.randn(2, 256)).numpy()

        # --- TA_IPAttnProcessor2_0, tuple input (ctx, [ip_tokens])
        attn2 = _AttnStub(128, 96, heads=2)
        proc = TA_IPAttnProcessor2_0(128, 96, time_embedding_dim=256, num_tokens=16)
        for ln in (proc.ln_k_ip, proc.ln_v_ip):
            ln.linear.weight.normal_(0, 0.05)
            ln.linear.bias.normal_(0, 0.05)
        ctx, ip, temb = torch.randn(2, 13, 96), torch.randn(2, 16, 96), torch.randn(2, 256)
        out.update(np_sd(attn2, "ca."))
        out.update(np_sd(proc, "ca.processor."))
        out["ca_x"], out["ca_ctx"], out["ca_ip"], out["ca_temb"] = hs.numpy(), ctx.numpy(), ip.numpy(), temb.numpy()
        out["ca_out"] = proc(attn2, hs, encoder_hidden_states=(ctx, [ip]), temb=temb).numpy()
        # concatenated-tensor input form (attention_processor.py:1118-1123)
        out["ca_out_cat"] = proc(attn2, hs, encoder_hidden_states=torch.cat([ctx, ip], dim=1), temb=temb).numpy()
    np.savez_compressed(os.path.join(OUT, "ip_adapter.npz"), **out)
    print("ip_adapter.npz:", len(out), "arrays")


def _install_diffusers_shim():
    """Config plumbing only -- no arithmetic (see module docstring)."""
    import functools
    import inspect
    import logging as pylog

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    def register_to_config(init):
        @functools.wraps(init)
        def wrapper(self, *a, **kw):
            sig = inspect.signature(init)
            bound = sig.bind(self, *a, **kw)
            bound.apply_defaults()
            self.config = _Cfg({k: v for k, v in bound.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return wrapper

    class ConfigMixin:
        @classmethod
        def from_config(cls, cfg, **kw):
            names = set(inspect.signature(cls.__init__).parameters) - {"self"}
            args = {k: v for k, v in dict(cfg).items() if k in names}
            args.update(kw)
            return cls(**args)

    class SchedulerMixin:
        pass

    @dataclass
    class BaseOutput(OrderedDict):
        pass

    class _Log:
        @staticmethod
        def get_logger(name):
            return pylog.getLogger(name)

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        return torch.randn(shape, generator=generator, dtype=dtype)

    d = types.ModuleType("diffusers")
    cu = types.ModuleType("diffusers.configuration_utils")
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    ut = types.ModuleType("diffusers.utils")
    ut.BaseOutput, ut.logging = BaseOutput, _Log
    tu = types.ModuleType("diffusers.utils.torch_utils")
    tu.randn_tensor = randn_tensor
    sc = types.ModuleType("diffusers.schedulers")
    su = types.ModuleType("diffusers.schedulers.scheduling_utils")
    su.SchedulerMixin = SchedulerMixin
    for name, m in [("diffusers", d), ("diffusers.configuration_utils", cu), ("diffusers.utils", ut),
                    ("diffusers.utils.torch_utils", tu), ("diffusers.schedulers", sc),
                    ("diffusers.schedulers.scheduling_utils", su)]:
        sys.modules[name] = m


def gen_lcm():
    _install_diffusers_shim()
    from schedulers.lcm_single_step_scheduler import LCMSingleStepScheduler

    # the effective SDXL scheduler_config.json keys
    s = LCMSingleStepScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                               timestep_spacing="leading", set_alpha_to_one=False)
    out = {"alphas_cumprod": s.alphas_cumprod.numpy()}
    ts = np.array([999, 958, 925, 499, 34, 1, 0], dtype=np.int64)
    out["t"] = ts
    cs, co = s.get_scalings_for_boundary_condition_discrete(torch.from_numpy(ts))
    out["c_skip"], out["c_out"] = cs.numpy(), co.numpy()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g)
    e = torch.randn(2, 4, 8, 8, generator=g)
    out["x"], out["eps"] = x.numpy(), e.numpy()
    out["step"] = np.stack([s.step(e, torch.tensor(int(t)), x, return_dict=False)[0].numpy() for t in ts])
    out["add_noise"] = np.stack([s.add_noise(x, e, torch.tensor([int(t)] * 2)).numpy() for t in ts])
    for n in (1, 2, 4, 8):
        s.set_timesteps(n)
        out[f"set_timesteps_{n}"] = s.timesteps.numpy()
    np.savez_compressed(os.path.join(OUT, "lcm_scheduler.npz"), **out)
    print("lcm_scheduler.npz:", len(out), "arrays;", "step(958)[:4] =", out["step"][1].ravel()[:4])


def _extend_shim_for_min_sdxl_and_aggregator():
    """Re-export the reference's own AttnProcessor classes; name-only placeholders for the rest."""
    from module.ip_adapter import attention_processor as ref_ap

    models = types.ModuleType("diffusers.models")
    ap = types.ModuleType("diffusers.models.attention_processor")
    ap.AttnProcessor, ap.AttnProcessor2_0 = ref_ap.AttnProcessor, ref_ap.AttnProcessor2_0
    sys.modules["diffusers.models"] = models
    sys.modules["diffusers.models.attention_processor"] = ap

    def placeholder(modname, names):
        m = sys.modules.get(modname) or types.ModuleType(modname)
        for n in names:
            if not hasattr(m, n):
                setattr(m, n, type(n, (), {}))
        sys.modules[modname] = m

    placeholder("diffusers.loaders", [])
    placeholder("diffusers.loaders.single_file_model", ["FromOriginalModelMixin"])
    placeholder("diffusers.models.attention_processor",
                ["ADDED_KV_ATTENTION_PROCESSORS", "CROSS_ATTENTION_PROCESSORS", "AttentionProcessor",
                 "AttnAddedKVProcessor"])
    placeholder("diffusers.models.embeddings", ["TextImageProjection", "TextImageTimeEmbedding", "TextTimeEmbedding",
                                                "TimestepEmbedding", "Timesteps"])
    placeholder("diffusers.models.modeling_utils", ["ModelMixin"])
    placeholder("diffusers.models.unets", [])
    placeholder("diffusers.models.unets.unet_2d_blocks", ["CrossAttnDownBlock2D", "DownBlock2D", "UNetMidBlock2D",
                                                          "UNetMidBlock2DCrossAttn", "get_down_block"])
    placeholder("diffusers.models.unets.unet_2d_condition", ["UNet2DConditionModel"])


def _load_seeded(mod, seed, zero=()):
    from seeded import pack_inventory, seeded_fill
    sd = mod.state_dict()
    names, shapes = list(sd.keys()), [tuple(v.shape) for v in sd.values()]
    mod.load_state_dict(seeded_fill(names, shapes, seed, zero), strict=True)
    return pack_inventory(sd)


def gen_min_sdxl():
    """Leaf blocks and block wrappers of module/min_sdxl.py at small channel counts (its hard-wired
    constants stay: temb 1280, cross-attention dim 2048, head_dim 64, 32 groups), then ONE forward of its
    hard-coded SDXL-base `UNet2DConditionModel` (2.6 B seeded parameters) on a 16x16 latent."""
    sys.path.insert(0, OUT)
    import module.min_sdxl as M

    g = torch.Generator().manual_seed(4321)

    def rn(*shape, s=1.0):
        return torch.randn(*shape, generator=g) * s

    out = {}

    def case(tag, mod, seed, *args, **kw):
        # min_sdxl.Attention hands itself to the reference's AttnProcessor2_0, which reads the configuration
        # attributes of a diffusers `Attention` (attention_processor.py:348-412).  Set them to the values SDXL's
        # transformer blocks use -- attributes only, as `_AttnStub` above does; all arithmetic stays reference code.
        for m in mod.modules():
            if isinstance(m, M.Attention):
                m.spatial_norm, m.group_norm, m.norm_cross = None, None, False
                m.heads, m.residual_connection, m.rescale_output_factor = m.num_heads, False, 1.0
        out[tag + "__inv"] = np.array(_load_seeded(mod, seed))
        out[tag + "__seed"] = np.int64(seed)
        with torch.no_grad():
            y = mod(*args, **kw)
        if isinstance(y, (tuple, list)):
            ys = []
            for v in y:
                ys += list(v) if isinstance(v, (tuple, list)) else [v]
            for i, v in enumerate(ys):
                out[f"{tag}__out{i}"] = v.numpy()
        else:
            out[tag + "__out"] = y.numpy()

    with torch.no_grad():
        t = torch.tensor([0, 1, 34, 499, 958, 999])
        out["ts_t"] = t.numpy()
        out["ts_320"] = M.Timesteps()(t).numpy()
        out["ts_256"] = M.Timesteps(256)(torch.tensor([1024., 768., 0., 3.5])).numpy()

    x = rn(2, 32); out["te_x"] = x.numpy()
    case("te", M.TimestepEmbedding(32, 64), 11, x)

    temb = rn(2, 1280); out["temb"] = temb.numpy()
    x64 = rn(2, 64, 8, 8); out["x64"] = x64.numpy()
    x128 = rn(2, 128, 8, 8); out["x128"] = x128.numpy()
    ctx = rn(2, 13, 2048); out["ctx"] = ctx.numpy()
    tok = rn(2, 24, 128); out["tok"] = tok.numpy()

    case("res_sc", M.ResnetBlock2D(64, 128), 12, x64, temb)
    case("res_id", M.ResnetBlock2D(64, 64, conv_shortcut=False), 13, x64, temb)
    case("geglu", M.GEGLU(128, 512), 14, tok)
    case("ff", M.FeedForward(128, 128), 15, tok)
    case("btb", M.BasicTransformerBlock(128), 16, tok, ctx)
    case("t2d", M.Transformer2DModel(128, 128, 2), 17, x128, ctx)
    case("down", M.Downsample2D(64, 64), 18, x64)
    case("up", M.Upsample2D(64, 64), 19, x64)
    case("dblk", M.DownBlock2D(64, 64), 20, x64, temb)
    case("cadb", M.CrossAttnDownBlock2D(64, 128, 1), 21, x64, temb, ctx)
    case("cadb_nods", M.CrossAttnDownBlock2D(128, 128, 1, has_downsamplers=False), 22, x128, temb, ctx)
    s_a, s_b, s_c = rn(2, 64, 8, 8), rn(2, 128, 8, 8), rn(2, 128, 8, 8)
    out["caub_skips"] = np.stack([s_b.numpy(), s_c.numpy()]); out["caub_skip0"] = s_a.numpy()
    case("caub", M.CrossAttnUpBlock2D(64, 128, 128, 1), 23, hidden_states=x128, res_hidden_states_tuple=[s_a, s_b, s_c],
         temb=temb, encoder_hidden_states=ctx)
    u_a, u_b, u_c = rn(2, 64, 8, 8), rn(2, 64, 8, 8), rn(2, 64, 8, 8)
    out["upb_skips"] = np.stack([u_a.numpy(), u_b.numpy(), u_c.numpy()])
    case("upb", M.UpBlock2D(64, 64, 128), 24, x128, [u_a, u_b, u_c], temb)
    xm = rn(2, 128, 4, 4); out["xm"] = xm.numpy()
    case("mid", M.UNetMidBlock2DCrossAttn(128), 25, xm, temb, ctx)

    # ---- the hard-coded SDXL-base UNet, one forward, 16x16 latent, 77 context tokens, 2 rows
    unet = M.UNet2DConditionModel()
    sample = rn(2, 4, 16, 16); ctx77 = rn(2, 77, 2048); pooled = rn(2, 1280)
    time_ids = torch.tensor([[128., 128., 0., 0., 128., 128.]] * 2)
    out["unet_sample"], out["unet_ctx"], out["unet_pooled"], out["unet_time_ids"] = (
        sample.numpy(), ctx77.numpy(), pooled.numpy(), time_ids.numpy())
    out["unet_t"] = np.int64(499)
    case("unet", unet, 26, sample, torch.tensor([499]), ctx77, {"text_embeds": pooled, "time_ids": time_ids})
    np.savez_compressed(os.path.join(OUT, "min_sdxl.npz"), **out)
    print("min_sdxl.npz:", len(out), "arrays; unet out std", float(out["unet__out0"].std()))


def gen_sft():
    """SFT.forward (module/aggregator.py:60-90) followed by the zero-initialised 1x1 conv it is wrapped
    with (`nn.Sequential(SFT, zero_module(Conv2d 1x1))`, :414-417), 1x1 given non-zero seeded values."""
    import module.aggregator as A

    g = torch.Generator().manual_seed(777)
    out = {}
    head = torch.nn.Sequential(A.SFT(64, 64), A.zero_module(torch.nn.Conv2d(64, 64, 1)))
    assert float(head[1].weight.abs().sum()) == 0.0          # zero_module really zeroes
    out["inv"] = np.array(_load_seeded(head, 31))
    out["seed"] = np.int64(31)
    c, h = torch.randn(2, 64, 8, 6, generator=g), torch.randn(2, 64, 8, 6, generator=g)
    out["c"], out["h"] = c.numpy(), h.numpy()
    with torch.no_grad():
        out["out"] = head((c, h)).numpy()
    np.savez_compressed(os.path.join(OUT, "sft.npz"), **out)
    print("sft.npz:", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["ip", "lcm", "min_sdxl", "sft"]
    if "ip" in which:
        gen_ip_adapter()
    _install_diffusers_shim()
    if "lcm" in which:
        gen_lcm()
    _extend_shim_for_min_sdxl_and_aggregator()
    if "min_sdxl" in which:
        gen_min_sdxl()
    if "sft" in which:
        gen_sft()
