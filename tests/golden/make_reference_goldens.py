"""Generate golden input/output vectors by RUNNING the reference's own importable modules.

Run in the build container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_goldens.py

Outputs (committed, small): tests/golden/ip_adapter.npz, tests/golden/lcm_scheduler.npz.
Nothing here travels except those data files: inputs, the reference modules' own randomly
initialised parameters, and the reference's outputs.

Importable as-is: module.ip_adapter.{resampler,attention_processor,ip_adapter}.
`schedulers.lcm_single_step_scheduler` imports four names from the third-party `diffusers`
(absent here).  Following SURVEY.md section 8c it is imported behind a shim that supplies ONLY config
plumbing (register_to_config / ConfigMixin / SchedulerMixin / BaseOutput / logger /
randn_tensor) and no arithmetic, so every number in lcm_scheduler.npz is computed by the
reference file itself.  The shim exists only inside this generator process.
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


if __name__ == "__main__":
    gen_ip_adapter()
    gen_lcm()
