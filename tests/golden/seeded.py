"""Seeded parameter fill shared by the golden generator and the tests that re-create its weights.

Several fixtures (the full min_sdxl `UNet2DConditionModel`: 2.6 B parameters) are too large to commit,
so the generator fills the REFERENCE module's own `state_dict()` entries from one seeded
`torch.Generator` and commits only (seed, names, shapes, inputs, the reference's outputs).  A test
re-creates bit-identical parameters with the same function (same torch build here and on the GPU
box) and feeds them to the oracle or to the HIP engine.  Pure data plumbing: no reference code.
"""
from __future__ import annotations

import json

import numpy as np
import torch

_DAMPED = ("to_out.0.weight", "ff.net.2.weight", "conv2.weight", "proj_out.weight", "to_out.weight", ".3.weight")


def seeded_fill(names, shapes, seed, zero=()):
    """name -> float32 CPU tensor.  Matrices / conv kernels: N(0, 1/fan_in) (x0.5 where the layer feeds a
    residual sum); 1-D `weight` (norm gains): 1 + 0.1 N(0,1); biases: 0.02 N(0,1).  Names containing an
    entry of `zero` are zero-filled (still drawn, so the stream does not depend on `zero`)."""
    g = torch.Generator()
    g.manual_seed(int(seed))
    out = {}
    for name, shape in zip(names, shapes):
        shape = tuple(int(s) for s in shape)
        t = torch.randn(shape, generator=g, dtype=torch.float32)
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t.mul_((0.5 if name.endswith(_DAMPED) else 1.0) / fan_in ** 0.5)
        elif name.endswith("weight"):
            t.mul_(0.1).add_(1.0)
        else:
            t.mul_(0.02)
        if any(z in name for z in zero):
            t.zero_()
        out[name] = t
    return out


def pack_inventory(state_dict):
    """(names, shapes) of a module's state_dict as one JSON string (npz-storable)."""
    return json.dumps([[k, list(v.shape)] for k, v in state_dict.items()])


def unpack_inventory(blob):
    inv = json.loads(str(np.asarray(blob).item()) if not isinstance(blob, str) else blob)
    return [k for k, _ in inv], [tuple(s) for _, s in inv]
