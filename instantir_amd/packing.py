"""Host-side weight re-layout for the HIP kernels (one-time, at load)."""
from __future__ import annotations

import torch


def pair_rows(value: torch.Tensor, partner: torch.Tensor) -> torch.Tensor:
    """Interleave two (n, ...) row blocks in groups of 8 rows: [V0..7, G0..7, V8..15, G8..15, ...].

    This is the layout the GEGLU / SFT epilogues of iir_gemm_f16 / iir_conv2d_nhwc_f16 expect
    (include/instantir_hip.h "Pair permutation"): inside every 16-column MFMA tile the value columns sit in the
    lower lane half and their partners 32 lanes away, one cross-half exchange apart."""
    n = value.shape[0]
    if partner.shape != value.shape or n % 8:
        raise ValueError("pair_rows needs equal shapes with a multiple of 8 rows")
    v = value.reshape(n // 8, 1, 8, *value.shape[1:])
    g = partner.reshape(n // 8, 1, 8, *partner.shape[1:])
    return torch.cat([v, g], dim=1).reshape(2 * n, *value.shape[1:]).contiguous()


def conv_weight_nhwc(w: torch.Tensor, cin_pad: int | None = None) -> torch.Tensor:
    """(Cout, Cin, k, k) torch layout -> (Cout, k, k, Cin[_pad]) with zero-filled channel padding."""
    w = w.permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad > w.shape[3]:
        w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[3]))
    return w.contiguous()

