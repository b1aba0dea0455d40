"""A few attention launches per shape for rocprofv3 --pmc passes (kernel generation chosen by IIR_ATTN_V)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).half().to(dev)
for B, h, T in [(2, 10, 4096), (2, 20, 1024), (2, 10, 8192)]:
    C = h * 64
    q, k, vt = rnd(B * T, C), rnd(B * T, C), rnd(C, B * T)
    o = torch.empty(B * T, C, dtype=torch.half, device=dev)
    for _ in range(3):
        ops.attention(q, o, [(k, T, vt, T, T)], B, h, T)
    torch.cuda.synchronize()
