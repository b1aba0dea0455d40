"""Residency probe: attention at T = 4096 with 5..60 (batch x head) pairs (32 workgroups each): time vs workgroup count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import ops
dev = torch.device("cuda:0")
T = int(os.environ.get("T", "4096"))
print("IIR_ATTN_V =", os.environ.get("IIR_ATTN_V", "(default)"), "T =", T)
for h in (4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48):
    B, C = 1, h * 64
    q, k, vt = (torch.randn(B * T, C, device=dev).half() for _ in range(2)) .__iter__().__next__(), None, None
    q = torch.randn(B * T, C, device=dev).half(); k = torch.randn(B * T, C, device=dev).half(); vt = torch.randn(C, B * T, device=dev).half()
    o = torch.empty(B * T, C, dtype=torch.half, device=dev)
    for _ in range(5):
        ops.attention(q, o, [(k, T, vt, T, T)], B, h, T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(q, o, [(k, T, vt, T, T)], B, h, T)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    wgs = h * (T // 128)
    print(f"pairs {h:3d}  workgroups {wgs:5d} ({wgs / 256:.2f} per CU): {us:8.1f} us   {4.0 * h * T * T * 64 / us / 1e6:7.1f} TFLOP/s", flush=True)
