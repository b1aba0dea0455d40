"""Thin launch wrappers: torch CUDA tensors in, one C-ABI call out.  torch is used for device
memory and the current stream only; no arithmetic happens here.

All activations are fp16, channel/feature dimension contiguous.  Matrices are 2-D views whose
row stride may exceed the width (column slices of wider buffers are passed as views).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L
from .lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_SILU, EPI_GEGLU, EPI_PLAIN, EPI_SFT, EPI_XATTN  # noqa: F401


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


_DT = {torch.float16: 0, torch.bfloat16: 1}       # IIR_DT_F16 / IIR_DT_BF16


def _chk2d(t, name, dtype=torch.float16):
    """2-D CUDA tensor of `dtype` (fp16 unless the call runs the bf16 build: the VAE) with unit column stride."""
    if t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise ValueError(f"{name}: expected a 2-D {dtype} CUDA tensor with unit column stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")


class LaunchProfiler:
    """Optional per-launch timing of the MFMA kernels (bench.py's roofline leg): each launch is issued with a HIP
    start/stop event pair that the runtime stamps with the kernel's own begin / end timestamps on its stream
    (`iir_timing_arm`, hipExtLaunchKernelGGL).  Records (kernel class, algorithmic FLOPs, start, stop)."""

    def __init__(self):
        self.records = []
        self._pool = []

    def events(self):
        h = L.load()
        e0, e1 = h.iir_timing_event_create(), h.iir_timing_event_create()
        if not e0 or not e1:
            raise L.HipLibraryError("hipEventCreate failed")
        self._pool += [e0, e1]
        return e0, e1

    def summary(self):
        torch.cuda.synchronize()
        h = L.load()
        out = {}
        us = C.c_float()
        for cls, flops, e0, e1, nbytes in self.records:
            L.check(h.iir_timing_elapsed_us(e0, e1, C.byref(us)), "iir_timing_elapsed_us")
            d = out.setdefault(cls, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += us.value * 1e-3
            d["flops"] += flops
            d["bytes"] += nbytes
        for e in self._pool:
            h.iir_timing_event_destroy(e)
        self._pool = []
        return out


PROFILER = None     # set to a LaunchProfiler to time gemm / conv / attention launches

_TILE_NAMES = {1: "128x128", 2: "128x64", 3: "64x64", 4: "128x160", 5: "64x160", 6: "256x128", 7: "128x320", 8: "256x256", 9: "32x160", 0: "256x320"}   # 0: tile id 90


def auto_tile(M, N, paired=False, K=0):
    """The tile `tile=0` resolves to (single source of truth: pick_tile() in csrc/gemm_conv.hip)."""
    return L.load().iir_gemm_pick_tile(M, N, K, int(paired))


class _Timed:
    def __init__(self, cls, flops, nbytes=0.0):
        self.cls, self.flops, self.nbytes = cls, flops, nbytes      # algorithmic FLOPs and bytes (each operand once)

    def __enter__(self):
        if PROFILER is not None:
            e0, e1 = PROFILER.events()
            L.check(L.load().iir_timing_arm(e0, e1), "iir_timing_arm")
            PROFILER.records.append((self.cls, self.flops, e0, e1, self.nbytes))

    def __exit__(self, *a):
        pass


_zero_pages = {}


def zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.float16, device=device)
        _zero_pages[device] = z
    return z


def splitk_workspace(M, N, device):
    """Zeroed split-K workspace for (M, N) problems (None when the 2-slice form does not apply).  One per stream."""
    n = L.load().iir_gemm_splitk_workspace_bytes(M, N)
    return torch.zeros(n, dtype=torch.uint8, device=device) if n > 0 else None


FP8_MAX = 448.0          # largest finite E4M3 (OCP) value


class Fp8Weight:
    """A linear layer's weight as fp8-E4M3 bytes + per-output-channel fp32 scale; `gemm` accepts it in place of `w`."""

    def __init__(self, q, scale):
        self.q, self.scale = q, scale
        self.shape = q.shape

    def numel(self):
        return self.q.numel() // 2          # in fp16-element units: what the weight-arena bookkeeping counts

    def data_ptr(self):
        return self.q.data_ptr()


def quantize_fp8_rows(w):
    """Per-output-channel E4M3 quantisation of a weight matrix (N, K): returns (bytes (N, K) as torch.float8_e4m3fn,
    fp32 scale (N,)) with w ~= bytes * scale[:, None].  Pack-time plumbing for `gemm(..., wscale=)`."""
    w32 = w.float()
    scale = (w32.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX).contiguous()
    q = (w32 / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).contiguous()
    return q, scale


_FP8_DTYPES = (torch.float8_e4m3fn, torch.uint8)


def gemm_fp8(a8, w8, out, a_scale=1.0, bias=None, res=None, epi=EPI_PLAIN, act=ACT_NONE, tile=0, prefetch=None, out_t=None):
    """out = epi((a8 @ w8.q.T) * w8.scale[None, :] * a_scale ...): BOTH operands fp8-E4M3 (`iir_gemm_desc.a_fp8`).
    a8 (M, K) torch.float8_e4m3fn (or its bytes) view with a 16-byte-aligned row stride, K % 128 == 0; w8 an `Fp8Weight`;
    out / bias / res fp16.  One K tile is 128 K values = the same 128-byte rows the fp16 path stages for 64."""
    if not isinstance(w8, Fp8Weight):
        raise ValueError("gemm_fp8: w8 must be an Fp8Weight")
    if a8.dtype not in _FP8_DTYPES or a8.dim() != 2 or a8.stride(1) != 1 or not a8.is_cuda:
        raise ValueError("gemm_fp8: a8 must be a 2-D CUDA view of torch.float8_e4m3fn (or its bytes) with contiguous rows")
    M, K = a8.shape
    N = w8.q.shape[0]
    if w8.q.shape[1] != K or K % 128 or a8.stride(0) % 16:
        raise ValueError("gemm_fp8: K must match, K % 128 == 0, row stride % 16 == 0")
    dt = torch.float16
    c_fp8 = out.dtype in _FP8_DTYPES            # the output is the NEXT all-fp8 GEMM's A operand: stored as E4M3 bytes (fp16 rounding first)
    if c_fp8:
        if out.dim() != 2 or out.stride(1) != 1 or out.stride(0) % 8 or not out.is_cuda:
            raise ValueError("gemm_fp8: an fp8 `out` needs contiguous rows with a stride % 8 == 0")
    else:
        _chk2d(out, "out", dt)
    n_out = (N if epi == EPI_PLAIN else N // 2) if out_t is None else out_t[1]
    if out.shape != (M, n_out):
        raise ValueError(f"out shape {tuple(out.shape)} != {(M, n_out)}")
    for t_, n_ in ((bias, "bias"), (res, "res")):
        if t_ is not None and t_.dtype != dt:
            raise ValueError(f"{n_}: fp16 expected")
    d = L.GemmDesc()
    d.c_fp8 = int(c_fp8)
    if out_t is not None:                      # columns >= tr_from leave transposed (the V third of q|k|v), as in `gemm`
        ct, tr_from = out_t
        _chk2d(ct, "out_t", dt)
        if c_fp8 or ct.shape[0] < N - tr_from or ct.shape[1] < M:
            raise ValueError("out_t: fp16 (N - tr_from, M), with an fp16 `out`")
        d.Ct, d.ldct, d.tr_from = ct.data_ptr(), ct.stride(0), tr_from
    d.A, d.lda = a8.data_ptr(), a8.stride(0)
    d.W, d.wscale = w8.q.data_ptr(), w8.scale.data_ptr()
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.bias = _p(bias)
    if res is not None:
        _chk2d(res, "res", dt)
        d.res, d.ldr = res.data_ptr(), res.stride(0)
    d.epi, d.act, d.out_scale, d.tile = epi, act, 1.0, tile
    d.dtype, d.a_fp8, d.a_scale = _DT[dt], 1, float(a_scale)
    if prefetch is not None:
        d.prefetch, d.prefetch_bytes = prefetch
    t_name = tile if tile else auto_tile(M, N, epi != EPI_PLAIN, K)
    cls = "gemm_kernel<%s,gemm-f8>" % _TILE_NAMES[t_name % 10]
    if tile == 91 or (PROFILER is not None and tile == 0 and L.load().iir_gemm_resolve_tile(C.byref(d)) == 91):
        cls = "gemm8_kernel<256x320,gemm-f8>"
    with _Timed(cls, 2.0 * M * N * K, 1.0 * (M * K + N * K) + 2.0 * M * n_out * (2 if res is not None else 1)):
        L.check(L.load().iir_gemm_f16(C.byref(d), _stream()), "iir_gemm_f16")
    return out


def fp8_out_supported(M, N, K, paired=False):
    """Can the all-fp8 launch of (M, N, K) store its result as fp8 bytes (an fp8 `out` of `gemm_fp8`)?"""
    return bool(L.load().iir_gemm_fp8_out_supported(M, N, K, int(paired)))


def quantize_fp8_tensor(x):
    """Per-tensor E4M3 quantisation of an activation matrix: (bytes as torch.float8_e4m3fn, scale) with x ~= bytes * scale."""
    x32 = x.float()
    scale = float(x32.abs().amax().clamp_min(1e-12) / FP8_MAX)
    return (x32 / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).contiguous(), scale


class LnFold:
    """A LayerNorm folded into the nn.Linear that follows it (weight-pack time): `w` = W . diag(gamma) (fp16, or an Fp8Weight of
    it), `colsum[n]` = sum_k w[n][k] in fp32 (of the STORED values, so the mean term cancels exactly), `bias` = b + W . beta."""

    def __init__(self, weight, gamma, beta, bias=None, eps=1e-5, fp8=False, pair=None):
        w32 = weight.float() * gamma.float()[None, :]
        b32 = weight.float() @ beta.float() + (bias.float() if bias is not None else 0.0)
        if pair is not None:                       # GEGLU: value rows | gate rows interleaved like the weight (packing.pair_rows)
            n = w32.shape[0] // 2
            w32, b32 = pair(w32[:n], w32[n:]), pair(b32[:n], b32[n:])
        w16 = w32.to(torch.float16).contiguous()
        if fp8:
            q, sc = quantize_fp8_rows(w16)
            self.w = Fp8Weight(q, sc)
            self.colsum = (q.float().sum(dim=1) * sc).contiguous()
        else:
            self.w = w16
            self.colsum = w16.float().sum(dim=1).contiguous()
        self.bias = b32.to(torch.float16).contiguous()
        self.eps = float(eps)


def ln_parts(M, N, K):
    """Partials per row the plain-epilogue, tile = 0 launch of (M, N, K) leaves in `ln_out` (0: it cannot)."""
    return L.load().iir_gemm_ln_parts(M, N, K)


GN_PARTIALS_MAX = 20 * 256          # slabs x channels-per-group one (image, group) of iir_groupnorm_from_partials may hold (csrc/norm.hip)


def gn_supported(M, N, K, conv=False):
    """Can the tile = 0 launch of (M, N, K) leave GroupNorm partials (`gn_out=`)?"""
    return bool(L.load().iir_gemm_gn_supported(M, N, K, int(conv)))


def _chk_gn_out(gn_out, M, N):
    """GroupNorm partials of a producing launch: fp32 (M / 64, N, 2), contiguous (`iir_gemm_desc.gn_stats_out`)."""
    if gn_out.dtype != torch.float32 or not gn_out.is_contiguous() or M % 64 or tuple(gn_out.shape) != (M // 64, N, 2):
        raise ValueError(f"gn_out: contiguous fp32 ({M} // 64, {N}, 2) with M % 64 == 0, got {tuple(gn_out.shape)} {gn_out.dtype}")


def gemm(a, w, out, bias=None, rowbias=None, rows_per_rb=1, res=None, epi=EPI_PLAIN, act=ACT_NONE, out_scale=1.0,
         tile=0, prefetch=None, splitk_ws=None, out_t=None, wscale=None, ln_out=None, ln_in=None, gn_out=None, xattn=None):
    """out = epi(a @ w.T).  a (M,K) view, w (N,K) contiguous, out (M,N) view ((M,N/2) for paired epilogues).
    out_t = (Ct, tr_from): output columns >= tr_from go, transposed, to Ct[n - tr_from, m]; `out` then is (M, tr_from).
    wscale (fp32 (N,)): `w` holds fp8-E4M3 bytes (torch.float8_e4m3fn / uint8) with that per-row scale (fp8 MFMA).
    ln_out (fp32 (parts, M, 2), parts = ln_parts(M, N, K)): also leave LayerNorm partials of the rows written.
    ln_in = (partials (parts, M, 2) fp32, colsum (N,) fp32, eps): `a` holds raw rows, `w` / `bias` are an `LnFold`'s.
    xattn = (kv, Tq) with epi = EPI_XATTN: `w` is a cross-attention q projection with `attn_q_factor()` folded in; `out` receives the
    text + IP cross-attention output of those queries (kv: two tuples as for `attention`, Tq: query rows per image)."""
    dt = a.dtype
    if isinstance(w, Fp8Weight):
        w, wscale = w.q, w.scale
    if wscale is not None:
        if w.dtype not in (torch.float8_e4m3fn, torch.uint8) or w.dim() != 2 or not w.is_contiguous() or not w.is_cuda:
            raise ValueError("fp8 weights: contiguous 2-D CUDA tensor of torch.float8_e4m3fn (or its bytes)")
        if wscale.dtype != torch.float32 or wscale.numel() != w.shape[0] or not wscale.is_contiguous():
            raise ValueError("wscale: contiguous fp32 (N,)")
        if dt != torch.float16:
            raise ValueError("fp8 weights take fp16 activations")
    if dt not in _DT:
        raise ValueError(f"a: fp16 or bf16 expected, got {dt}")
    c_f32 = out.dtype == torch.float32                 # fp32 output (plain epilogue only): the VAE's attention scores
    _chk2d(a, "a", dt); _chk2d(out, "out", torch.float32 if c_f32 else dt)
    if wscale is None:
        _chk2d(w, "w", dt)
    for t_, n_ in ((bias, "bias"), (rowbias, "rowbias"), (res, "res")):
        if t_ is not None and t_.dtype != dt:
            raise ValueError(f"{n_}: dtype {t_.dtype} does not match the operands ({dt})")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous():
        raise ValueError("w must be contiguous (N,K) with K matching a")
    n_out = (N if epi in (EPI_PLAIN, EPI_XATTN) else N // 2) if out_t is None else out_t[1]
    if out.shape != (M, n_out):
        raise ValueError(f"out shape {tuple(out.shape)} != {(M, n_out)}")
    d = L.GemmDesc()
    d.A, d.lda = a.data_ptr(), a.stride(0)
    d.W = w.data_ptr()
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.bias = _p(bias)
    if rowbias is not None:
        _chk2d(rowbias, "rowbias", dt)
        d.rowbias, d.ldrb, d.rows_per_rb = rowbias.data_ptr(), rowbias.stride(0), rows_per_rb
    if res is not None:
        _chk2d(res, "res", dt)
        d.res, d.ldr = res.data_ptr(), res.stride(0)
    d.epi, d.act, d.out_scale, d.tile = epi, act, out_scale, tile
    d.dtype, d.c_f32 = _DT[dt], int(c_f32)
    if wscale is not None:
        d.wscale = wscale.data_ptr()
    if ln_out is not None:
        if ln_out.dtype != torch.float32 or not ln_out.is_contiguous() or ln_out.dim() != 3 or ln_out.shape[1:] != (M, 2) \
                or ln_out.shape[0] != ln_parts(M, N, K) or tile != 0:
            raise ValueError("ln_out: contiguous fp32 (ln_parts(M, N, K), M, 2), tile = 0")
        d.ln_stats_out = ln_out.data_ptr()
    if gn_out is not None:
        _chk_gn_out(gn_out, M, N)
        d.gn_stats_out = gn_out.data_ptr()
    if ln_in is not None:
        part, colsum, eps = ln_in
        if part.dtype != torch.float32 or not part.is_contiguous() or part.dim() != 3 or part.shape[1:] != (M, 2) or K % part.shape[0]:
            raise ValueError("ln_in partials: contiguous fp32 (parts, M, 2) with parts dividing K")
        if colsum.dtype != torch.float32 or colsum.numel() != N or not colsum.is_contiguous():
            raise ValueError("ln_in colsum: contiguous fp32 (N,)")
        d.ln_stats_in, d.ln_colsum, d.ln_parts, d.ln_part_cols, d.ln_eps = part.data_ptr(), colsum.data_ptr(), part.shape[0], K // part.shape[0], eps
    kv_keep = None
    if (epi == EPI_XATTN) != (xattn is not None):
        raise ValueError("epi = EPI_XATTN goes with xattn = (kv, Tq)")
    if xattn is not None:
        kvs, tq = xattn
        if len(kvs) != 2:
            raise ValueError("xattn: two K / V^T segments (text, IP tokens)")
        kv_keep = (L.AttnKV * 2)()
        for i, (k, k_rows, vt, vbs, tkv) in enumerate(kvs):
            _chk2d(k, "k"); _chk2d(vt, "vt")
            kv_keep[i].K, kv_keep[i].ldk, kv_keep[i].k_batch_stride = k.data_ptr(), k.stride(0), k_rows * k.stride(0)
            kv_keep[i].Vt, kv_keep[i].ldvt, kv_keep[i].vt_batch_stride = vt.data_ptr(), vt.stride(0), vbs
            kv_keep[i].Tkv = tkv
        d.xattn_kv, d.xattn_tq = kv_keep, tq
    if xattn is not None:
        tile = 93
    elif tile == 0:       # the library decides; resolve the same choice here only to NAME the launch for the profiler
        tile = 4 if splitk_ws is not None and L.load().iir_gemm_uses_splitk(M, N, K, splitk_ws.numel()) else auto_tile(M, N, epi != EPI_PLAIN, K)
    if prefetch is not None:
        d.prefetch, d.prefetch_bytes = prefetch
    if splitk_ws is not None:
        d.splitk_ws, d.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel()
    if out_t is not None:
        ct, tr_from = out_t
        _chk2d(ct, "out_t", dt)
        if ct.shape[0] < N - tr_from or ct.shape[1] < M:
            raise ValueError("out_t must hold (N - tr_from, M)")
        d.Ct, d.ldct, d.tr_from = ct.data_ptr(), ct.stride(0), tr_from
    No = N // 2 if epi not in (EPI_PLAIN, EPI_XATTN) else N
    if xattn is not None:
        cls = "gemm_kernel<64x128,gemm+xattn>"
    elif PROFILER is not None and d.tile == 0 and L.load().iir_gemm_resolve_tile(C.byref(d)) == 91:
        cls = "gemm8_kernel<256x320,gemm>"             # the 8-wave kernel of csrc/gemm8.hip
    else:
        cls = "gemm_kernel<%s,%s>" % (_TILE_NAMES[tile % 10], "gemm" if wscale is None else "gemm-w8")
    xa_flops = 4.0 * M * N * sum(k[4] for k in xattn[0]) if xattn is not None else 0.0
    with _Timed(cls, 2.0 * M * N * K + xa_flops,
                2.0 * (M * K + N * K * (1.0 if wscale is None else 0.5) + M * No + (M * No if res is not None else 0))):
        L.check(L.load().iir_gemm_f16(C.byref(d), _stream()), "iir_gemm_f16")
    return out


def conv2d(x, w, out, ksize=3, stride=1, upsample=False, bias=None, rowbias=None, rows_per_rb=1, res=None,
           epi=EPI_PLAIN, act=ACT_NONE, out_scale=1.0, tile=0, y_img_rows=0, res_img_rows=0, pad_mode=0, prefetch=None,
           splitk_ws=None, gn_out=None):
    """x (R,H,W,Cin) NHWC view (pixel stride x.stride(2), image stride x.stride(0) free), w (Cout,k,k,Cin) contiguous,
    out (rows, Cout[/2]) 2-D view; image i's pixels start at row i*y_img_rows (0 = dense)."""
    R, H, Wd, Cin = x.shape
    dt = x.dtype
    if dt not in _DT or x.stride(3) != 1 or x.stride(1) != Wd * x.stride(2):
        raise ValueError("x must be an NHWC fp16 / bf16 view with dense rows (pixel and image strides are free)")
    _chk2d(out, "out", dt)
    for t_, n_ in ((w, "w"), (bias, "bias"), (rowbias, "rowbias"), (res, "res")):
        if t_ is not None and t_.dtype != dt:
            raise ValueError(f"{n_}: dtype {t_.dtype} does not match x ({dt})")
    Cout = w.shape[0]
    if tuple(w.shape[1:]) != (ksize, ksize, Cin) or not w.is_contiguous():
        raise ValueError("w must be contiguous (Cout,k,k,Cin)")
    d = L.ConvDesc()
    d.X, d.ldx = x.data_ptr(), x.stride(2)
    d.R, d.H, d.Wd, d.Cin = R, H, Wd, Cin
    d.Wt = w.data_ptr()
    d.Y, d.ldy = out.data_ptr(), out.stride(0)
    d.Cout, d.ksize, d.stride, d.upsample = Cout, ksize, stride, int(bool(upsample))
    d.bias = _p(bias)
    if rowbias is not None:
        _chk2d(rowbias, "rowbias", dt)
        d.rowbias, d.ldrb, d.rows_per_rb = rowbias.data_ptr(), rowbias.stride(0), rows_per_rb
    if res is not None:
        _chk2d(res, "res", dt)
        d.res, d.ldr = res.data_ptr(), res.stride(0)
    d.dtype = _DT[dt]
    Hi, Wi = (2 * H, 2 * Wd) if upsample else (H, Wd)
    pad2 = 1 if pad_mode == 1 else 2 * (ksize // 2)
    Mo = R * ((Hi + pad2 - ksize) // stride + 1) * ((Wi + pad2 - ksize) // stride + 1)
    d.epi, d.act, d.out_scale, d.tile = epi, act, out_scale, tile
    if gn_out is not None:
        _chk_gn_out(gn_out, Mo, Cout)
        d.gn_stats_out = gn_out.data_ptr()
    if tile == 0:       # as in gemm(): the library decides, this only names the launch
        Kc = ksize * ksize * Cin
        tile = 4 if splitk_ws is not None and L.load().iir_gemm_uses_splitk(Mo, Cout, Kc, splitk_ws.numel()) else auto_tile(Mo, Cout, epi != EPI_PLAIN, Kc)
    d.zero_page = zero_page(x.device).data_ptr()          # (all-zero bits are zero in fp16 and bf16 alike)
    if prefetch is not None:
        d.prefetch, d.prefetch_bytes = prefetch
    if splitk_ws is not None:
        d.splitk_ws, d.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel()
    d.x_img_stride, d.y_img_rows, d.res_img_rows, d.pad_mode = x.stride(0), y_img_rows, res_img_rows, pad_mode
    Co = Cout // 2 if epi != EPI_PLAIN else Cout
    with _Timed("gemm_kernel<%s,conv>" % _TILE_NAMES[tile % 10], 2.0 * Mo * Cout * ksize * ksize * Cin,
                2.0 * (R * H * Wd * Cin + Cout * ksize * ksize * Cin + Mo * Co + (Mo * Co if res is not None else 0))):
        L.check(L.load().iir_conv2d_nhwc_f16(C.byref(d), _stream()), "iir_conv2d_nhwc_f16")
    return out


ATTN_LOG2E = 1.4426950408889634


def attn_q_factor(scale=0.125):
    """The factor a caller folds into its q-projection weights to pass `q_prescaled=True` (float32, as the kernel forms it)."""
    import numpy as np
    return float(np.float32(scale) * np.float32(ATTN_LOG2E))


def attention(q, o, kv, batch, heads, Tq, scale=0.125, causal=False, q_prescaled=False):
    """q, o: 2-D views (batch*Tq, heads*64).  kv: list of 1-2 tuples (k2d, k_batch_rows, vt2d, vt_batch_stride, Tkv):
    k2d (batch*k_batch_rows, heads*64) view, vt2d (heads*64, cols) view with batch b starting at column b*vt_batch_stride."""
    _chk2d(q, "q")
    d = L.AttnDesc()
    if o.dtype in _FP8_DTYPES:          # the output feeds an all-fp8 `to_out` GEMM: E4M3 bytes (rounded to fp16 first)
        if o.dim() != 2 or o.stride(1) != 1 or o.stride(0) % 4 or not o.is_cuda:
            raise ValueError("attention: an fp8 `o` needs contiguous rows with a stride % 4 == 0")
        d.o_fp8 = 1
    else:
        _chk2d(o, "o")
    d.Q, d.ldq, d.q_batch_stride = q.data_ptr(), q.stride(0), Tq * q.stride(0)
    d.O, d.ldo, d.o_batch_stride = o.data_ptr(), o.stride(0), Tq * o.stride(0)
    d.batch, d.heads, d.Tq, d.nseg, d.scale = batch, heads, Tq, len(kv), scale
    d.causal = int(bool(causal))
    d.q_prescaled = int(bool(q_prescaled))
    for i, (k, k_rows, vt, vbs, tkv) in enumerate(kv):
        _chk2d(k, "k"); _chk2d(vt, "vt")
        d.kv[i].K, d.kv[i].ldk, d.kv[i].k_batch_stride = k.data_ptr(), k.stride(0), k_rows * k.stride(0)
        d.kv[i].Vt, d.kv[i].ldvt, d.kv[i].vt_batch_stride = vt.data_ptr(), vt.stride(0), vbs
        d.kv[i].Tkv = tkv
    with _Timed("attn_kernel", 4.0 * batch * heads * Tq * 64 * sum(k[4] for k in kv),
                2.0 * batch * heads * 64 * (2 * Tq + 2 * sum(k[4] for k in kv))):
        L.check(L.load().iir_attention_d64_f16(C.byref(d), _stream()), "iir_attention_d64_f16")
    return o


_gn_ws = {}


def _gn_workspace(device, R, groups):
    """fp32 partial-sum scratch, one per (device, stream): launches on different streams may overlap."""
    need = L.load().iir_groupnorm_workspace_bytes(R, groups)
    key = (device, _stream())
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(need // 4, dtype=torch.float32, device=device)
        _gn_ws[key] = ws
    return ws


def gn_workspace(device, R, groups=32):
    """Caller-owned fp32 scratch for `groupnorm` (one per concurrently running network)."""
    return torch.empty(L.load().iir_groupnorm_workspace_bytes(R, groups) // 4, dtype=torch.float32, device=device)


def groupnorm(x, out, R, HW, gamma, beta, eps, silu, groups=32, ws=None, partials=None):
    """x, out: 2-D views (R*HW, C).  `ws`: scratch from gn_workspace() (a shared per-stream one if omitted).
    `partials` (fp32 (R*HW / 64, C, 2)): the statistics the launch that PRODUCED x left (`gn_out=` of gemm / conv2d): no
    statistics pass over x."""
    dt = x.dtype
    _chk2d(x, "x", dt if dt in _DT else torch.float16); _chk2d(out, "out", dt)
    if gamma.dtype != dt or beta.dtype != dt:
        raise ValueError(f"gamma / beta must have the activations' dtype ({dt})")
    Cc = x.shape[1]
    if ws is None:
        ws = _gn_workspace(x.device, R, groups)
    if partials is not None:
        _chk_gn_out(partials, R * HW, Cc)
        if HW % 64 or (HW // 64) * (Cc // groups) > GN_PARTIALS_MAX:
            raise ValueError(f"groupnorm(partials=): images of whole 64-row slabs and at most {GN_PARTIALS_MAX} partials per (image, group); "
                             f"got HW={HW}, {Cc // groups} channels per group")
        L.check(L.load().iir_groupnorm_from_partials(partials.data_ptr(), Cc, x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0),
                                                     R, HW, Cc, groups, gamma.data_ptr(), beta.data_ptr(), eps, int(silu), ws.data_ptr(),
                                                     ws.numel() * 4, _DT[dt], _stream()), "iir_groupnorm_from_partials")
        return out
    L.check(L.load().iir_groupnorm_nhwc(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), R, HW, Cc, groups,
                                        gamma.data_ptr(), beta.data_ptr(), eps, int(silu), ws.data_ptr(),
                                        ws.numel() * 4, _DT[dt], _stream()), "iir_groupnorm_nhwc")
    return out


def layernorm(x, out, gamma=None, beta=None, eps=1e-5, shift=None, scale=None, rows_per_mod=1, transposed=False,
              tr_rows=1, tr_bstride=0):
    """`out` of dtype torch.float8_e4m3fn (or uint8): the normalised rows are stored as E4M3 bytes (rounded to fp16 first) -- the
    A operand of an all-fp8 GEMM (`gemm_fp8`)."""
    _chk2d(x, "x")
    if out.dtype in (torch.float8_e4m3fn, torch.uint8):
        if transposed or out.dim() != 2 or out.stride(1) != 1 or out.stride(0) % 8 or not out.is_cuda or out.shape != x.shape:
            raise ValueError("layernorm: an fp8 `out` needs x's shape, contiguous rows with a stride % 8 == 0, no transposition")
        transposed = 2
    else:
        _chk2d(out, "out")
    rows, Cc = x.shape
    ldmod = shift.stride(0) if shift is not None else 0
    L.check(L.load().iir_layernorm_f16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, Cc, _p(gamma),
                                       _p(beta), eps, _p(shift), _p(scale), ldmod, rows_per_mod, int(transposed), tr_rows,
                                       tr_bstride, _stream()), "iir_layernorm_f16")
    return out


def adaln_job_table(jobs, device):
    """jobs: list of (x, out, shift, scale, transposed) 2-D fp16 views -> device table of iir_adaln_job records.
    The caller keeps the views alive; the table holds raw addresses."""
    arr = (L.AdaLNJob * len(jobs))()
    for i, (x, out, shift, scale, tr) in enumerate(jobs):
        _chk2d(x, "x"); _chk2d(out, "out")
        arr[i] = L.AdaLNJob(x.data_ptr(), out.data_ptr(), shift.data_ptr(), scale.data_ptr(), x.stride(0), out.stride(0),
                            x.shape[1], int(tr))
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return raw.to(device)


def adaln_batch(table, njobs, rows, max_C, ldmod, rows_per_mod, tr_rows, tr_bstride, eps=1e-6):
    L.check(L.load().iir_adaln_batch_f16(table.data_ptr(), njobs, rows, max_C, eps, ldmod, rows_per_mod, tr_rows, tr_bstride,
                                         _stream()), "iir_adaln_batch_f16")


def sinusoid(vals, out, dim, col_off=0):
    """vals fp32 (rows, n_vals) device tensor; out 2-D fp16 view."""
    rows, n_vals = vals.shape
    L.check(L.load().iir_sinusoid_f16(vals.data_ptr(), n_vals, rows, dim, out.data_ptr(), out.stride(0), col_off,
                                      _stream()), "iir_sinusoid_f16")
    return out


def silu(x, out):
    L.check(L.load().iir_silu_f16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "iir_silu_f16")
    return out


def copy_add(src, dst, dst_off=0, add=None, add_scale=None, rows_per_scale=1):
    """dst[:, dst_off:dst_off+C] = src + add * add_scale[row // rows_per_scale]."""
    _chk2d(src, "src"); _chk2d(dst, "dst")
    M, Cc = src.shape
    L.check(L.load().iir_copy_add_f16(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), dst_off, M, Cc,
                                      _p(add), add.stride(0) if add is not None else 0, _p(add_scale), rows_per_scale,
                                      _stream()), "iir_copy_add_f16")
    return dst


def pack_latent(x, out, rep=1, scale=1.0):
    """x fp32 (B,C,H,W) contiguous -> out 2-D fp16 view (rep*B*H*W, ld>=C)."""
    B, Cc, H, Wd = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and out.dtype in _DT
    L.check(L.load().iir_pack_latent_t(x.data_ptr(), B, Cc, H * Wd, out.data_ptr(), out.stride(0), rep, scale, _DT[out.dtype],
                                       _stream()), "iir_pack_latent_t")
    return out


def unpack_latent(x2d, out):
    """x2d fp16 view (R*H*W, ld) -> out fp32 (R,C,H,W)."""
    R, Cc, H, Wd = out.shape
    assert out.dtype == torch.float32 and out.is_contiguous() and x2d.dtype in _DT
    L.check(L.load().iir_unpack_latent_t(x2d.data_ptr(), x2d.stride(0), R, Cc, H * Wd, out.data_ptr(), _DT[x2d.dtype], _stream()),
            "iir_unpack_latent_t")
    return out


def sched_step(eps2d, B, coef, x, prev, noise=None, cfg=True, x0_out=None, eps_out=None, eps_factor=None):
    _, Cc, H, Wd = x.shape
    L.check(L.load().iir_sched_step(eps2d.data_ptr(), eps2d.stride(0), B, Cc, H * Wd, int(cfg), coef.data_ptr(),
                                    x.data_ptr(), _p(noise), prev.data_ptr(), _p(x0_out), _p(eps_out), _p(eps_factor),
                                    _stream()), "iir_sched_step")
    return prev


def cfg_rescale_factor(eps2d, B, coef, x, guidance_rescale, factor):
    """factor (B,) fp32 device: the per-image multiplier `rescale_noise_cfg` applies to the guided eps."""
    _, Cc, H, Wd = x.shape
    L.check(L.load().iir_cfg_rescale_factor(eps2d.data_ptr(), eps2d.stride(0), B, Cc, H * Wd, coef.data_ptr(),
                                            float(guidance_rescale), factor.data_ptr(), _stream()), "iir_cfg_rescale_factor")
    return factor


def lcm_step(eps2d, B, rep, coef, x, out2d, out_nchw=None):
    _, Cc, H, Wd = x.shape
    L.check(L.load().iir_lcm_step(eps2d.data_ptr(), eps2d.stride(0), B, rep, Cc, H * Wd, coef.data_ptr(), x.data_ptr(),
                                  out2d.data_ptr(), out2d.stride(0), _p(out_nchw), _stream()), "iir_lcm_step")
    return out2d


def transpose(x, out, rows_pad):
    _chk2d(x, "x"); _chk2d(out, "out")
    rows, cols = x.shape
    L.check(L.load().iir_transpose_f16(x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), out.stride(0), rows_pad,
                                       _stream()), "iir_transpose_f16")
    return out


def sched_step_f32(eps, x, coef, prev, noise=None, x0_out=None):
    """fp32 contiguous tensors of equal shape; coef fp32 device (8,)."""
    L.check(L.load().iir_sched_step_f32(eps.data_ptr(), x.data_ptr(), _p(noise), coef.data_ptr(), x.numel(), prev.data_ptr(),
                                        _p(x0_out), _stream()), "iir_sched_step_f32")
    return prev


def axpby_f32(x, y, coef, out):
    L.check(L.load().iir_axpby_f32(x.data_ptr(), y.data_ptr(), coef.data_ptr(), x.numel(), out.data_ptr(), _stream()),
            "iir_axpby_f32")
    return out


def prefetch(ptr, nbytes, blocks=32):
    """Pull [ptr, ptr+nbytes) towards the Infinity Cache on the current stream."""
    L.check(L.load().iir_prefetch(ptr, nbytes, blocks, _stream()), "iir_prefetch")


def softmax_rows(x):
    """In-place softmax over the columns of a 2-D fp16 view (cols <= 16384)."""
    _chk2d(x, "x")
    L.check(L.load().iir_softmax_rows_f16(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], _stream()), "iir_softmax_rows_f16")
    return x


def softmax_rows_f32(s, p):
    """p = softmax over the columns of the fp32 scores s (cols <= 16384, % 4 == 0); p is fp16 or bf16."""
    if s.dtype != torch.float32 or s.dim() != 2 or s.stride(1) != 1 or p.dtype not in _DT or p.shape != s.shape:
        raise ValueError("softmax_rows_f32: s fp32 2-D, p fp16/bf16 of the same shape")
    L.check(L.load().iir_softmax_rows_f32(s.data_ptr(), s.stride(0), p.data_ptr(), p.stride(0), s.shape[0], s.shape[1], _DT[p.dtype],
                                          _stream()), "iir_softmax_rows_f32")
    return p


def blend_tiles(a, b, extent, vertical):
    """In-place seam blend of fp32 NCHW tile `b` with its upper (vertical) or left neighbour `a`."""
    assert a.dtype == b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    L.check(L.load().iir_blend_tiles_f32(a.data_ptr(), b.data_ptr(), a.shape[0] * a.shape[1], a.shape[2], a.shape[3], b.shape[2],
                                         b.shape[3], extent, int(vertical), _stream()), "iir_blend_tiles_f32")
    return b
