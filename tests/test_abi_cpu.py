"""CPU: the C-ABI library builds, loads and exports every symbol include/instantir_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os

from instantir_amd import lib


def test_header_symbols_have_bindings():
    syms = lib.declared_symbols()
    assert len(syms) >= 17 and "iir_gemm_f16" in syms and "iir_attention_d64_f16" in syms
    assert sorted(syms) == sorted(lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), "build instantir_amd/csrc/build.sh first (python -c 'import __graft_entry__ as g; g.build()')"
    h = ctypes.CDLL(lib.LIB_PATH)
    for s in lib.declared_symbols():
        assert hasattr(h, s), s
    h.iir_abi_version.restype = ctypes.c_int
    assert h.iir_abi_version() == 1


def test_descriptor_layouts_match_header():
    """ctypes mirrors of the structs: natural alignment, sizes as a C compiler lays them out."""
    import shutil
    import subprocess
    import tempfile
    sizes = (ctypes.sizeof(lib.GemmDesc), ctypes.sizeof(lib.ConvDesc), ctypes.sizeof(lib.AttnKV), ctypes.sizeof(lib.AttnDesc),
             ctypes.sizeof(lib.AdaLNJob))
    assert all(s % 8 == 0 for s in sizes) and sizes[2] == 56
    if shutil.which("gcc"):      # ask the C compiler itself
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "sz.c")
            open(src, "w").write('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu", sizeof(iir_gemm_desc),'
                                 'sizeof(iir_conv_desc), sizeof(iir_attn_kv), sizeof(iir_attn_desc), sizeof(iir_adaln_job));}' % lib.HEADER_PATH)
            subprocess.run(["gcc", src, "-o", os.path.join(td, "sz")], check=True)
            out = subprocess.run([os.path.join(td, "sz")], capture_output=True, text=True, check=True).stdout
        assert tuple(int(x) for x in out.split()) == sizes


def test_invalid_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any HIP call, so it can be exercised on CPU."""
    h = lib.load()
    d = lib.GemmDesc()
    assert h.iir_gemm_f16(ctypes.byref(d), None) == -1            # null pointers
    d.A = d.W = d.C = 4096
    d.M, d.N, d.K = 8, 8, 48                                       # K not a multiple of 64
    assert h.iir_gemm_f16(ctypes.byref(d), None) == -1
    c = lib.ConvDesc()
    c.X = c.Wt = c.Y = c.zero_page = 4096
    c.ksize, c.Cin, c.stride = 5, 64, 1
    assert h.iir_conv2d_nhwc_f16(ctypes.byref(c), None) == -1
    a = lib.AttnDesc()
    a.Q = a.O = 4096
    a.nseg = 3
    assert h.iir_attention_d64_f16(ctypes.byref(a), None) == -1
    assert h.iir_silu_f16(4096, 4096, 7, None) == -1              # n % 8


def test_xattn_epilogue_arguments_are_validated_without_a_gpu():
    """IIR_EPI_XATTN (to_q + cross-attention in one launch): every precondition the header states is refused with -1 before
    any HIP call."""
    h = lib.load()

    def desc():
        d = lib.GemmDesc()
        d.A = d.W = d.C = 4096
        d.lda = d.ldc = 256
        d.M, d.N, d.K = 128, 256, 256
        d.epi = lib.EPI_XATTN
        kv = (lib.AttnKV * 2)()
        for i, t in enumerate((77, 64)):
            kv[i].K = kv[i].Vt = 4096
            kv[i].ldk, kv[i].k_batch_stride, kv[i].ldvt, kv[i].vt_batch_stride, kv[i].Tkv = 256, 256 * t, 160, 80, t
        d.xattn_kv, d.xattn_tq = kv, 64
        return d, kv

    d, kv = desc()
    d.xattn_kv = None
    assert h.iir_gemm_f16(ctypes.byref(d), None) == -1            # no K / V
    for field, bad in (("xattn_tq", 72), ("xattn_tq", 0), ("N", 192), ("M", 96), ("act", lib.ACT_SILU), ("res", 4096), ("dtype", 1),
                       ("tile", 5), ("c_f32", 1)):
        d, kv = desc()
        setattr(d, field, bad)
        assert h.iir_gemm_f16(ctypes.byref(d), None) == -1, field
    for seg, field, bad in ((0, "Tkv", 81), (1, "Tkv", 65), (0, "Tkv", 0), (1, "ldk", 100), (0, "vt_batch_stride", 77), (1, "K", 4100)):
        d, kv = desc()
        setattr(kv[seg], field, bad)
        assert h.iir_gemm_f16(ctypes.byref(d), None) == -1, (seg, field)
    d, kv = desc()
    assert h.iir_gemm_resolve_tile(ctypes.byref(d)) == 93         # the valid descriptor resolves to the head-aligned 64 x 128 tile
