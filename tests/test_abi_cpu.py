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
