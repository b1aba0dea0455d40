"""Builds the code objects tools/probes/hazard_probe loads: variant V0 of shfl_probe.hip compiled to assembly, `s_nop 7; s_nop 7`
inserted at ONE class of places per object, assembled with clang + ld.lld into tools/probes/bin/hz_<name>.co.
    python tools/probes/make_hazard_variants.py"""
import os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-I", ROOT + "/include", "--cuda-device-only", "-S", ROOT + "/tools/probes/shfl_probe.hip",
                "-o", tmp + "/base.s"], check=True, stderr=subprocess.DEVNULL)
src = open(tmp + "/base.s").read().split("\n")
start = next(i for i, l in enumerate(src) if l.startswith("_Z3finILi0E")) + 1
end = next(i for i, l in enumerate(src) if l.startswith(".Lfunc_end0"))
NOP = ["\ts_nop 7", "\ts_nop 7"]
def variant(name, fn):
    out = src[:start]
    for l in src[start:end]:
        pre, post = fn(l.strip())
        out += pre + [l] + post
    out += src[end:]
    open(f"{tmp}/{name}.s", "w").write("\n".join(out))
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", f"{tmp}/{name}.s", "-o", f"{tmp}/{name}.o"], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", f"{tmp}/{name}.o", "-o", f"{ROOT}/tools/probes/bin/hz_{name}.co"], check=True)
    print("built", name)
os.makedirs(ROOT + "/tools/probes/bin", exist_ok=True)
variant("e0_base", lambda l: ([], []))
variant("e1_after_vcmp_e64", lambda l: ([], NOP if re.match(r"v_cmp\w*_e64 s\[", l) else []))                     # VALU writes SGPR -> SALU reads it
variant("e2_before_saveexec", lambda l: (NOP if l.startswith("s_and_saveexec") else [], []))
variant("e3_after_exec_write", lambda l: ([], NOP if l.startswith(("s_or_b64 exec", "s_and_saveexec", "s_and_b64 exec")) else []))   # SALU writes EXEC -> VMEM / VALU
variant("e4_before_global_load", lambda l: (NOP if l.startswith("global_load") else [], []))
variant("e5_after_global_load", lambda l: ([], NOP if l.startswith("global_load") else []))                      # VMEM issue -> VALU overwrites its address registers
variant("e6_after_v_mad_u64", lambda l: ([], NOP if l.startswith("v_mad_u64_u32") else []))                      # VALU writes SGPR carry -> ...
variant("e7_after_branch_targets", lambda l: ([], NOP if re.match(r"\.LBB0_\d+:", l) else []))
