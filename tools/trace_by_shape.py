"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid size): calls, total and average duration, share.
python tools/trace_by_shape.py <dir> [min_share_percent]"""
import csv, sys, glob, collections, re
d = sys.argv[1]; min_share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
agg = collections.defaultdict(lambda: [0, 0])
tot = 0
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"gemm_kernel<([^>]*)>", name)
        short = ("gemm<" + m.group(1).replace(" ", "") + ">") if m else re.sub(r"\(.*", "", name)[-50:]
        if "attn_kernel" in name: short = "attn_kernel2" if "kernel2" in name else "attn_kernel"
        if "gemm8_kernel" in name: short = "gemm8<256x%s>" % ("320" if "Li320E" in name or "<320" in name else "256")
        m2 = re.search(r"gemm_kernelI(DF16_|DF16b)?Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d)", name)
        if m2: short = f"gemm<{'bf16,' if m2.group(1) == 'DF16b' else ''}{m2.group(2)}x{m2.group(3)},st{m2.group(4)},{'conv' if m2.group(5) == '1' else 'gemm'},w{m2.group(6)}>"
        key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "")))
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[key][0] += 1; agg[key][1] += dur; tot += dur
print(f"total kernel time {tot / 1e6:.1f} ms")
for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if 100.0 * t / tot < min_share: continue
    print(f"{100.0 * t / tot:5.1f}%  {t / 1e6:8.2f} ms  calls {n:6d}  avg {t / n / 1e3:8.1f} us  grid {g:>8s}  {k}")
