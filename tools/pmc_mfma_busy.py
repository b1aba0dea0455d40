"""MFMA-busy fraction per kernel class from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)
over the bench: busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
sums the 8 XCDs; MI355X_MICROARCH.md, DVFS section).  python tools/pmc_mfma_busy.py <counter_collection.csv>"""
import csv, json, re, sys, collections
def cls_of(name):
    m = re.search(r"gemm_kernel<(?:[A-Za-z_0-9]+, )?(\d+), (\d+), (\d+), (true|false)", name)
    if m: return "gemm_kernel<%sx%s,%s>" % (m.group(1), m.group(2), "conv" if m.group(4) == "true" else "gemm")
    m = re.search(r"gemm_kernelI(?:DF16_|DF16b)?Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)
    if m: return "gemm_kernel<%sx%s,%s>" % (m.group(1), m.group(2), "conv" if m.group(4) == "1" else "gemm")
    if "gemm8_kernel" in name: return "gemm8_kernel<256x320,gemm>"
    if "attn_kernel" in name: return "attn_kernel"
    return None
rows = collections.defaultdict(dict)
names = {}
for r in csv.DictReader(open(sys.argv[1])):
    c = cls_of(r["Kernel_Name"])
    if c:
        rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); names[r["Dispatch_Id"]] = c
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for d, v in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        a = agg[names[d]]; a[0] += 1; a[1] += v["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += v["GRBM_GUI_ACTIVE"] / 8.0
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-vae`",
       "definition": "mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * sum GRBM_GUI_ACTIVE / 8)", "classes": {}}
for c, (n, busy, cyc) in sorted(agg.items()):
    out["classes"][c] = {"launches": n, "mfma_busy_frac": round(busy / (1024.0 * cyc), 4), "avg_kernel_cycles": round(cyc / n)}
print(json.dumps(out, indent=1))
