"""Per-kernel-class HBM traffic per launch from two rocprofv3 PMC passes over bench.py (FETCH_SIZE pass, WRITE_SIZE pass).

  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_pmc_traffic.json

Units and corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: both counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 B, so it is DOUBLED; WRITE_SIZE is
exact for 16 B/lane stores.  Both kernels' operand reads (LDS-DMA dwordx4) and stores (dwordx4) are of that kind."""
import csv, json, re, sys, collections

def cls_of(name):
    m = re.search(r"gemm_kernel<(?:[A-Za-z_0-9]+, )?(\d+), (\d+), (\d+), (true|false)", name)
    if m:
        return "gemm_kernel<%sx%s,%s>" % (m.group(1), m.group(2), "conv" if m.group(4) == "true" else "gemm")
    m = re.search(r"gemm_kernelI(?:DF16_|DF16b)?Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)          # mangled form
    if m:
        return "gemm_kernel<%sx%s,%s>" % (m.group(1), m.group(2), "conv" if m.group(4) == "1" else "gemm")
    if "gemm8_kernel" in name: return "gemm8_kernel<256x320,gemm>"
    if "attn_kernel" in name:
        return "attn_kernel"
    return None

def collect(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        c = cls_of(r["Kernel_Name"])
        if c:
            d[c][0] += 1
            d[c][1] += float(r["Counter_Value"])
    return d

f, w = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-vae`",
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE as is; KiB -> bytes", "classes": {}}
for c in sorted(set(f) | set(w)):
    nf, sf = f.get(c, [0, 0.0]); nw, sw = w.get(c, [0, 0.0])
    rd = 2.0 * sf * 1024 / max(nf, 1); wr = sw * 1024 / max(nw, 1)
    out["classes"][c] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                         "hbm_bytes_per_launch": round(rd + wr)}
print(json.dumps(out, indent=1))
