import csv, sys, glob, collections
d = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "gemm_kernel" not in name and "attn_kernel" not in name:
            continue
        key = (r["Dispatch_Id"], name[:60], r.get("Grid_Size", ""))
        agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in agg.items():
    print(k[0], k[1], "grid", k[2], " ".join(f"{a}={b:.4g}" for a, b in v.items()))
