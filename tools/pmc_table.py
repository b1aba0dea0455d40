"""Average the counters of a rocprofv3 --pmc run per (kernel, grid): python tools/pmc_table.py <dir> [substring]"""
import csv, sys, glob, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
agg = collections.OrderedDict()
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if sub not in name:
            continue
        key = (name.split("(")[0][-48:], r.get("Grid_Size", ""))
        a = agg.setdefault(key, collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1; c[1] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k[0], "grid", k[1], "launches", next(iter(v.values()))[0])
    for a, (n, s) in v.items():
        print(f"    {a:32s} {s / n:.6g}")
