"""Per-step kernel breakdown (name x grid) of the timed region of a `rocprofv3 --kernel-trace` run of bench.py (rocpd .db)."""
import sqlite3, collections, re, sys
db, S = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x, workgroup_x, grid_y, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'sched_step' in r[0]]
sel = rows[idx[-S - 1] + 1: idx[-1] + 1]
print('span ms/step', (sel[-1][5] - sel[0][4]) / 1e6 / S, 'kernels/step', len(sel) / S)
d = collections.defaultdict(list)
for r in sel:
    n = re.sub(r'\(.*', '', r[0].replace('(anonymous namespace)::', '').replace('void ', ''))
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)[:40]
    d[(n, r[1] // r[2], r[3])].append(r[5] - r[4])
print('sum kernel ms/step', sum(sum(v) for v in d.values()) / 1e6 / S)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    v.sort()
    print(f"{k[0]:42s} blocks={k[1]:6d} y={k[2]:3d} n/step={len(v)/S:6.1f} med={v[len(v)//2]/1e3:7.1f}us  ms/step={sum(v)/1e6/S:6.2f}")
