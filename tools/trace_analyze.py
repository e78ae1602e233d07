"""Summarise a rocprofv3 kernel_trace.csv of tools/trace_fit.py: the last fit only (kernels after the last long gap),
grouped by kernel and launch shape, plus how much of the span has 0 / 1 / ≥2 kernels in flight."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void gpmi::", "").replace("gpmi::", "")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short,
               (int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0)),
               r.get("Stream_Id", r.get("Queue_Id", "?"))))
ks.sort()
# split into fits: the assembly kernel (kmat with the largest grid) starts each fit
big = max(k[3][0] * max(k[3][1], 1) for k in ks if k[2].startswith("kmat_kernel"))
starts = [i for i, k in enumerate(ks) if k[2].startswith("kmat_kernel") and k[3][0] * max(k[3][1], 1) == big]
last = ks[starts[-1]:]
t0, t1 = last[0][0], max(k[1] for k in last)
print(f"last fit: {len(last)} kernels, span {(t1 - t0) / 1e6:.3f} ms")
grp = defaultdict(lambda: [0, 0])
for s, e, nm, g, q in last:
    key = (nm, g, q)
    grp[key][0] += 1
    grp[key][1] += e - s
byname = defaultdict(lambda: [0, 0])
for (nm, g, q), (c, d) in grp.items():
    byname[(nm, q)][0] += c
    byname[(nm, q)][1] += d
print("-- by kernel, stream")
for (nm, q), (c, d) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
    print(f"{d / 1e6:9.3f} ms {c:5d} x {d / c / 1e3:8.1f} us  {nm[:60]} [stream {q}]")
print("-- top launch shapes")
for (nm, g, q), (c, d) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{d / 1e6:9.3f} ms {c:5d} x {d / c / 1e3:8.1f} us  {nm[:44]} grid={g} [stream {q}]")
ev = []
for s, e, *_ in last:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
occ = defaultdict(int)
cur, prev = 0, t0
for t, d in ev:
    occ[min(cur, 2)] += t - prev
    prev = t
    cur += d
print("-- in flight: " + ", ".join(f"{k}{'+' if k == 2 else ''}: {v / 1e6:.3f} ms" for k, v in sorted(occ.items())))
# gaps: idle time between consecutive kernels when nothing is in flight, by the kernel that follows
