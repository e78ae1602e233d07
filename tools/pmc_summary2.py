"""Per-kernel averages of rocprofv3 --pmc passes laid out as <root>/pmc_<tag>_<COUNTER>/**/pmc_counter_collection.csv (+ the
kernel trace of the same pass for durations).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; the gfx950
correction (FETCH_SIZE x2 for 16-B/lane streaming reads, MI355X_MICROARCH.md §HBM) is applied by the consumer (bench.py)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(root + "/pmc_*/")):
    tag = os.path.basename(d.rstrip("/"))[4:]
    ccs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kts = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not ccs:
        continue
    kt = {}
    for f in kts:
        for r in csv.DictReader(open(f)):
            kt[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for f in ccs:
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].split("(")[0].replace("void gpmi::", "")
            vals[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] in kt:
                durs[nm].append(kt[r["Dispatch_Id"]])
    res = {}
    for nm, cs in vals.items():
        tot = sum(durs[nm])
        res[nm] = {"dispatches": max(len(v) for v in cs.values()), "total_us": round(tot, 1),
                   **{c: {"avg": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}}
    top = dict(sorted(res.items(), key=lambda kv: -kv[1]["total_us"])[:8])
    out[tag] = top
print(json.dumps(out, indent=1))
