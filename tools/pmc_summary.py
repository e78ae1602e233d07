"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc/p*/pmc_counter_collection.csv): per-kernel average counter
values per dispatch and the durations from the kernel trace of the same pass."""
import csv
import collections
import glob
import json
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_nt_sub"
out = {}
for d in sorted(glob.glob(root + "/p*/")):
    cc = list(csv.DictReader(open(d + "pmc_counter_collection.csv")))
    kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(d + "pmc_kernel_trace.csv"))}
    vals = collections.defaultdict(list)
    durs = {}
    for r in cc:
        if pat not in r["Kernel_Name"]:
            continue
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        t = kt.get(r["Dispatch_Id"])
        if t:
            durs[r["Dispatch_Id"]] = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
    dl = sorted(durs.values())
    for k, v in vals.items():
        # skip the first (warm-up) dispatch when there are several
        vv = v[1:] if len(v) > 1 else v
        out[k] = {"avg": sum(vv) / len(vv), "n": len(vv), "pass": d.split("/")[-2],
                  "dur_us_med": dl[len(dl) // 2] if dl else None}
print(json.dumps(out, indent=1))
