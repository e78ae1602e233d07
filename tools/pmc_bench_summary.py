"""Round 5: one JSON from the rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0` (tools/gpu_r5_final.sh; separate passes, kernel-trace only):
  <root>/p_fetch, <root>/p_write   FETCH_SIZE / WRITE_SIZE (KiB as rocprofv3 reports them; the consumer doubles FETCH_SIZE, the gfx950 correction)
  <root>/p_sq                      SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
for the bench's dominant kernel class (gemm_nt_dma_kernel<double> + gemm_nt_sk_kernel<double>: EVERY dispatch of the pass, none skipped, so that the
launch count is a multiple of the bench's launches_per_step — tests/test_bench_line.py), plus the same SQ ratios for the pure-MFMA reference kernel of
the same process (mfma_rate_f64_kernel) and for the leaf / in-panel kernels.   python tools/pmc_bench_summary.py <root>"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
GEMM = ("gemm_nt_dma_kernel<double", "gemm_nt_sk_kernel<double")


def load(tag):
    cc = glob.glob(f"{root}/{tag}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"{root}/{tag}/**/*kernel_trace.csv", recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch -> counter -> value (summed over its rows)
    for f in cc:
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].replace("void gpmi::", "").split("(")[0]
            d = per[nm][r["Dispatch_Id"]]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return per, dur


out = {}
for tag, ctr in (("p_fetch", "FETCH_SIZE"), ("p_write", "WRITE_SIZE")):
    per, dur = load(tag)
    vals = [c[ctr] for nm, ds in per.items() if nm.startswith(GEMM) for c in ds.values() if ctr in c]
    if vals:
        out[ctr] = {"avg": sum(vals) / len(vals), "n": len(vals), "unit": "KiB per launch (rocprofv3), every MFMA GEMM dispatch of the pass"}
per, dur = load("p_sq")


def sq(match):
    agg = collections.defaultdict(float)
    n = 0
    us = 0.0
    for nm, ds in per.items():
        if not match(nm):
            continue
        for did, c in ds.items():
            n += 1
            us += dur.get(did, 0.0)
            for k, v in c.items():
                agg[k] += v
    if not n or not agg.get("GRBM_GUI_ACTIVE"):
        return None
    wc = agg.get("SQ_WAVE_CYCLES", 0.0)
    return {"dispatches": n, "total_ms": us / 1e3, "clock_ghz": agg["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3) if us else None,  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            "mfma_busy_over_active": agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / agg["GRBM_GUI_ACTIVE"],
            "waves_parked": agg.get("SQ_WAIT_ANY", 0.0) / wc if wc else None, "waves_issue_stall": agg.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
            "waves_issuing": agg.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None, "waves_lds_stall": agg.get("SQ_WAIT_INST_LDS", 0.0) / wc if wc else None}


ref = sq(lambda nm: "mfma_rate_f64" in nm)
g = sq(lambda nm: nm.startswith(GEMM))
if g:
    if ref:
        g["mfma_busy_vs_pure_mfma_kernel"] = g["mfma_busy_over_active"] / ref["mfma_busy_over_active"]
    out["SQ"] = {"gemm": g, "pure_mfma_reference": ref, "leaf": sq(lambda nm: nm.startswith("panel64v2_kernel")), "in_panel_update": sq(lambda nm: nm.startswith("panel_updk_kernel")),
                 "note": "mfma_busy_over_active = SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (both summed over XCDs; meaningful as a RATIO to the pure-MFMA kernel of the same pass); "
                         "clock = GRBM_GUI_ACTIVE / 8 / kernel wall time (MI355X_MICROARCH.md: DVFS give-back); waves_* = fractions of SQ_WAVE_CYCLES"}
print(json.dumps(out, indent=1))
