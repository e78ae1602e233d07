"""Per-kernel table from a pmc_summary2.py JSON holding SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_LDS: effective clock = GRBM_GUI_ACTIVE / wall time, MFMA-busy relative to the pure-MFMA reference kernel of the
same pass (mfma_rate_*), and where the waves' cycles go (parked at s_waitcnt / barriers, stalled at issue, issuing)."""
import json
import sys

d = json.load(open(sys.argv[1]))
for tag, ks in d.items():
    ref = None
    for name, c in ks.items():
        if "mfma_rate" in name and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["GRBM_GUI_ACTIVE"]["sum"] > 0:
            ref = c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / c["GRBM_GUI_ACTIVE"]["sum"]
    print(f"== {tag}  (pure-MFMA reference busy/active = {ref})")
    for name, c in ks.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        act, us = c["GRBM_GUI_ACTIVE"]["sum"], c["total_us"]
        ghz = act / (us * 1e3) if us > 0 else 0.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("sum", 0.0) / act if act else 0.0
        wc = c.get("SQ_WAVE_CYCLES", {}).get("sum", 0.0)
        fr = lambda k: (c.get(k, {}).get("sum", 0.0) / wc) if wc else 0.0
        print(f"{name[:46]:46s} n={c['dispatches']:5d} {us / 1e3:9.2f} ms  clock {ghz:5.2f} GHz  mfma_busy/active {busy:7.3f}"
              + (f" = {busy / ref:5.3f} of ref" if ref else "")
              + f"  waves: parked {fr('SQ_WAIT_ANY'):5.3f} issue-stall {fr('SQ_WAIT_INST_ANY'):5.3f} (lds {fr('SQ_WAIT_INST_LDS'):5.3f}) issuing {fr('SQ_ACTIVE_INST_ANY'):5.3f}")
