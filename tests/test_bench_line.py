"""CPU: the committed bench line of the round (profiles/r*/bench_c4.json, what `python bench.py` printed on an MI355X) against the contract the
driver parses — keys, types, the metric / workload BASELINE.json names, the two objects of the hot-path tier (roofline, cpu_baseline) — and against
its own arithmetic (value = N / step time, roofline.frac = achieved / peak, the kernel's launch average against the committed rocprofv3 table)."""
import csv
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
ROUNDS = ("r6", "r5", "r4", "r3", "r2", "r1")


def _latest():
    for r in ROUNDS:
        p = ROOT / "profiles" / r / "bench_c4.json"
        if p.exists():
            return r, json.loads(p.read_text().strip().splitlines()[-1])
    pytest.skip("no committed bench line")


def test_bench_line_has_the_contract_keys_and_types():
    _, d = _latest()
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                     ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], typ), (key, type(d.get(key)))
    assert d["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    base = json.loads((ROOT / "BASELINE.json").read_text())
    assert d["metric"] == base["metric"], (d["metric"], base["metric"])
    assert "65536" in d["config"]["workload"].replace(" ", "") or d["config"].get("n") == 65536


def test_bench_line_arithmetic_is_self_consistent():
    _, d = _latest()
    n = d["config"]["n"]
    assert d["value"] == pytest.approx(n / (d["ms_per_step"] * 1e-3), rel=1e-9)
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-9)
    assert rf["achieved"] == pytest.approx((n**3 / 3 + 3 * n**2) / (d["ms_per_step"] * 1e-3) / 1e12, rel=1e-6)  # F_pair / t_pair
    assert 0.5 < rf["frac"] < 1.0 and 0.5 < rf["kernel_frac"] < 1.0
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["unit"] == d["unit"] and cb["cores"] >= 1


def test_kernel_launch_average_agrees_with_the_committed_rocprof_table():
    """roofline.avg_launch_ms (HIP events inside bench.py) against rocprofv3 --kernel-trace --stats of the same command (profiles/r*/
    bench_c4_kernel_stats.csv): the two tile-GEMM kernels' total duration over their launch count."""
    r, d = _latest()
    stats = ROOT / "profiles" / r / "bench_c4_kernel_stats.csv"
    if not stats.exists():
        pytest.skip("no committed rocprofv3 table for that round")
    calls = tot = 0
    with open(stats) as fh:
        for row in csv.DictReader(fh):
            if "gemm_nt_dma_kernel" in row["Name"] or "gemm_nt_sk_kernel" in row["Name"]:
                calls += int(row["Calls"])
                tot += float(row["TotalDurationNs"])
    assert calls > 0
    avg_ms = tot / calls / 1e6
    assert d["roofline"]["avg_launch_ms"] == pytest.approx(avg_ms, rel=0.03), (d["roofline"]["avg_launch_ms"], avg_ms)


def test_bench_line_carries_the_parity_checks_and_the_other_configs():
    _, d = _latest()
    assert d.get("check_vs_oracle_digest") == "pass"
    assert d["check_logpdf_rel_vs_oracle_digest"] <= 1e-10 and d["check_alpha_rel_vs_oracle_digest"] <= 1e-8
    assert d["check_residual_max"] <= 1e-9
    oc = d["other_configs"]
    for name in ("C2", "C3", "C5"):
        assert oc[name]["steps"] >= 5 and oc[name]["statistic"] == "median"
    for name in ("mean_and_var_4096", "cov_1024", "sequential_update_8192", "value_and_gradient"):
        assert name in oc["next"], name
    # every timed row carries its own check (round 6); the sparse objective's gradient joined the line late in round 6
    for name in ("mean_and_var_4096", "cov_1024", "sequential_update_8192"):
        if "check" in oc["next"][name]:
            assert oc["next"][name]["check"]["pass"] is True, name
    for name, row in oc["next"]["value_and_gradient"].items():
        if "check" in row:
            assert row["check"]["pass"] is True, name
    if "gradient" in oc["C5"]:
        assert oc["C5"]["gradient"]["check"]["pass"] is True and oc["C5"]["gradient"]["check"]["rel"] <= 1e-4


def test_pmc_summary_counts_whole_passes_of_the_bench():
    """roofline.traffic / mfma_busy are replayed from profiles/r*/pmc_bench_summary.json (PMC cannot be sampled inside the timed run): the file must come
    from the SAME launch mix — its launch count is a whole multiple of the line's launches_per_step (round 5 on: every dispatch of the pass is counted)."""
    r, d = _latest()
    p = ROOT / "profiles" / r / "pmc_bench_summary.json"
    if not p.exists():
        pytest.skip("no committed PMC summary for that round")
    s = json.loads(p.read_text())
    lps = int(d["roofline"]["launches_per_step"])
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        n = int(s[ctr]["n"])
        if r in ("r1", "r2", "r3", "r4"):
            n += 1  # the summariser of those rounds dropped the first dispatch of the pass
        assert n >= lps and n % lps == 0, (ctr, n, lps)
    if "SQ" in s:
        assert int(s["SQ"]["gemm"]["dispatches"]) % lps == 0
        assert d["roofline"]["traffic_detail"]["source"].endswith(f"{r}/pmc_bench_summary.json")
