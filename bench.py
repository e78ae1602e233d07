#!/usr/bin/env python
"""bench.py — logpdf+posterior throughput (points/s, fp64) of the MI355X-native exact-GP engine.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one (logpdf, posterior-fit) pair on BASELINE config C4: GP(SqExponentialKernel()) on
65 536 3-D points, σ² = 0.01, fp64 — one Gram assembly, one Cholesky, logdet, the
forward/backward solves, logpdf scalar and α back on the host.  Inputs are synthetic (PCG64 seed 4); the
C ABI takes HOST x, y, so every timed step uploads them (2 MB) and downloads α (0.5 MB) inside the timed region: `value` is
already the PCIe-inclusive rate (≈ 0.1 ms of 1 370); the N×N matrix is assembled in HBM and never leaves it.
N > 1: the N×N matrix is partitioned 2D block-cyclically over the N devices INSIDE the library (gp_ctx_create_multi,
csrc/multi.hip: one internal host thread per device, RCCL grouped send/recv or peer copies over xGMI) — the caller is one
process, as the reference's caller is (one Julia process calling posterior(fx, y)).  `python bench.py --gpus N` therefore uses
N GPUs by itself; under the launcher (torch.distributed.run, N ranks) rank 0 drives the N devices and the other ranks only
take part in the barriers.  Same total work for every N ("strong" scaling).  Rank 0 prints ONE JSON line.  It exits non-zero
if fewer than N GPUs are visible — never a silent 1-GPU run.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet: 256 CU × 4 SIMD × 32 FLOP/clk × 2.4 GHz


_T0 = time.perf_counter()


def note(msg: str) -> None:
    """progress on stderr (the JSON line on stdout stays the only stdout output)"""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def lib_sha16() -> str:
    """sha256 (first 16 hex digits) of the libgpmi355.so this process loaded: which build a recorded line came from"""
    import hashlib

    p = ROOT / "abstractgps.jl_amd" / "csrc" / "libgpmi355.so"
    return hashlib.sha256(p.read_bytes()).hexdigest()[:16] if p.exists() else ""


def f_pair(n: int) -> float:
    """Algorithmic flops of one pair: N³/3 + 3N²."""
    return n**3 / 3.0 + 3.0 * n**2


def synth_inputs(n: int, d: int, seed: int):
    """Deterministic synthetic workload: X ~ N(0, I_d) from PCG64(seed), y_i = sin(Σ_d x_id) + 0.1 ε_i.
    (Same recipe as oracle.gp_oracle.synth_inputs; restated here so that the timed path does not touch oracle/.)"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    eps = rng.standard_normal(n)
    y = np.sin(X.sum(axis=1)) + 0.1 * eps
    return (X[:, 0].copy() if d == 1 else X), y


def se_rows(xi: np.ndarray, x: np.ndarray) -> np.ndarray:
    """rows of the unit SE Gram matrix, exp(−‖xi − xj‖²/2), for the post-run residual check (host, NumPy)."""
    d2 = ((xi[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d2)


def gram_threads_default() -> int:
    return min(32, os.cpu_count() or 1)


def host_mem_available_gb() -> float:
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return float(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(n_full: int, d: int, full: bool = True):
    """Oracle (NumPy/SciPy -> OpenBLAS LAPACK, the routines Julia's cholesky reaches) timed on this box's host cores.
    `value` is MEASURED: the oracle's in-place fused pair (one Fortran-ordered N×N, dpotrf('U') in place — SURVEY.md §8(d): "C4 in full when
    the host has >= 40 GB") run once at n_full on the same inputs the GPU was timed on (≈ 2–3 minutes at C4), its result checked against the
    engine's.  `samples` are the same pair at N = 8 192 / 16 384 / 32 768 (after a discarded warm-up call that starts the thread pools), Gram /
    potrf / solves timed separately; `extrapolated` is what they predict for n_full (potrf cubic, Gram and solves quadratic, each through
    its largest sample) — reported beside the measurement, and used as `value` only when the host cannot hold the N×N matrix (the sample
    string says which).  `two_factorisations` is the pair as the reference executes it (logpdf and posterior each rebuild and refactor the
    Gram matrix): the measured pair plus one more Gram + potrf."""
    from oracle import gp_oracle as o

    pools, cores = [], os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info

        pools = [{k: p.get(k) for k in ("user_api", "internal_api", "version", "num_threads", "threading_layer")}
                 for p in threadpool_info()]
        blas = [p.get("num_threads") or 1 for p in pools if p.get("user_api") == "blas"]  # dpotrf / dtrsv run on the BLAS pool
        cores = max(blas + [gram_threads_default()]) if blas else max([p.get("num_threads") or 1 for p in pools] + [1])
    except Exception:
        pass
    gram_threads = gram_threads_default()
    x, y = synth_inputs(n_full, d, 4)
    f = o.GP(o.Kernel(o.SE))
    sizes = [nn for nn in (8192, 16384, 32768) if nn <= n_full]
    o.logpdf_and_posterior_inplace(o.FiniteGP(f, x[:sizes[0]], 0.01), y[:sizes[0]], threads=gram_threads)   # warm-up, discarded
    rows, ts = [], []
    for nn in sizes:
        tm = {}
        t0 = time.perf_counter()
        o.logpdf_and_posterior_inplace(o.FiniteGP(f, x[:nn], 0.01), y[:nn], threads=gram_threads, timings=tm)
        ts.append(time.perf_counter() - t0)
        rows.append({"n": nn, "pair_s": ts[-1], **{k: round(v, 4) for k, v in tm.items()},
                     "potrf_gflops": nn**3 / 3 / tm["potrf_s"] / 1e9})
    last = rows[-1]
    r = n_full / float(last["n"])
    ex = {"gram_s": last["gram_s"] * r**2, "potrf_s": last["potrf_s"] * r**3, "solves_s": last["solves_s"] * r**2}
    t_ex = sum(ex.values())
    out = {"unit": "points/s", "cores": int(cores), "kind": "port", "samples": rows, "gram_threads": gram_threads, "threadpools": pools,
           "extrapolated": {"pair_s": t_ex, "phases_s": ex, "points_per_s": n_full / t_ex,
                            "how": f"from the N={last['n']} sample: potrf x (N/n)^3, Gram and solves x (N/n)^2"}}
    need_gb = 8.0 * n_full * n_full / 1e9 + 6.0
    avail = host_mem_available_gb()
    if full and avail >= need_gb:
        note(f"cpu baseline: full N={n_full} oracle run ({avail:.0f} GB of host memory available, {need_gb:.0f} needed)")
        tm = {}
        t0 = time.perf_counter()
        lp, alpha, _ = o.logpdf_and_posterior_inplace(o.FiniteGP(f, x, 0.01), y, threads=gram_threads, timings=tm)
        t_full = time.perf_counter() - t0
        out.update({"value": n_full / t_full, "pair_s": t_full, "phases_s": {k: round(v, 3) for k, v in tm.items()},
                    "potrf_gflops": n_full**3 / 3 / tm["potrf_s"] / 1e9, "logpdf": float(lp), "_alpha": alpha,
                    "sample": (f"full run: the oracle's in-place fused logpdf+posterior pair (one Gram + one dpotrf) at N={n_full}, the bench's own "
                               f"inputs, measured once on this box's host: {t_full:.1f} s (Gram {tm['gram_s']:.1f} / potrf {tm['potrf_s']:.1f} / "
                               f"solves {tm['solves_s']:.1f})")})
        t_twice = t_full + tm["gram_s"] + tm["potrf_s"]
    else:
        out.update({"value": n_full / t_ex,
                    "sample": (f"EXTRAPOLATED (host has {avail:.0f} GB available, the full run needs {need_gb:.0f}): in-place fused pair of the oracle at "
                               f"N={sizes} ({', '.join(f'{t:.2f}s' for t in ts)}) -> {t_ex:.0f} s at N={n_full}")})
        t_twice = t_ex + ex["gram_s"] + ex["potrf_s"]
    out["two_factorisations"] = {"value": n_full / t_twice, "unit": "points/s",
                                 "note": "the reference's own logpdf(fx,y) + posterior(fx,y) assemble and factor K twice"}
    return out


def pmc_traffic(n: int) -> dict:
    """HBM/fabric bytes per launch of the dominant kernel, and its MFMA-pipe occupancy, from the committed rocprofv3 --pmc passes over THIS command
    (tools/gpu_final.sh -> profiles/r<round>/pmc_bench_summary.json, written by tools/pmc_bench_summary.py; the newest round's file is taken; FETCH_SIZE and WRITE_SIZE are reported
    in KiB and FETCH_SIZE is doubled, the gfx950 correction for 16-B/lane streaming reads of MI355X_MICROARCH.md §HBM).  PMC cannot be sampled from inside
    the timed run, so these are REPLAYED from the file (`source` says which) and null when it is absent or for another N; tests/test_bench_line.py ties the
    file's launch count to `launches_per_step`."""
    path = next((q for q in (ROOT / "profiles" / r / "pmc_bench_summary.json" for r in ("r6", "r5", "r4", "r3", "r2", "r1")) if q.exists()), ROOT / "nonexistent")
    if n != 65536 or not path.exists():
        return {"traffic": None}
    s = json.loads(path.read_text())
    rd = 2.0 * s["FETCH_SIZE"]["avg"] * 1024.0
    wr = s["WRITE_SIZE"]["avg"] * 1024.0
    out = {"traffic": rd + wr, "traffic_detail": {"unit": "bytes per launch (average over the bench's MFMA GEMM launches)",
                                                  "read": rd, "write": wr, "source": str(path.relative_to(ROOT))}}
    g = (s.get("SQ") or {}).get("gemm")
    if g:  # SQ counters of the same launches: MFMA pipe busy relative to the pure-MFMA kernel of the same pass, effective clock, where the waves' cycles go
        out["mfma_busy"] = g.get("mfma_busy_vs_pure_mfma_kernel")
        out["mfma_busy_detail"] = {"definition": "(SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the MFMA GEMM launches) / (the same ratio of gp_bench_mfma_f64's kernel in the same rocprofv3 pass)",
                                   "clock_ghz": g.get("clock_ghz"), "waves_parked": g.get("waves_parked"), "waves_issue_stall": g.get("waves_issue_stall"),
                                   "waves_issuing": g.get("waves_issuing"), "source": str(path.relative_to(ROOT))}
    return out


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak


def other_configs(agp, ctx, steps: int = 5) -> dict:
    """The other single-GPU BASELINE configs, driver-observed: C2 (N = 16 384, D = 3, SE), C3 (N = 32 768, D = 8,
    Matern32 ∘ ScaleTransform(0.5)) — one (logpdf, posterior) pair per step, fraction of the fp64 MFMA peak with F_pair = N³/3 + 3N² —
    and C5 (VFE, N = 262 144, M = 4 096, fp32: posterior + ELBO per step, F = 2NM² + 2M³/3 against the fp32 MFMA peak).
    Untimed with respect to the headline: runs after the C4 loop; one warm-up step each, `steps` (>= 5) timed steps, wall clock per step,
    the MEDIAN is reported (min alongside)."""
    out = {}
    last_times = []

    def timed(fn):
        fn()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        last_times[:] = ts
        return float(np.median(ts))

    for name, n, d, seed, kern, desc in (
            ("C2", 16384, 3, 2, agp.SqExponentialKernel(), "GP(SqExponentialKernel()) on 16384 3-D points, sigma2=0.01, fp64"),
            ("C3", 32768, 8, 3, agp.Matern32Kernel() @ agp.ScaleTransform(0.5), "GP(Matern32Kernel() ∘ ScaleTransform(0.5)) on 32768 8-D points, sigma2=0.01, fp64")):
        x, y = synth_inputs(n, d, seed)
        fx = agp.GP(kern, ctx=ctx)(agp.RowVecs(x), 0.01)

        def pair():
            agp.posterior(fx, y).data.C.free()

        dt = timed(pair)
        tf = f_pair(n) / dt / 1e12
        out[name] = {"workload": desc, "ms_per_step": dt * 1e3, "ms_min": min(last_times) * 1e3, "steps": steps, "statistic": "median", "points_per_s": n / dt,
                     "tflops": tf, "frac": tf / FP64_MFMA_PEAK_TFLOPS, "peak": FP64_MFMA_PEAK_TFLOPS}
    rng = np.random.default_rng(5)
    n, m, d = 262144, 4096, 3
    X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
    y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
    z = X[rng.permutation(n)[:m]].copy()
    f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
    fx = f(agp.RowVecs(X), np.float32(0.1))
    approx = agp.VFE(f(agp.RowVecs(z), 1e-4))
    obj = []

    def fit():
        p = agp.posterior(approx, fx, y)
        obj.append(float(p.objective))
        del p

    dt = timed(fit)
    flops = 2.0 * n * m * m + 2.0 * m**3 / 3
    out["C5"] = {"workload": "VFE posterior + ELBO, N=262144, M=4096 pseudo-points, D=3, sigma2=0.1, jitter 1e-4, fp32", "ms_per_step": dt * 1e3,
                 "ms_min": min(last_times) * 1e3, "steps": steps, "statistic": "median",
                 "points_per_s": n / dt, "tflops": flops / dt / 1e12, "frac_fp32": flops / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, "peak": FP32_MFMA_PEAK_TFLOPS,
                 "elbo": obj[-1]}
    # value + gradient of the ELBO (gp_vfe_grad: what a caller maximising the ELBO over kernel parameters / pseudo-points does next — the reference's
    # examples/0-intro-1d/script.jl:385-394): the backward pass on the resident fp32 posterior (the pass itself is always fp64), median of 3 after a warm-up
    post = agp.posterior(approx, fx, y)
    post.objective_grad()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        g32 = post.objective_grad()
        ts.append(time.perf_counter() - t0)
    gdt = float(np.median(ts))
    ph = ctx.timings()
    gflops = 2.0 * n * m * m + 5.0 * m**3   # one CH×M×M product per chunk; M×M side: two triangular inverses, A⁻¹, two conjugations
    del post
    # check: the gradient of an fp64 posterior of the same inputs along (variance, scale, noise, z) against a central difference of two fp64 fits, and
    # the fp32 posterior's hyper-parameter gradients against it
    X64, y64, z64 = X.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
    dirs, dZ, h = np.array([0.3, -0.2, 0.05]), np.random.default_rng(6).standard_normal(z64.shape), 1e-5

    def obj64(var, sc, s2, zc, grad=False):
        f64 = agp.GP(var * agp.SqExponentialKernel() @ agp.ScaleTransform(sc), ctx=ctx)
        a64, fx64 = agp.VFE(f64(agp.RowVecs(zc), 1e-4)), f64(agp.RowVecs(X64), s2)
        return agp.elbo_and_grad(a64, fx64, y64) if grad else float(agp.approx_log_evidence(a64, fx64, y64))

    t0 = time.perf_counter()
    _, g64 = obj64(1.0, 1.0, 0.1, z64, grad=True)
    t64 = time.perf_counter() - t0
    fd = (obj64(1 + h * dirs[0], 1 + h * dirs[1], 0.1 + h * dirs[2], z64 + h * dZ) - obj64(1 - h * dirs[0], 1 - h * dirs[1], 0.1 - h * dirs[2], z64 - h * dZ)) / (2 * h)
    an = g64["variance"] * dirs[0] + g64["scale"] * dirs[1] + g64["noise"] * dirs[2] + float(np.sum(g64["z"] * dZ))
    r32 = max(abs(g32["variance"] - g64["variance"]) / abs(g64["variance"]), abs(g32["noise"] - g64["noise"]) / abs(g64["noise"]))
    out["C5"]["gradient"] = {
        "what": "d ELBO / d(variance, noise, y) of the resident fp32 posterior (gp_vfe_grad; + scale, pseudo-inputs on fp64 posteriors): M×M side, one fp64 MFMA product per chunk, fused reductions",
        "ms": gdt * 1e3, "ms_all": [t * 1e3 for t in ts], "phases_ms": {"mxm_side": ph["assemble_ms"], "streamed_pass": ph["potrf_ms"], "kzz_term": ph["solve_ms"]},
        "flops": gflops, "tflops": gflops / gdt / 1e12, "frac": gflops / gdt / 1e12 / FP64_MFMA_PEAK_TFLOPS, "peak": FP64_MFMA_PEAK_TFLOPS,
        "fp64_fit_plus_gradient_ms": t64 * 1e3,
        "check": {"directional_derivative_fp64": an, "central_difference_of_fp64_fits": fd, "rel": abs(an - fd) / abs(fd), "tol": 1e-4,
                  "fp32_vs_fp64_posterior_variance_noise_rel": r32, "tol_fp32": 3e-2, "pass": bool(abs(an - fd) <= 1e-4 * abs(fd) and r32 <= 3e-2)}}
    ctx.trim()
    return out


def cached_cpu_baseline(n_full: int):
    """The oracle's FULL C4 run on an MI355X box's host cores as recorded by tools/fullsize_parity.py (committed under profiles/):
    what a multi-GPU line carries as its CPU baseline (the bounded live sample is timed by the N = 1 run only)."""
    for rnd in ("r3", "r2"):
        rec_path = ROOT / "profiles" / rnd / "fullsize_parity.jsonl"
        if not rec_path.exists():
            continue
        for line in rec_path.read_text().splitlines():
            r = json.loads(line)
            if r.get("config") == "C4" and r.get("n") == n_full and "oracle_pair_s" in r:
                return {"value": r["oracle_points_per_s_fused"], "unit": "points/s", "cores": (r.get("host") or {}).get("cpu_count"), "kind": "port",
                        "sample": f"cached: the oracle's full in-place fused pair at N={n_full} on an MI355X box's host ({r['oracle_pair_s']:.1f} s), "
                                  f"profiles/{rnd}/fullsize_parity.jsonl; the live bounded sample is timed by the --gpus 1 run",
                        "two_factorisations": {"value": r["oracle_points_per_s_two_factorisations"], "unit": "points/s"}}
    return None


HBM_SPEC_TBPS = 8.0      # MI355X_MICROARCH.md: HBM3E spec
HBM_COPY_TBPS = 6.29     # MI355X_MICROARCH.md: measured float4 copy


def digest_check(n: int, logpdf_val: float, alpha: np.ndarray) -> dict:
    """The timed engine's result against the committed digest of the ORACLE's C4 run (tests/golden/digests/c4_oracle_digest.npz, written by
    tests/golden/make_c4_digest.py / the GPU suite's C4 value test from the oracle on an MI355X box's host): logpdf, and α through its
    norm, every 64th entry and eight seeded Gaussian projections.  Tolerances of SURVEY.md §8(c): 1e-10 / 1e-8."""
    path = ROOT / "tests" / "golden" / "digests" / "c4_oracle_digest.npz"
    if not path.exists():
        return {"check_vs_oracle_digest": "digest file absent"}
    dig = np.load(path)
    if int(dig["n"]) != n:
        return {"check_vs_oracle_digest": f"digest is for N={int(dig['n'])}"}
    a = np.asarray(alpha, dtype=np.float64)
    nrm = float(np.linalg.norm(a))
    scale = float(dig["alpha_norm"]) / np.sqrt(n)
    proj = np.array([float(np.random.default_rng(20260926 + i).standard_normal(n) @ a) / np.sqrt(n) for i in range(8)])
    e = max(abs(nrm - float(dig["alpha_norm"])) / float(dig["alpha_norm"]),
            float(np.linalg.norm(a[::64] - dig["alpha_sub"]) / np.linalg.norm(dig["alpha_sub"])),
            float(np.max(np.abs(proj - dig["alpha_proj"])) / scale))
    lp_rel = abs(logpdf_val - float(dig["logpdf"])) / abs(float(dig["logpdf"]))
    return {"check_logpdf_rel_vs_oracle_digest": lp_rel, "check_alpha_rel_vs_oracle_digest": e,
            "check_vs_oracle_digest": "pass" if (lp_rel <= 1e-10 and e <= 1e-8) else "FAIL"}


def next_rows(agp, ctx, post, x, y, n: int, sigma2: float) -> dict:
    """SURVEY.md §8(f) on the engine of this run, at the bench size, each with its roofline fraction (fp64 MFMA 78.6 TF/s; algorithmic flops):
    predictive marginals at 4 096 points (TRSM against the resident factor: N²·N* flops), the full 1 024² predictive covariance (N²·N* + N·N*²),
    sequential conditioning on 8 192 new observations (N²·n2 + N·n2² + n2³/3), value + gradient of logpdf (N³/3 for the factor + 2N³/3 for C⁻¹
    by triangular inverse and LᵀL product).  One warm-up where the call is cheap, wall clock."""
    out = {}
    rng = np.random.default_rng(11)
    pk = FP64_MFMA_PEAK_TFLOPS * 1e12

    def rec(name, dt, flops, note):
        out[name] = {"ms": dt * 1e3, "flops": flops, "tflops": flops / dt / 1e12, "frac": flops / dt / pk, "what": note}

    def med3(fn, reps=3):
        """same-size warm-up call, then the MEDIAN of `reps` timed calls (round 4 timed ONE call after a 256-point warm-up: the timed call paid
        the 2.1 GB workspace allocation, and DESIGN quoted the favourable run)"""
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts

    keep = {}
    xs = rng.standard_normal((4096, x.shape[1]))
    dt, ts = med3(lambda: keep.__setitem__("mv", post.mean_and_var(agp.RowVecs(xs))))
    rec("mean_and_var_4096", dt, float(n) * n * 4096, "marginals at 4096 test points: K_*x, TRSM with the resident factor, column sums")
    out["mean_and_var_4096"]["ms_all"] = [t * 1e3 for t in ts]
    dt, ts = med3(lambda: keep.__setitem__("cov", post.cov(agp.RowVecs(xs[:1024]))))
    rec("cov_1024", dt, float(n) * n * 1024 + float(n) * 1024 * 1024, "full 1024x1024 predictive covariance: TRSM + SYRK")
    out["cov_1024"]["ms_all"] = [t * 1e3 for t in ts]
    # checks on the TIMED results (untimed): the two rows reach the same numbers through different kernels (column sums of squares of the
    # solve vs the MFMA SYRK), the mean of 16 points is recomputed on the host from α, variances lie in [0, k(x, x)]
    m4, v4 = keep["mv"]
    c1 = keep["cov"]
    alpha = np.asarray(post.data.alpha, dtype=np.float64)
    out["mean_and_var_4096"]["check"] = {
        "mean_vs_host_K_alpha_max_abs_16pts": float(np.max(np.abs(se_rows(xs[:16], x) @ alpha - m4[:16]))), "tol_mean": 1e-8,
        "var_min": float(v4.min()), "var_max": float(v4.max()), "var_in_0_1": bool(v4.min() >= -1e-9 and v4.max() <= 1 + 1e-9)}
    out["cov_1024"]["check"] = {"diag_vs_mean_and_var_max_abs": float(np.max(np.abs(np.diag(c1) - v4[:1024]))), "tol": 1e-9,
                                "asymmetry_max_abs": float(np.max(np.abs(c1 - c1.T)))}
    out["mean_and_var_4096"]["check"]["pass"] = bool(out["mean_and_var_4096"]["check"]["mean_vs_host_K_alpha_max_abs_16pts"] <= 1e-8
                                                      and out["mean_and_var_4096"]["check"]["var_in_0_1"])
    out["cov_1024"]["check"]["pass"] = bool(out["cov_1024"]["check"]["diag_vs_mean_and_var_max_abs"] <= 1e-9
                                            and out["cov_1024"]["check"]["asymmetry_max_abs"] <= 1e-12)
    n2 = 8192
    x2 = rng.standard_normal((n2, x.shape[1]))
    y2 = np.sin(x2.sum(1)) + 0.1 * rng.standard_normal(n2)

    def upd():
        old = keep.pop("p2", None)
        if old is not None:
            old.data.C.free()
        keep["p2"] = agp.posterior(post(agp.RowVecs(x2), sigma2), y2)

    # (the warm-up call allocates the (N + n2)² block: hipMalloc of 43 GB ≈ 1.3 s)
    dt, ts = med3(upd)
    rec("sequential_update_8192", dt, float(n) * n * n2 + float(n) * n2 * n2 + n2**3 / 3.0,
        "posterior(post(x2, s2), y2) with 8192 new observations: bordered Cholesky on the resident factor")
    out["sequential_update_8192"]["ms_all"] = [t * 1e3 for t in ts]
    # check on the TIMED result: (K + σ²I) α = δ over the UNION of the observations, through the Gram-row kernel (no factor involved) at 256 old and
    # 256 new inputs, and on the host for 8 + 8 rows
    p2 = keep.pop("p2")
    a2, d2 = np.asarray(p2.data.alpha, dtype=np.float64), np.asarray(p2.data.delta, dtype=np.float64)
    xa = np.concatenate([x, x2], axis=0)
    idx = np.concatenate([np.linspace(0, n - 1, 256).astype(int), n + np.linspace(0, n2 - 1, 256).astype(int)])
    res_dev = float(np.max(np.abs(p2.mean(agp.RowVecs(xa[idx])) - (d2[idx] - sigma2 * a2[idx]))))
    hidx = np.concatenate([idx[:8], idx[-8:]])
    res_host = float(np.max(np.abs(se_rows(xa[hidx], xa) @ a2 + sigma2 * a2[hidx] - d2[hidx])))
    p2.data.C.free()
    out["sequential_update_8192"]["check"] = {"normal_equations_residual_max_abs_512_rows_device": res_dev,
                                              "normal_equations_residual_max_abs_16_rows_host": res_host, "tol": 1e-8,
                                              "pass": bool(res_dev <= 1e-8 and res_host <= 1e-8)}
    for v in out.values():
        v["statistic"] = "median of 3 after a same-size warm-up"
    return out


def grad_rows(agp, ctx) -> dict:
    """value + gradient of logpdf w.r.t. (variance, scale, noise, y) at C2 and C4."""
    out = {}
    pk = FP64_MFMA_PEAK_TFLOPS * 1e12
    for name, n, seed in (("C2", 16384, 2), ("C4", 65536, 4)):
        x, y = synth_inputs(n, 3, seed)
        fx = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(1.0), ctx=ctx)(agp.RowVecs(x), 0.01)
        for _ in range(1 if name == "C2" else 2):  # warm-up: the first calls at a size allocate the N×N workspaces (3 × 34 GB at C4, against a 96 GB cache cap:
            agp.logpdf_and_grad(fx, y)            # 7.8 s, 6.9 s, then 4.94 s steady — profiles/r4/grad_probe.jsonl)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            lp, g = agp.logpdf_and_grad(fx, y)
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts))
        flops = float(n)**3  # N³/3 factor + 2N³/3 for C⁻¹ (trtri N³/3 + lauum N³/3)
        out[name] = {"ms": dt * 1e3, "flops": flops, "tflops": flops / dt / 1e12, "frac": flops / dt / pk, "logpdf": float(lp),
                     "ms_all": [t * 1e3 for t in ts], "statistic": "median of 3 after warm-up",
                     "what": "logpdf + d/d(variance, scale, noise, y): factor, C^-1 = L^-T L^-1 (triangular inverse + triangular product), one fused gradient pass"}
        # check on the TIMED result: the gradient along the direction (variance, scale, noise) -> (1 ± h)·(variance, scale, noise) against the central
        # difference of two gp_logpdf calls (a different entry point: no C⁻¹, no gradient kernels)
        h = 1e-4
        lps = [float(agp.logpdf(agp.GP((1 + sg * h) * agp.SqExponentialKernel() @ agp.ScaleTransform(1.0 + sg * h), ctx=ctx)(agp.RowVecs(x), 0.01 * (1 + sg * h)), y))
               for sg in (+1, -1)]
        fd = (lps[0] - lps[1]) / (2 * h)
        an = float(g["variance"]) * 1.0 + float(g["scale"]) * 1.0 + float(g["noise"]) * 0.01
        out[name]["check"] = {"directional_derivative": an, "central_difference_of_gp_logpdf": fd, "rel": abs(an - fd) / abs(fd), "tol": 1e-5,
                              "pass": bool(abs(an - fd) <= 1e-5 * abs(fd))}
        ctx.trim()
    return out


def rocsolver_comparator(timeout_s: float = 120.0) -> dict:
    """The comparator in a CHILD process with a hard time limit: loading librocsolver.so / librocblas.so (GBs of code objects) on a box whose
    image is still paging in has taken from 20 s to more than 5 minutes — a yardstick must never take the bench line down or past the
    driver's clock.  The child prints one JSON object; on timeout the line says so."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--comparator-child"], capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"skipped": f"librocsolver did not load and run within {timeout_s:.0f} s on this box (comparator only; see profiles/ for a completed run)"}
    for ln in reversed(r.stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return {"error": f"comparator child rc={r.returncode}: {r.stderr.strip()[-300:]}"}


def rocsolver_comparator_child(sizes=(16384, 32768, 65536)) -> dict:
    """COMPARATOR ONLY (SURVEY.md §7 allows vendor libraries as yardsticks; nothing in the product path loads them): rocsolver_dpotrf of an SPD
    matrix of the same order on the same GPU, run after the timed region through ctypes.  Reports ms and the fraction of the fp64 MFMA peak
    for N³/3 flops, next to which the engine's own factorisation phase can be read."""
    import ctypes as C

    import torch

    out = {}
    try:
        rb = C.CDLL("/opt/rocm/lib/librocblas.so", mode=C.RTLD_GLOBAL)
        rs = C.CDLL("/opt/rocm/lib/librocsolver.so", mode=C.RTLD_GLOBAL)
    except OSError as e:
        return {"error": f"rocSOLVER unavailable: {e}"}
    h = C.c_void_p()
    if rb.rocblas_create_handle(C.byref(h)) != 0:
        return {"error": "rocblas_create_handle failed"}
    rb.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    rs.rocsolver_dpotrf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    try:
        for n in sizes:
            g = torch.randn(n, 64, dtype=torch.float64, device="cuda")
            a0 = g @ g.T
            a0.diagonal().add_(float(n))
            del g
            ts = []
            for rep in range(3 if n <= 32768 else 2):
                a = a0.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = rs.rocsolver_dpotrf(h, 122, n, C.c_void_p(a.data_ptr()), n, C.c_void_p(info.data_ptr()))  # rocblas_fill_lower = 122
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                del a
            dt = min(ts[1:]) if len(ts) > 1 else ts[0]
            out[f"N{n}"] = {"ms": dt * 1e3, "tflops": n**3 / 3 / dt / 1e12, "frac": n**3 / 3 / dt / (FP64_MFMA_PEAK_TFLOPS * 1e12), "status": int(st),
                            "info": int(info.item())}
            del a0
            torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 — a comparator must never take the bench line down
        out["error"] = repr(e)
    finally:
        rb.rocblas_destroy_handle(h)
    out["note"] = "rocsolver_dpotrf (ROCm 7.2) on a random SPD matrix of the same order, same GPU, after the timed region; comparator only"
    return out


def selftest(ngpus: int, virtual: int, grid: str) -> int:
    """Small multi-device fit checked against the single-device engine (same library, same inputs): run in a SUBPROCESS by the
    bench before it trusts a transport (RCCL first, peer copies as the fallback), so that a hanging or failing transport
    costs a timeout instead of the whole run."""
    import abstractgps_jl_amd as agp

    n = 8192
    x, y = synth_inputs(n, 3, 4)
    devices = [0] * virtual if virtual else list(range(ngpus))
    P, Q = (int(v) for v in grid.split("x")) if grid else (0, 0)
    one = agp.Context(devices[0])
    ref = agp.posterior(agp.GP(agp.SqExponentialKernel(), ctx=one)(agp.RowVecs(x), 0.01), y)
    ctx = agp.Context(devices=devices, P=P, Q=Q)
    info = ctx.multi_info()
    post = agp.posterior(agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(x), 0.01), y)
    rel = abs(float(post.logpdf_value) - float(ref.logpdf_value)) / abs(float(ref.logpdf_value))
    arel = float(np.linalg.norm(post.data.alpha - ref.data.alpha) / np.linalg.norm(ref.data.alpha))
    ok = rel <= 1e-10 and arel <= 1e-8
    print(f"[selftest] devices={devices} grid={info['P']}x{info['Q']} comm={info['comm']} logpdf rel {rel:.1e} alpha rel {arel:.1e} "
          f"-> {'OK' if ok else 'MISMATCH'}", flush=True)
    return 0 if ok else 4


def choose_transport(ngpus: int, grid: str) -> str:
    """RCCL unless its self-test fails or hangs; then peer copies (self-tested too).  Returns a note for the JSON line."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                  "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")}
    notes = []
    for comm in ([os.environ["GPMI_COMM"]] if os.environ.get("GPMI_COMM") else ["rccl", "p2p"]):
        env["GPMI_COMM"] = comm
        cmd = [sys.executable, str(Path(__file__).resolve()), "--selftest", "--gpus", str(ngpus)] + (["--grid", grid] if grid else [])
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            notes.append(f"{comm}: rc={r.returncode} {tail}")
            if r.returncode == 0:
                os.environ["GPMI_COMM"] = comm
                return "; ".join(notes)
        except subprocess.TimeoutExpired:
            notes.append(f"{comm}: self-test timed out")
    raise SystemExit("[bench] no working multi-GPU transport: " + "; ".join(notes))


def dry_launcher(args, rank: int, world: int):
    """The process-level protocol of an N > 1 run without a device: every launcher rank joins the gloo group and the barriers, rank 0
    is the driver (here: a stub step that sleeps), the timed region is bracketed by barriers, the maximum over ranks is taken and
    rank 0 alone prints the JSON line.  tests/test_bench_launcher_cpu.py runs this with world_size 2."""
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo")
    have_pg = dist.is_initialized()

    def barrier():
        if have_pg:
            dist.barrier()

    for _ in range(args.warmup):
        if rank == 0:
            time.sleep(0.002)
    barrier()
    t0 = time.perf_counter()
    if rank == 0:
        for _ in range(args.steps):
            time.sleep(0.01)
    barrier()
    dt = time.perf_counter() - t0
    if have_pg:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "dry launcher (no device)", "value": args.n / (dt / args.steps), "unit": "points/s", "n_gpus": max(args.gpus, world),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "strong", "world": world, "driver_rank": 0}), flush=True)
    if have_pg:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--npoints", dest="n", type=int, default=65536, help="(--npoints under torch.distributed.run: its parser claims --n)")
    ap.add_argument("--d", "--dim", dest="d", type=int, default=3)
    ap.add_argument("--nb", type=int, default=0, help="outer panel width override (single GPU) / distribution block (multi GPU)")
    ap.add_argument("--grid", default="", help="process grid PxQ of the multi-device run (default: chosen by the library)")
    ap.add_argument("--depth", type=int, default=0, help="look-ahead depth of the multi-device schedule")
    ap.add_argument("--params", default="", help="ctx parameters name=value,name=value applied before the warm-up (e.g. multi_chain_cus=16,multi_trsm_inv=0: the A/B switches of tools/scale_sweep.sh)")
    ap.add_argument("--vranks", dest="virtual", type=int, default=0, help="V virtual ranks sharing GPU 0 (schedule test / 1-rank overhead measurement)")
    ap.add_argument("--selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--comparator-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-sample-only", action="store_true", help="skip the full-size oracle run (≈ 2-3 minutes at C4); value is then the extrapolation")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C2 / C3 / C5 lines (other_configs)")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run parity properties")
    ap.add_argument("--no-comparator", action="store_true", help="skip the rocsolver_dpotrf comparator (run after the timed region; never on the product path)")
    ap.add_argument("--dry-launcher", action="store_true",
                    help="launcher protocol only (gloo process group, barriers, max over ranks, rank 0 prints the line) with a stub step and no GPU: "
                         "what the world_size-2 CPU test runs")
    args = ap.parse_args()

    if args.selftest:
        sys.exit(selftest(args.gpus, args.virtual, args.grid))
    if args.comparator_child:
        print(json.dumps(rocsolver_comparator_child()), flush=True)
        return

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_launcher:
        return dry_launcher(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    ngpus = max(args.gpus, world)
    if world > 1 and world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}")
    if not args.virtual and torch.cuda.device_count() < ngpus:
        raise SystemExit(f"[bench] --gpus {ngpus} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank if local_rank < torch.cuda.device_count() else 0)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # the library owns the devices and RCCL; the launcher's other ranks only take part in the barriers, over gloo
        dist.init_process_group("gloo")
    have_pg = dist.is_initialized()

    import abstractgps_jl_amd as agp

    n, d = args.n, args.d
    x, y = synth_inputs(n, d, 4)
    kernel = agp.SqExponentialKernel()
    sigma2 = 0.01

    def barrier():
        torch.cuda.synchronize()
        if have_pg:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v: float) -> float:
        if not have_pg:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    extra = {}
    multi = ngpus > 1 or args.virtual > 0
    if True:
        driver = rank == 0  # in-library driver: one process drives every device; launcher ranks > 0 idle at the barriers
        transport = None
        if driver:
            if multi:
                if not args.virtual:
                    transport = choose_transport(ngpus, args.grid)
                P, Q = (int(v) for v in args.grid.split("x")) if args.grid else (0, 0)
                devices = [0] * args.virtual if args.virtual else list(range(ngpus))
                ctx = agp.Context(devices=devices, P=P, Q=Q, nb=args.nb or 0)
                if args.depth:
                    ctx.set_param("lookahead_depth", args.depth)
                for kv in [kv for kv in args.params.split(",") if "=" in kv]:
                    ctx.set_param(kv.split("=")[0].strip(), int(kv.split("=")[1]))
                info = ctx.multi_info()
                info["trsm_inv"], info["chain_cus"] = ctx.get_param("multi_trsm_inv"), ctx.get_param("multi_chain_cus")
            else:
                ctx = agp.Context(local_rank)
                if args.nb:
                    ctx.set_param("nb", args.nb)
                for kv in [kv for kv in args.params.split(",") if "=" in kv]:
                    ctx.set_param(kv.split("=")[0].strip(), int(kv.split("=")[1]))
            f = agp.GP(kernel, ctx=ctx)
            fx = f(agp.RowVecs(x), sigma2)

            def step():
                return agp.posterior(fx, y)  # one library call: Gram, Cholesky, logdet, solves -> (logpdf, α)

            for _ in range(args.warmup):
                step().data.C.free()
        # ---- the timed region: the production configuration (no per-kernel instrumentation)
        phases = {"assemble_ms": 0.0, "potrf_ms": 0.0, "solve_ms": 0.0}
        barrier()
        t0 = time.perf_counter()
        post = None
        if driver:
            for _ in range(args.steps):
                if post is not None:
                    post.data.C.free()
                post = step()
                tm = ctx.timings()  # phase events recorded by every call (not per-kernel)
                for kname in phases:
                    phases[kname] += tm[kname] / args.steps
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        if driver:
            note(f"timed region done: {dt / args.steps * 1e3:.1f} ms per step")
            logpdf_val, alpha = float(post.logpdf_value), post.data.alpha
            # ---- separate, untimed instrumented pass: every MFMA GEMM launch bracketed by HIP events on its own stream
            post.data.C.free()
            ctx.set_param("time_kernels", 1)
            post = step()
            tm = ctx.timings()
            ctx.set_param("time_kernels", 0)
            gemm_ms, gemm_flops, gemm_bytes, gemm_launches = tm["gemm_ms"], tm["gemm_flops"], tm.get("gemm_bytes", 0.0), tm["gemm_launches"]
            kernel_tflops = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            # the same launches with the look-ahead off: every GEMM then has the machine to itself, so its event-bracketed duration is the
            # kernel's own (with the look-ahead the register-resident leaves of the panel stream share CUs with the update, which shortens the
            # wall time and lengthens each launch as the events — and rocprofv3 — see it)
            kernel_alone = None
            note("instrumented pass done")
            if not multi:
                post.data.C.free()
                la_prev = ctx.get_param("lookahead")
                ctx.set_param("lookahead", 0)
                ctx.set_param("time_kernels", 1)
                post = step()
                tm0 = ctx.timings()
                ctx.set_param("time_kernels", 0)
                ctx.set_param("lookahead", la_prev)
                if tm0["gemm_ms"] > 0:
                    kernel_alone = tm0["gemm_flops"] / (tm0["gemm_ms"] * 1e-3) / 1e12
            note("look-ahead-off pass done")
            mfma_ceiling = agp._lib.C.c_double()
            agp._lib.check(ctx.lib.gp_bench_mfma_f64(ctx.handle, 20000, agp._lib.C.byref(mfma_ceiling)))
            pair_tf = f_pair(n) / (dt / args.steps) / 1e12
            peak = FP64_MFMA_PEAK_TFLOPS * (1 if args.virtual else ngpus)
            # roofline: the SURVEY.md §8(d) number — F_pair / t_pair over the whole job — with the dominant kernel's own rate beside it
            roofline = {"bound": "mfma", "achieved": pair_tf, "peak": peak, "unit": "TFLOP/s",
                        "frac": pair_tf / peak,
                        **(pmc_traffic(n) if not multi else {"traffic": None, "traffic_note": "HBM/fabric bytes come from separate rocprofv3 --pmc passes over the "
                           "single-GPU command (profiles/r*/pmc_bench_summary.json: same kernel, per launch); PMC cannot be sampled inside a timed multi-GPU run"}),
                        "definition": "achieved = (N^3/3 + 3N^2) / wall time of one pair (SURVEY.md 8(d)), peak = 78.6 TF/s x n_gpus; "
                                      "kernel_* = the dominant kernel alone" + (" (rank 0's launches)" if multi else ""),
                        "kernel": "gemm_nt_dma_kernel<double> (v_mfma_f64_16x16x4_f64 trailing update, LDS-DMA operands; launches of <= 4096 tiles run its "
                                  "persistent stream-K variant gemm_nt_sk_kernel — kernel_* and the per-launch averages cover both)",
                        "kernel_achieved": kernel_tflops, "kernel_frac": kernel_tflops / FP64_MFMA_PEAK_TFLOPS,
                        "kernel_achieved_lookahead_off": kernel_alone, "kernel_frac_lookahead_off": None if kernel_alone is None else kernel_alone / FP64_MFMA_PEAK_TFLOPS,
                        "kernel_timing": "separate untimed pass with time_kernels=1 (HIP events around each launch on its stream)",
                        "algorithmic_bytes_per_launch_avg": gemm_bytes / max(gemm_launches, 1),
                        "launches_per_step": gemm_launches,
                        "avg_launch_ms": gemm_ms / max(gemm_launches, 1),
                        "flops_per_launch_avg": gemm_flops / max(gemm_launches, 1),
                        "measured_mfma_f64_ceiling_tflops": mfma_ceiling.value}
            extra["phases_ms"] = phases
            if not multi and phases["assemble_ms"] > 0:  # the Gram phase is HBM-write bound: 8·N(N+1)/2 written once + 8·N·D read
                kb = 8.0 * n * (n + 1) / 2 + 8.0 * n * d
                gbps = kb / (phases["assemble_ms"] * 1e-3) / 1e9
                roofline["kmat"] = {"kernel": "kmat_kernel<double>", "bound": "hbm", "algorithmic_bytes": kb, "ms": phases["assemble_ms"], "achieved": gbps,
                                    "unit": "GB/s", "frac_of_8TBps_spec": gbps / (HBM_SPEC_TBPS * 1e3), "frac_of_6.29TBps_copy": gbps / (HBM_COPY_TBPS * 1e3)}
            if not args.no_check and n == 65536:
                extra.update(digest_check(n, logpdf_val, alpha))
            if not args.no_check:  # size-independent parity properties at full size
                r = np.asarray(post.data.delta, dtype=np.float64)
                # (K + σ²I) α = δ  checked through a second, independent device path: posterior mean at the
                # training inputs is K α = δ − σ² α
                idx = np.linspace(0, n - 1, 512).astype(int)
                m_tr = post.mean(agp.RowVecs(x[idx]))
                extra["check_residual_max"] = float(np.max(np.abs(m_tr - (r[idx] - sigma2 * alpha[idx]))))
            note("parity checks done")
            if not multi and not args.no_other_configs and n == 65536:
                nxt = next_rows(agp, ctx, post, x, y, n, sigma2)
                note("next rows (predict / cov / sequential update) done")
                post.data.C.free()
                ctx.trim()
                extra["other_configs"] = other_configs(agp, ctx)
                note("other configs (C2 / C3 / C5) done")
                nxt["value_and_gradient"] = grad_rows(agp, ctx)
                note("value + gradient rows done")
                extra["other_configs"]["next"] = nxt
                ctx.trim()
                if not args.no_comparator:
                    extra["comparator_rocsolver_dpotrf"] = rocsolver_comparator()
                    note("rocsolver comparator done")
            if multi:
                extra["multi_stats"] = ctx.multi_stats()  # fits / retries (repetitions after a failed self-check) / solves: a non-zero retry count is never silent
            if multi:
                parallelism = (f"in-library 2D block-cyclic {info['P']}x{info['Q']}, nb={info['nb']}, look-ahead {info['lookahead_depth']}, "
                               f"transport {info['comm']}, rows-below solve {'inverse block' if info['trsm_inv'] else 'substitution'}, "
                               f"chain stream {'on ' + str(info['chain_cus']) + ' masked CUs' if info['chain_cus'] else 'unmasked'}"
                               + (f" [{args.virtual} virtual ranks on one GPU]" if args.virtual else "")
                               + ("; one driver process, launcher ranks > 0 idle" if world > 1 else ""))
                if transport:
                    extra["transport_selftest"] = transport
            else:
                parallelism = "1 GPU"
        scaling = "strong"
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n / (dt / args.steps)
        line = {
            "metric": "logpdf+posterior throughput (points/sec, fp64) at N=65536; % MFMA roofline",
            "value": value, "unit": "points/s", "n_gpus": 1 if args.virtual else ngpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C4: GP(SqExponentialKernel()) on {n} {d}-D points, sigma2=0.01, fp64, "
                                   "one (logpdf, posterior-fit) pair per step", "n": n, "d": d,
                       "parallelism": parallelism},
            "roofline": roofline,
            "logpdf": logpdf_val,
            "lib_sha16": lib_sha16(),
        }
        line.update(extra)
        if not multi and not args.no_cpu_baseline:
            cb = cpu_baseline(n, d, full=not args.cpu_baseline_sample_only)
            a_cpu = cb.pop("_alpha", None)
            if a_cpu is not None:  # the measured oracle run is also a live parity check of the timed result (1e-10 / 1e-8)
                cb["engine_vs_this_run"] = {"logpdf_rel": abs(logpdf_val - cb["logpdf"]) / abs(cb["logpdf"]),
                                            "alpha_rel": float(np.linalg.norm(np.asarray(alpha, dtype=np.float64) - a_cpu) / np.linalg.norm(a_cpu))}
                cb["engine_vs_this_run"]["pass"] = bool(cb["engine_vs_this_run"]["logpdf_rel"] <= 1e-10 and cb["engine_vs_this_run"]["alpha_rel"] <= 1e-8)
            line["cpu_baseline"] = cb
            note("cpu baseline done")
        elif multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cached_cpu_baseline(n) or cpu_baseline(n, d, full=False)
        print(json.dumps(line), flush=True)
    if have_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
