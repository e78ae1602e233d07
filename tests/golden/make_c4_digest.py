#!/usr/bin/env python
"""Digest of the ORACLE's result at the headline size (BASELINE config C4: N = 65 536, D = 3, SE, σ² = 0.01, seed 4).

  python tests/golden/make_c4_digest.py [--n 65536] [--threads 8] [--out tests/golden/digests/c4_oracle_digest.npz]

Runs oracle.gp_oracle.logpdf_and_posterior_inplace (the fused pair in one Fortran-ordered N×N array: ≈ 35 GB of host
memory, minutes of host BLAS) and stores what a checker needs to pin a device result WITHOUT re-running the oracle:

  logpdf, logdet, ‖α‖₂, α[::64] (1 024 entries), and the projections of α on 8 seeded Gaussian vectors
  (PCG64 seed 20260926 + i) — a wrong α that keeps its norm, its sampled entries AND all eight projections would have to be
  wrong in a 65 536 − 1 033 dimensional subspace by design, not by accident.

bench.py compares the timed engine's logpdf / α against this file on every run (`check_logpdf_rel_vs_digest`,
`check_alpha_rel_vs_digest`), tests/test_gpu_fullsize.py asserts it, and the GPU test-suite ALSO re-runs the oracle itself
at this size (test_c4_full_size_values_vs_oracle).  Test infrastructure: the product never reads this file.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

PROJ_SEED = 20260926
NPROJ = 8
STRIDE = 64


def digest_of(alpha: np.ndarray) -> dict:
    n = alpha.shape[0]
    proj = np.empty(NPROJ)
    for i in range(NPROJ):
        g = np.random.default_rng(PROJ_SEED + i).standard_normal(n)
        proj[i] = float(g @ alpha) / np.sqrt(n)
    return {"alpha_norm": float(np.linalg.norm(alpha)), "alpha_sub": np.array(alpha[::STRIDE]), "alpha_proj": proj}


def compare(alpha: np.ndarray, dig) -> float:
    """max of the relative deviations of norm / sampled entries / projections (each relative to the oracle's α scale)."""
    mine = digest_of(np.asarray(alpha, dtype=np.float64))
    scale = float(dig["alpha_norm"]) / np.sqrt(alpha.shape[0])  # rms entry of the oracle's α
    e_norm = abs(mine["alpha_norm"] - float(dig["alpha_norm"])) / float(dig["alpha_norm"])
    e_sub = float(np.linalg.norm(mine["alpha_sub"] - dig["alpha_sub"]) / np.linalg.norm(dig["alpha_sub"]))
    e_proj = float(np.max(np.abs(mine["alpha_proj"] - dig["alpha_proj"])) / scale)
    return max(e_norm, e_sub, e_proj)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=str(Path(__file__).resolve().parent / "digests" / "c4_oracle_digest.npz"))
    a = ap.parse_args()
    from oracle import gp_oracle as o

    x, y = o.synth_inputs(a.n, a.d, a.seed)
    tm = {}
    t0 = time.perf_counter()
    lp, alpha, logdet = o.logpdf_and_posterior_inplace(o.FiniteGP(o.GP(o.Kernel(o.SE)), x, 0.01), y, threads=a.threads, timings=tm)
    wall = time.perf_counter() - t0
    d = digest_of(alpha)
    np.savez(a.out, n=a.n, d=a.d, seed=a.seed, sigma2=0.01, logpdf=lp, logdet=logdet, oracle_wall_s=wall,
             gram_s=tm["gram_s"], potrf_s=tm["potrf_s"], solves_s=tm["solves_s"], **d)
    print(f"n={a.n} logpdf={lp!r} logdet={logdet!r} |alpha|={d['alpha_norm']!r} wall={wall:.1f}s -> {a.out}", flush=True)


if __name__ == "__main__":
    main()
