"""Synthetic workload generator shared by the measurement tools (same recipe as bench.py / SURVEY.md §8(d))."""
import numpy as np


def synth_inputs(n: int, d: int, seed: int):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    eps = rng.standard_normal(n)
    y = np.sin(X.sum(axis=1)) + 0.1 * eps
    return (X[:, 0].copy() if d == 1 else X), y
