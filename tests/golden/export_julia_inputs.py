#!/usr/bin/env python
"""Writes tests/golden/julia_inputs/<case>.gpb — the INPUTS of every committed fixture tests/golden/*.npz (the same bits the Python
oracle and the device tests read) plus the split points of the sequential / update cases — for tests/golden/make_golden.jl, which
runs the REAL AbstractGPs.jl on them.  Re-run after make_golden.py:  python tests/golden/export_julia_inputs.py"""
import glob
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from tests.golden import gpb  # noqa: E402

INPUT_KEYS = ("x", "y", "Y", "xs", "kind", "variance", "scale", "sigma2", "mean", "z", "jitter")


def split_points(n: int, m: int) -> dict:
    """n1: observations of the first fit (the remaining n − n1 arrive by sequential conditioning / update_posterior);
    m1: pseudo-points of the first sparse fit (the remaining m − m1 are appended by update_posterior(post, fz))."""
    return {"n1": float(n - max(2, n // 4)), "m1": float(m - max(2, m // 4))}


def main():
    for p in sorted(glob.glob(str(HERE / "*.npz"))):
        g = np.load(p)
        arrays = {k: g[k] for k in INPUT_KEYS}
        arrays.update(split_points(g["x"].shape[0], g["z"].shape[0]))
        out = HERE / "julia_inputs" / (Path(p).stem + ".gpb")
        gpb.write(out, arrays)
        back = gpb.read(out)
        assert all(np.array_equal(np.asarray(back[k]), np.asarray(arrays[k], dtype=np.float64), equal_nan=True) for k in arrays)
        print(out.relative_to(HERE.parent.parent))


if __name__ == "__main__":
    main()
