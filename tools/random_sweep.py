"""One-off wider run of tests/test_gpu_random.py's case generator (N seeds), printing failures."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from tests.test_gpu_random import _case  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for seed in range(n):
    try:
        _case(agp, np.random.default_rng(5000 + seed))
    except Exception as e:  # report and continue
        bad += 1
        print("FAIL seed", 5000 + seed, repr(e)[:300], flush=True)
print(f"{n - bad}/{n} passed")
