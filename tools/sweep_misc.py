"""Stream-K for the few-tile GEMMs and the diagonal block of the vector solves, at the small / mid sizes and C5."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import sweep_r2 as S  # noqa: E402

agp = S.agp
for p in ({}, {"gemm_streamk": 1}, {"trsv_nb": 512}, {"trsv_nb": 256}, {"trsv_nb": 128}, {"gemm_streamk": 1, "trsv_nb": 256}):
    q = {"gemm_streamk": 0, "trsv_nb": 1024, **p}
    S.exact("M4096", 4096, 3, 2, agp.SqExponentialKernel(), q, reps=4)
    S.exact("M8192", 8192, 3, 2, agp.SqExponentialKernel(), q, reps=3)
    S.exact("C2", 16384, 3, 2, agp.SqExponentialKernel(), q)
    S.exact("C3", 32768, 3, 2, agp.SqExponentialKernel(), q, reps=2)
    S.vfe(q)
