import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
import sweep_r2 as S
agp = S.agp
for p in ({"gemm_streamk": 0, "trsv_nb": 1024}, {"gemm_streamk": 1, "trsv_nb": 1024}, {"gemm_streamk": 1, "trsv_nb": 256}, {"gemm_streamk": 0, "trsv_nb": 256}):
    S.exact("C4", 65536, 8, 4, agp.Matern52Kernel(), p, reps=2)
