"""C5 (VFE fp32) parameter sweep: row padding (HBM channel aliasing of the 64 KiB-strided Y rows), chunk size, GEMM variants."""
import itertools
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools.sweep_r2 import vfe  # noqa: E402
import abstractgps_jl_amd as agp  # noqa: E402

base = {"ldpad": 32, "vfe_chunk": 16384, "xcd_swizzle": 0, "gemm_pad_lds": 0}
variants = [{}, {"ldpad": 64}, {"ldpad": 96}, {"ldpad": 160}, {"ldpad": 288}, {"ldpad": 544}, {"vfe_chunk": 8192, "ldpad": 96},
            {"vfe_chunk": 32768, "ldpad": 96}, {"xcd_swizzle": 1}, {"xcd_swizzle": 1, "ldpad": 96}, {"gemm_pad_lds": 20480}]
for v in variants:
    vfe({**base, **v}, reps=2)
