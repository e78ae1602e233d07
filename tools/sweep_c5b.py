"""C5 sweep 2: three-stage operand ring, partial-product length, stream-K on the M×M side."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools.sweep_r2 import vfe  # noqa: E402

ORACLE_ELBO_F32_INPUTS = -57800.799534829974  # profiles/r2/fullsize_parity.jsonl (fp64 oracle on the fp32-representable inputs)
base = {"vfe_chunk": 16384, "gemm_ring3": 0, "vfe_ks": 2048, "vfe_sk": 0}
for v in [{}, {"gemm_ring3": 1}, {"vfe_ks": 4096}, {"vfe_sk": 1}, {"gemm_ring3": 1, "vfe_ks": 4096, "vfe_sk": 1},
          {"gemm_ring3": 1, "vfe_ks": 4096, "vfe_sk": 1, "vfe_chunk": 32768}, {"gemm_ring3": 1, "vfe_ks": 8192, "vfe_chunk": 32768}]:
    vfe({**base, **v}, reps=2)
print("oracle", ORACLE_ELBO_F32_INPUTS)
