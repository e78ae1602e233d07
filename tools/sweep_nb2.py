"""Panel width / look-ahead after the leaf and stream-K changes (C2, C3, C4)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
import sweep_r2 as S
agp = S.agp
for nb in (1024, 2048, 4096):
    for la in (0, 1):
        p = {"nb": nb, "lookahead": la}
        S.exact("C2", 16384, 3, 2, agp.SqExponentialKernel(), p)
        S.exact("C3", 32768, 3, 2, agp.SqExponentialKernel(), p, reps=2)
for nb, la in ((2048, 0), (2048, 1), (4096, 1), (1024, 1)):
    S.exact("C4", 65536, 3, 4, agp.SqExponentialKernel(), {"nb": nb, "lookahead": la}, reps=1)
