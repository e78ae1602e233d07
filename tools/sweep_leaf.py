"""Left-looking leaf groups (leaf_group = 64 / 128 / 256 / 512) at the exact configs and the VFE config."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import sweep_r2 as S  # noqa: E402

agp = S.agp
which = sys.argv[1:] or ["M", "C2", "C3", "C5", "C4"]
for lg in (64, 128, 256, 512):
    p = {"leaf_group": lg}
    if "M" in which:
        S.exact("M4096", 4096, 3, 2, agp.SqExponentialKernel(), p, reps=4)
        S.exact("M8192", 8192, 3, 2, agp.SqExponentialKernel(), p, reps=3)
    if "C2" in which:
        S.exact("C2", 16384, 3, 2, agp.SqExponentialKernel(), p)
    if "C3" in which:
        S.exact("C3", 32768, 3, 2, agp.SqExponentialKernel(), p, reps=2)
    if "C5" in which:
        S.vfe(p)
if "C4" in which:
    for lg in (64, 256):
        S.exact("C4", 65536, 8, 4, agp.Matern52Kernel(), {"leaf_group": lg}, reps=1)
