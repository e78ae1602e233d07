"""CPU: the operand / accumulator lane maps the register-resident leaf and the in-panel update kernel (csrc/leaf.hpp) are written against, emulated
lane by lane in NumPy (tools/leaf_emu.py) — the products P1 / P2 and the update kernel's wave tile reproduce plain matrix arithmetic."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))


def test_register_layout_products_match_matrix_arithmetic():
    import leaf_emu

    for name, dev in leaf_emu.main().items():
        assert dev < 1e-12, (name, dev)
