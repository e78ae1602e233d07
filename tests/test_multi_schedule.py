"""CPU: happens-before check of the multi-device schedule (tools/multi_schedule_check.py replays the stream operations, event
waits and block accesses of abstractgps.jl_amd/csrc/multi.hip::fit_rank for the copy transport).  Every pair of operations that
touches the same matrix block / operand-buffer slot / partial-sum block with at least one write must be ordered."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import multi_schedule_check as M  # noqa: E402


@pytest.mark.parametrize("rccl", [False, True], ids=["copies", "rccl"])
@pytest.mark.parametrize("grid", M.GRIDS)
def test_no_unordered_conflicts(grid, rccl):
    """both transports: consumer pulls (hipMemcpy2DAsync / copy kernel) and the grouped ncclSend / ncclRecv pairs with their
    staging images (the RCCL schedule has never run on more than one GPU: this is its only check besides the API test)"""
    P, Q = grid
    for nblk in (1, 2, 3, 5, 9, 17):
        for depth in (1, 2, 3):
            g = M.build(P, Q, nblk, depth, rccl=rccl)
            rs = M.races(g)
            assert not rs, (grid, nblk, depth, rs[:4])
            if rccl:  # point-to-point operations of a pair of ranks match by posting order
                assert not M.rccl_pair_order(g, P * Q), (grid, nblk, depth)


def _variant(old, new):
    src = (ROOT / "tools" / "multi_schedule_check.py").read_text().split("if __name__")[0]
    assert old in src
    ns = {}
    exec(compile(src.replace(old, new), "variant", "exec"), ns)
    return ns


def test_checker_finds_the_round2_bug():
    """the dependency that was missing during development (a rank in the owner column whose updates read its own matrix did
    not wait for its own panel when it pulled nothing from itself): grids with gcd(P, Q) > 1 must be flagged without it"""
    ns = _variant("        if q == qk:\n            sc[r].wait(ready[r][k])\n", "")
    assert ns["races"](ns["build"](2, 2, 9, 2))
    assert ns["races"](ns["build"](4, 2, 9, 2))
    assert not ns["races"](ns["build"](2, 1, 9, 2))


def test_checker_finds_missing_bulk_wait_and_buffer_reuse():
    ns = _variant("        if i == first and first - 1 >= 0 and bulk_done[r][first - 1] is not None:\n"
                  "            sp[r].wait(bulk_done[r][first - 1])\n", "")
    assert ns["races"](ns["build"](2, 2, 9, 2))
    ns = _variant("            sc[r].wait(bulk_done[r][k - NBUF])\n            sc[r].wait(la_done[r][k - NBUF])\n", "            pass\n")
    assert ns["races"](ns["build"](2, 2, 9, 2))


def test_checker_finds_unstaged_rccl_sends():
    """RCCL transport: without the wait for the rank's own panel in front of the group, the sends would read the staging image
    before the panel stream has written it"""
    ns = _variant("            if q == qk:\n                sc[r].wait(ready[r][k])\n        sends, recvs", "        sends, recvs")
    assert ns["races"](ns["build"](2, 2, 9, 2, rccl=True))


@pytest.mark.parametrize("grid", M.GRIDS)
def test_rank_threads_cannot_block_each_other(grid):
    """host side: every generation-numbered event a rank thread spins on is published by its owner without that owner waiting,
    directly or through other ranks, for the spinning thread"""
    P, Q = grid
    for nblk in (1, 2, 3, 5, 9, 17):
        for depth in (1, 2, 3):
            assert not M.host_deadlock(P, Q, nblk, depth), (grid, nblk, depth)
