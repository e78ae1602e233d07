"""CPU (gloo, world_size 2 and 4): the multi-process 2D block-cyclic schedule of abstractgps.jl_amd/dist.py —
tile ownership, panel/diagonal broadcasts, the B-operand gather, the distributed backward sweep and the
scalar all-reduces — with a NumPy tile backend standing in for the HIP library.  Result must equal the
oracle's logpdf and α."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, grid, n, d, nb, kind, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import abstractgps_jl_amd as agp
        from abstractgps_jl_amd import dist as gdist
        from oracle import gp_oracle as o
        from tests._np_backend import NumpyTileBackend

        x, y = o.synth_inputs(n, d, 77)
        k = agp.Kernel(kind) @ agp.ScaleTransform(0.8)
        eng = gdist.BlockCyclicEngine(0, nb=nb, backend=NumpyTileBackend(), grid=grid)
        res = eng.fit(2.0 * k, x, 0.05, y, mean=np.full(n, 0.25))
        if rank == 0:
            q.put((res["logpdf"], res["alpha"], res["info"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,grid,n,d,nb", [(2, (1, 2), 700, 3, 128), (2, (2, 1), 515, 1, 256), (4, (2, 2), 900, 2, 128),
                                               (1, (1, 1), 300, 3, 128), (8, (2, 4), 1300, 3, 128), (4, (1, 4), 1100, 2, 128)])
def test_block_cyclic_fit_matches_oracle(world, grid, n, d, nb):
    from oracle import gp_oracle as o

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, grid, n, d, nb, o.MATERN52, q)) for r in range(world)]
    for p in procs:
        p.start()
    lp, alpha, info = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x, y = o.synth_inputs(n, d, 77)
    ref_lp, ref_post = o.logpdf_and_posterior(
        o.FiniteGP(o.GP(o.Kernel(o.MATERN52, 2.0, 0.8), 0.25), x, 0.05), y)
    assert info == 0
    assert lp == pytest.approx(ref_lp, rel=1e-10)
    assert np.linalg.norm(alpha - ref_post.alpha) <= 1e-8 * np.linalg.norm(ref_post.alpha)


def test_choose_grid():
    sys.path.insert(0, str(ROOT))
    from abstractgps_jl_amd import dist as gdist

    assert [gdist.choose_grid(w) for w in (1, 2, 4, 8)] == [(1, 1), (1, 2), (2, 2), (2, 4)]
