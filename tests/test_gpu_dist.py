"""GPU: the block-cyclic driver with the PRODUCT backend (HipTileBackend -> libgpmi355).
 * world 1 in-process;
 * 2 and 4 ranks sharing the single GPU of the test box (gloo carrying the device tensors): the real HIP kernels
   run under the P×Q block-cyclic tile predicate, the panel broadcasts and the distributed backward sweep.
On an 8-GPU node the same code runs one rank per GPU over RCCL (bench.py --gpus N)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _ref(n, d):
    from oracle import gp_oracle as o

    x, y = o.synth_inputs(n, d, 55)
    lp, post = o.logpdf_and_posterior(o.FiniteGP(o.GP(o.Kernel(o.SE, 1.5, 0.9)), x, 0.02), y)
    return x, y, lp, post.alpha


def test_world1_matches_oracle(agp):
    from abstractgps_jl_amd import dist as gdist

    x, y, lp, alpha = _ref(1500, 3)
    eng = gdist.BlockCyclicEngine(0, nb=256)
    res = eng.fit(1.5 * agp.SqExponentialKernel() @ agp.ScaleTransform(0.9), x, 0.02, y)
    assert res["info"] == 0
    assert res["logpdf"] == pytest.approx(lp, rel=1e-10)
    assert np.linalg.norm(res["alpha"] - alpha) <= 1e-8 * np.linalg.norm(alpha)


def _worker(rank, world, port, grid, n, d, nb, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import abstractgps_jl_amd as agp
        from abstractgps_jl_amd import dist as gdist
        from oracle import gp_oracle as o

        x, y = o.synth_inputs(n, d, 55)
        eng = gdist.BlockCyclicEngine(0, nb=nb, grid=grid)
        res = eng.fit(1.5 * agp.SqExponentialKernel() @ agp.ScaleTransform(0.9), x, 0.02, y)
        if rank == 0:
            q.put((res["logpdf"], res["alpha"], res["info"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,grid,n,nb", [(2, (1, 2), 1500, 256), (2, (2, 1), 1100, 128), (4, (2, 2), 2100, 256),
                                            (8, (2, 4), 2600, 128)])
def test_virtual_ranks_on_one_gpu(world, grid, n, nb):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, grid, n, 3, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    lp_g, alpha_g, info = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x, y, lp, alpha = _ref(n, 3)
    assert info == 0
    assert lp_g == pytest.approx(lp, rel=1e-10)
    assert np.linalg.norm(alpha_g - alpha) <= 1e-8 * np.linalg.norm(alpha)
