import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass on a box without a GPU: they are skipped with a loud reason when
    # deselected by "-m 'not gpu'", and ERROR (not skip) if explicitly selected without a device.
    if _have_gpu():
        return
    for item in items:
        if "gpu" in item.keywords and "gpu" not in (config.getoption("-m") or ""):
            item.add_marker(pytest.mark.skip(reason="no GPU in this container (gpu tests run via gpurun)"))


def rank_devices(n: int) -> list:
    """Device list for the n ranks of a multi-device context in the tests: n VIRTUAL ranks on GPU 0 — what the one-GPU test box
    runs — unless GPMI_TEST_REAL_DEVICES=1 and the node has at least n GPUs: then one rank per device, i.e. the same tests over
    real peer copies / real RCCL (the first thing to run on a multi-GPU node: `GPMI_TEST_REAL_DEVICES=1 pytest tests -m gpu -k multi`)."""
    if os.environ.get("GPMI_TEST_REAL_DEVICES") == "1":
        try:
            import torch

            if torch.cuda.device_count() >= n:
                return list(range(n))
        except Exception:
            pass
    return [0] * n


@pytest.fixture(scope="session")
def agp():
    # the .so is a build artefact (git-ignored): cross-compile it for gfx950 if this checkout has not been built yet
    lib = ROOT / "abstractgps.jl_amd" / "csrc" / "libgpmi355.so"
    if not lib.exists():
        import __graft_entry__

        __graft_entry__.build()
    import abstractgps_jl_amd as m

    return m


@pytest.fixture(scope="session")
def ctx(agp):
    return agp.default_context(0)
