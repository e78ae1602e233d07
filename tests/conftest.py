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


def documented_defaults() -> dict:
    """GPMI355_PARAM_DEFAULTS of include/gpmi355.h: name -> documented default of every single-device ctx parameter."""
    import re

    txt = (ROOT / "include" / "gpmi355.h").read_text()
    m = re.search(r"#define GPMI355_PARAM_DEFAULTS(.*?)\nint32_t gp_ctx_set_param", txt, flags=re.S)
    assert m, "GPMI355_PARAM_DEFAULTS not found in include/gpmi355.h"
    body = "".join(re.findall(r'"([^"]*)"', m.group(1)))
    out = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in body.split(",") if kv}
    # experiment runs only (e.g. the whole suite on a candidate default: GPMI_PARAMS=leaf_rank4=1 GPMI_TEST_EXPECT=leaf_rank4=1): the values the
    # default context is EXPECTED to hold instead of the header's — never set by the driver's runs
    for kv in [kv for kv in os.environ.get("GPMI_TEST_EXPECT", "").split(",") if "=" in kv]:
        out[kv.split("=")[0]] = int(kv.split("=")[1])
    return out


@pytest.fixture(autouse=True)
def _default_context_has_production_defaults(request):
    """Before every GPU test: every tunable of the SHARED default context reads back its documented default (include/gpmi355.h).  Round 4
    ran half of the suite with `gemm_streamk = 0` because one test's clean-up left it so; a test that needs another setting uses its own
    Context or restores what it changed — and this fixture fails the NEXT test loudly if it does not."""
    if "gpu" not in request.keywords or not _have_gpu():
        yield
        return
    import abstractgps_jl_amd as m

    c = m.default_context(0)

    def drift():
        return {k: (c.get_param(k), v) for k, v in documented_defaults().items() if c.get_param(k) != v}

    bad = drift()
    assert not bad, f"default context left with non-default parameters by an earlier test (name: (value, documented default)): {bad}"
    yield
    bad = drift()
    assert not bad, f"{request.node.name} left non-default parameters on the shared default context: {bad}"


@pytest.fixture
def exact_mode(request, agp):
    """Parametrised 'default' / 'no_atomics': the second runs the test with gemm_streamk = 0 and deterministic = 1 on the default context
    (hardware-dispatched GEMMs only, no floating-point atomics anywhere in the exact path) and restores the defaults afterwards — the
    non-default path covered on purpose, not by a leaked setting."""
    mode = getattr(request, "param", "default")
    c = agp.default_context(0)
    if mode == "no_atomics":
        c.set_param("gemm_streamk", 0)
        c.set_param("deterministic", 1)
    try:
        yield mode
    finally:
        c.set_param("gemm_streamk", 1)
        c.set_param("deterministic", 0)
