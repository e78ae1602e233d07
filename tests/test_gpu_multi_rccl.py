"""GPU: the RCCL transport of the in-library multi-device driver (csrc/multi.hip, comm == 1) — staging images, grouped
ncclSend / ncclRecv in matching posting order, the L_kk image hand-over, the backward-sweep transfers — EXECUTED with virtual
ranks on the one GPU of the test box.  The real librccl refuses several ranks on one device, so GPMI_RCCL_LIB points the
library at tests/rccl_mock (same seven entry points, RCCL's point-to-point matching rules, host rendezvous + hipMemcpyAsync;
unmatched or mismatching operations are errors).  Every grid is compared with the CPU oracle at the single-GPU tolerances,
with the schedule diagnostics on ("multi_check" 7: event markers, operand verification, NaN-poisoned buffers)."""
import ctypes
import os
from pathlib import Path

import numpy as np
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
MOCK = ROOT / "tests" / "rccl_mock" / "librccl_mock.so"
GRIDS = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (4, 2), (2, 4), (8, 1), (3, 1), (2, 3)]


def _relnorm(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture()
def rccl_env():
    if not MOCK.exists():
        import __graft_entry__

        __graft_entry__.build_mock()
    old = {k: os.environ.get(k) for k in ("GPMI_COMM", "GPMI_RCCL_LIB", "RCCL_MOCK_TIMEOUT_S")}
    os.environ.update(GPMI_COMM="rccl", GPMI_RCCL_LIB=str(MOCK), RCCL_MOCK_TIMEOUT_S="60")
    yield ctypes.CDLL(str(MOCK))
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _stats(mock):
    s, r, u = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    mock.rcclMockStats(ctypes.byref(s), ctypes.byref(r), ctypes.byref(u))
    return s.value, r.value, u.value


@pytest.mark.parametrize("P,Q", GRIDS, ids=lambda v: str(v))
def test_rccl_transport_virtual_ranks_vs_oracle(agp, rccl_env, P, Q):
    n, d, nb = 1500, 3, 128
    x, y = o.synth_inputs(n, d, 40 + P * 10 + Q)
    rng = np.random.default_rng(P * 100 + Q)
    s2 = 0.02 + 0.05 * rng.random(n)
    of = o.GP(o.Kernel(o.MATERN52, 1.4, 0.8), 0.25)
    ofx = o.FiniteGP(of, x, s2)
    lp_ref, opost = o.logpdf_and_posterior(ofx, y)
    s0, r0, _ = _stats(rccl_env)
    ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
    try:
        assert ctx.multi_info()["comm"] == "rccl"
        ctx.set_param("multi_check", 7)
        ctx.set_param("multi_timeout_s", 120)
        f = agp.GP(0.25, 1.4 * agp.Matern52Kernel() @ agp.ScaleTransform(0.8), ctx=ctx)
        fx = f(agp.RowVecs(x), s2)
        assert agp.logpdf(fx, y) == pytest.approx(lp_ref, rel=1e-10)
        Y = np.stack([y, np.cos(y), 0.3 * y], axis=1)
        np.testing.assert_allclose(agp.logpdf(fx, Y), o.logpdf(ofx, Y), rtol=1e-10)
        post = agp.posterior(fx, y)
        assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10)
        assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
        assert np.max(np.abs(post.data.C.U - opost.U)) <= 1e-10
        s1, r1, unmatched = _stats(rccl_env)
        assert unmatched == 0 and s1 - s0 == r1 - r0
        if P * Q > 1:
            assert s1 > s0, "the stand-in library served no transfer: the RCCL branch did not run"
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [1, 2, 3])
@pytest.mark.parametrize("P,Q,n,nb", [(2, 2, 2300, 256), (4, 1, 1111, 128), (1, 3, 1000, 128), (2, 4, 2049, 128)])
def test_rccl_transport_depths_and_ragged_sizes(agp, rccl_env, depth, P, Q, n, nb):
    x, y = o.synth_inputs(n, 2, 7 + depth)
    of = o.GP(o.Kernel(o.SE, 1.0, 1.3))
    lp_ref, opost = o.logpdf_and_posterior(o.FiniteGP(of, x, 0.05), y)
    ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
    try:
        ctx.set_param("lookahead_depth", depth)
        ctx.set_param("multi_check", 7)
        f = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(1.3), ctx=ctx)
        for _ in range(2):
            post = agp.posterior(f(agp.RowVecs(x), 0.05), y)
            assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10)
            assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
        assert _stats(rccl_env)[2] == 0
    finally:
        ctx.close()


def test_device_trace_of_a_real_fit_passes_the_schedule_checker(agp, rccl_env, tmp_path):
    """GPMI_TRACE_SCHEDULE on a real (device) fit: the lines come from the same layer as the dry run and must check clean"""
    import sys

    sys.path.insert(0, str(ROOT / "tools"))
    import multi_schedule_check as M

    x, y = o.synth_inputs(1200, 2, 3)
    for comm in ("rccl", "p2p"):
        os.environ["GPMI_COMM"] = comm
        trace = tmp_path / f"trace_{comm}.jsonl"
        os.environ["GPMI_TRACE_SCHEDULE"] = str(trace)
        try:
            ctx = agp.Context(devices=[0] * 4, P=2, Q=2, nb=128)
            try:
                agp.posterior(agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(x), 0.05), y)
            finally:
                ctx.close()
        finally:
            os.environ.pop("GPMI_TRACE_SCHEDULE", None)
        hdr, problems, rs = M.check_trace(trace)
        assert hdr["dry"] == 0 and hdr["comm"] == (1 if comm == "rccl" else 2)
        assert not problems and not rs, (comm, problems[:3], rs[:3])


_REAL_RCCL_SNIPPET = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
os.environ.pop("GPMI_RCCL_LIB", None)
os.environ["GPMI_COMM"] = "rccl"
import abstractgps_jl_amd as agp
lib = agp._lib.load()
err = ctypes.c_double(-1.0)
rc = lib.gp_rccl_selftest(0, 1 << 20, ctypes.byref(err))
out = {{"selftest_rc": rc, "selftest_err": err.value, "selftest_msg": lib.gp_last_error().decode(errors="replace") if rc else ""}}
if rc == 0:
    # the force_rccl path of gp_ctx_create_multi: ONE real device, the real library, ncclCommInitAll(1), a fit through the driver
    rng = np.random.default_rng(7)
    x = rng.standard_normal((700, 3)); y = np.sin(x.sum(1)) + 0.1 * rng.standard_normal(700)
    one = agp.Context(0)
    ref = agp.posterior(agp.GP(agp.SqExponentialKernel(), ctx=one)(agp.RowVecs(x), 0.01), y)
    ctx = agp.Context(devices=[0], nb=128)
    out["comm"] = ctx.multi_info()["comm"]
    post = agp.posterior(agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(x), 0.01), y)
    out["logpdf_rel"] = abs(float(post.logpdf_value) - float(ref.logpdf_value)) / abs(float(ref.logpdf_value))
    out["alpha_rel"] = float(np.linalg.norm(post.data.alpha - ref.data.alpha) / np.linalg.norm(ref.data.alpha))
    out["stats"] = ctx.multi_stats()
print("RESULT " + json.dumps(out))
"""


def test_real_librccl_loads_and_moves_doubles_on_one_device(agp):
    """The REAL librccl.so of the box (no stand-in): dlopen, the seven entry points, ncclCommInitAll(1), one grouped
    ncclSend / ncclRecv pair of 2^20 doubles (every element compared: the ncclFloat64 constant), then a fit on a one-device
    multi ctx created with GPMI_COMM=rccl (the `force_rccl` path of gp_ctx_create_multi).  In a subprocess with a timeout: a
    library that hangs at initialisation costs this test, not the suite.  What it cannot show: transfers between distinct devices."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("GPMI_RCCL_LIB", "GPMI_COMM")}
    r = subprocess.run([sys.executable, "-c", _REAL_RCCL_SNIPPET.format(root=str(ROOT))], env=env, capture_output=True, text=True, timeout=240)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[-1][7:])
    assert out["selftest_rc"] == 0, out
    assert out["selftest_err"] == 0.0, out
    assert out["comm"] == "rccl", out
    assert out["logpdf_rel"] <= 1e-10 and out["alpha_rel"] <= 1e-8, out


def test_stand_in_library_passes_the_same_selftest(agp, rccl_env):
    lib = agp._lib.load()
    err = ctypes.c_double(-1.0)
    assert lib.gp_rccl_selftest(0, 4096, ctypes.byref(err)) == 0, lib.gp_last_error()
    assert err.value == 0.0
