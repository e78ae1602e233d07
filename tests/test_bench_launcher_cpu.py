"""CPU (gloo, world_size 2 and 3): the process-level protocol of a multi-GPU bench run.  The N devices of a node are driven by ONE
process — the reference's caller is one Julia process — so under the driver's launcher (torch.distributed.run, N ranks) rank 0 is
the driver and the other ranks only take part in the barriers and in the max-over-ranks reduction.  `bench.py --dry-launcher` runs
exactly that protocol with a stub step and no device; the schedule the driver rank executes inside the library is covered by
tests/test_multi_schedule.py (the rank threads run without a device there)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_launcher_protocol_one_line_from_rank0(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--dry-launcher"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["world"] == world and rec["n_gpus"] == world and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["ms_per_step"] >= 10.0                    # three 10 ms stub steps bracketed by the barriers, maximum over ranks
    assert rec["value"] == pytest.approx(65536 / (rec["ms_per_step"] * 1e-3), rel=1e-9)


def test_dry_launcher_single_process():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-launcher", "--steps", "2", "--warmup", "0"], capture_output=True, text=True,
                       timeout=120, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["world"] == 1


def test_comparator_is_bounded_and_never_takes_the_line_down(monkeypatch):
    """The rocSOLVER yardstick runs in a child process with a hard time limit (one GPU box of round 4 needed more than five minutes to
    load librocsolver.so and the whole bench timed out in it).  Without a device the child answers with an error object; with a child that
    hangs, the parent reports the skip — in both cases a dict that goes into the JSON line."""
    import importlib.util
    import time

    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.rocsolver_comparator(timeout_s=120.0)
    assert isinstance(out, dict) and ("error" in out or "skipped" in out or any(k.startswith("N") for k in out)), out
    # a child that never answers: sys.executable is replaced by a sleeper
    sleeper = ROOT / "tests" / "_sleeper.py"
    sleeper.write_text("import time\ntime.sleep(60)\n")
    try:
        monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(sleeper))
        t0 = time.perf_counter()
        out = bench.rocsolver_comparator(timeout_s=2.0)
        assert "skipped" in out and time.perf_counter() - t0 < 20.0
    finally:
        sleeper.unlink()
