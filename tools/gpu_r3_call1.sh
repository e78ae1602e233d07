#!/bin/bash
# Round 3, GPU call 1: the new multi-device diagnostics + RCCL stand-in tests, the fresh-context stress, the new API tests, and the
# first CU-partition sweep.  Everything under its own timeout; logs under gpurun_out/r3/.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 300 python tools/sweep_r3.py C2 C3 > $O/sweep_cusplit.jsonl 2> $O/sweep_cusplit.err; echo "sweep rc=$?"
timeout 400 python tools/multi_fresh_stress.py 8 sk=1 prio=1 check=0 > $O/fresh_sk1_prio1_check0.log 2>&1; echo "fresh0 rc=$?"
timeout 400 python tools/multi_fresh_stress.py 8 sk=1 prio=1 check=7 > $O/fresh_sk1_prio1_check7.log 2>&1; echo "fresh7 rc=$?"
timeout 400 python tools/multi_fresh_stress.py 8 sk=1 prio=1 check=1 > $O/fresh_sk1_prio1_check1.log 2>&1; echo "fresh1 rc=$?"
timeout 400 python tools/multi_fresh_stress.py 6 sk=1 prio=1 check=7 comm=rccl > $O/fresh_rccl_check7.log 2>&1; echo "freshrccl rc=$?"
timeout 900 python -m pytest tests/test_gpu_multi_rccl.py tests/test_gpu_multi.py -q -k "not concurrent" --timeout 300 > $O/pytest_multi.log 2>&1; echo "pytest multi rc=$?"
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -q --timeout 300 > $O/pytest_api.log 2>&1; echo "pytest api rc=$?"
tail -3 $O/*.log | tail -60
