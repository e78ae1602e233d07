#!/bin/bash
# round 6, call 5: the persistent vector-solve launch ("trsv_persist"): unit test under a hard time limit first (a spinning kernel must not take the box),
# then the parity files that solve vectors, then the in-process A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_units.py -q -x -k "trsv" > $O/pytest_call5_trsv.log 2>&1; rc=$?; echo "pytest trsv rc=$rc"; tail -5 $O/pytest_call5_trsv.log
[ $rc -ne 0 ] && exit 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_random.py tests/test_gpu_random_vfe.py -q -x > $O/pytest_call5_parity.log 2>&1; echo "pytest parity rc=$?"; tail -4 $O/pytest_call5_parity.log
timeout 900 python tools/sweep_ab.py rounds=2 > $O/sweep_ab.jsonl 2> $O/sweep_ab.err; echo "sweep_ab rc=$?"; cat $O/sweep_ab.jsonl
