#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "beyond_2_pow_32" --durations=3 > $O/pytest_call6_big.log 2>&1; echo "pytest big rc=$?"; tail -12 $O/pytest_call6_big.log
GPMI_TEST_RANDOM_CASES=1500 GPMI_TEST_RANDOM_SEED=60000 timeout 1200 python -m pytest tests/test_gpu_random.py -m gpu -q -x > $O/random_sweep_extra.log 2>&1; echo "random rc=$?"; tail -3 $O/random_sweep_extra.log
GPMI_TEST_RANDOM_CASES=2400 GPMI_TEST_RANDOM_SEED=61000 timeout 1200 python -m pytest tests/test_gpu_random_vfe.py -m gpu -q -x > $O/random_sweep_vfe.log 2>&1; echo "random vfe rc=$?"; tail -3 $O/random_sweep_vfe.log
GPMI_TEST_RANDOM_CASES=900 GPMI_TEST_RANDOM_SEED=62000 timeout 1200 python -m pytest tests/test_gpu_random_next.py -m gpu -q -x > $O/random_sweep_next.log 2>&1; echo "random next rc=$?"; tail -3 $O/random_sweep_next.log
