#!/bin/bash
# Round 4, call 1: the new parity pins (C4 / C3-ARD values vs the oracle in the suite, generator-schema fields, real librccl on one
# device, the update self-check), the C4 oracle digest, and the cycle stamps of the current leaf as the baseline of the leaf work.
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/call1_dev.log 2>&1
timeout 300 python -m pytest tests/test_julia_golden.py tests/test_gpu_multi_rccl.py -q -m gpu --timeout 280 -k "generator_schema or real_librccl or stand_in or julia" > $O/pytest_call1a.log 2>&1; echo "a rc=$?"; tail -4 $O/pytest_call1a.log | cut -c1-400
timeout 200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_api.py -q -m gpu --timeout 180 -k "sequential_update or sqmahal_logdetcov or logdet" > $O/pytest_call1b.log 2>&1; echo "b rc=$?"; tail -4 $O/pytest_call1b.log | cut -c1-400
tools/bin/stamps 16384 > $O/stamps_base.log 2>&1; tools/bin/stamps 128 >> $O/stamps_base.log 2>&1; tail -6 $O/stamps_base.log
GPMI_WRITE_C4_DIGEST=$O/c4_oracle_digest.npz timeout 420 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout 400 -k "c3_ard_full_size_values or c4_full_size_values" > $O/pytest_call1c.log 2>&1; echo "c rc=$?"; tail -4 $O/pytest_call1c.log | cut -c1-400
ls -la $O
