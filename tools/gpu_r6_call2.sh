#!/bin/bash
# round 6, call 2: (a) the gradient against finite differences component by component up to C4 (the bench's new check failed there);
# (b) the multi-device tests on the tree with the inverse-block rows-below solve and the masked chain stream; (c) C4 through the driver on virtual ranks:
# P×1 and 2-D grids, solve variants
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/grad_check.py 16384 32768 49152 65536 > $O/grad_check.jsonl 2> $O/grad_check.err; echo "grad_check rc=$?"
cat $O/grad_check.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x -k "multi" --durations=8 > $O/pytest_call2_multi.log 2>&1; echo "pytest multi rc=$?"
tail -15 $O/pytest_call2_multi.log
VRANKS=1 OUT=$O/multi_virtual_bench.jsonl timeout 1500 bash tools/scale_sweep.sh 3 1 2> $O/multi_virtual_bench.err | tee $O/multi_virtual_bench.txt
tail -5 $O/multi_virtual_bench.err
