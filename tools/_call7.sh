#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
python -c "import hashlib;print('lib', hashlib.sha256(open('abstractgps.jl_amd/csrc/libgpmi355.so','rb').read()).hexdigest()[:16])"
timeout 900 python -m pytest tests -q -m gpu -x -k "vfe or c5 or sparse or approx or elbo or dtc" > $O/pytest_call7_vfe.log 2>&1; echo "pytest vfe rc=$?"; tail -3 $O/pytest_call7_vfe.log
timeout 600 python tools/c5_ab.py rounds=2 > $O/c5_ab3.jsonl 2> $O/c5_ab3.err; echo "c5_ab rc=$?"; cat $O/c5_ab3.jsonl
