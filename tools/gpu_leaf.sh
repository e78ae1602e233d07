#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/leaf
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 900 python tools/sweep_leaf.py > $OUT/sweep_leaf.jsonl 2> $OUT/sweep_leaf.err; echo "sweep rc=$?"; cat $OUT/sweep_leaf.jsonl; tail -3 $OUT/sweep_leaf.err
