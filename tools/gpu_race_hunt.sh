#!/bin/bash
# Next-round localisation of the wrong-first-fit item: every variant in its own bounded call.
set -u
R=${GRAFT_REPO_ROOT:-.}
cd $R; mkdir -p gpurun_out
for v in 1 4 5 6 7 8 9 0; do
  timeout 90 python tools/multi_first_fit.py 10 $v 2>&1 | tail -1
done | tee gpurun_out/race_hunt.txt
