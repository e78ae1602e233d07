#!/bin/bash
# round 6, call 4: C5 with the priority-staggered chunk GEMMs (Y on a high-priority third stream, SYRK on the main stream) and the host marshalling behind the
# prelude: sparse-path tests, the in-process A/B, a kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "vfe or c5 or sparse or approx or elbo or dtc" > $O/pytest_call4_vfe.log 2>&1; echo "pytest vfe rc=$?"
tail -4 $O/pytest_call4_vfe.log
timeout 600 python tools/c5_ab.py rounds=3 > $O/c5_ab2.jsonl 2> $O/c5_ab2.err; echo "c5_ab rc=$?"
cat $O/c5_ab2.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_C5 -o t -- python $GRAFT_REPO_ROOT/tools/c5_profile.py reps=4 > $GRAFT_REPO_ROOT/$O/stats_C5.log 2>&1; echo "stats C5 rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/st_C5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/C5_kernel_stats.csv; rm -rf $O/st_C5
tail -1 $O/stats_C5.log | cut -c1-700
head -6 $O/C5_kernel_stats.csv | cut -c1-160
