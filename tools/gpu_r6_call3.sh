#!/bin/bash
# round 6, call 3: the sparse path after "vfe_dual" (chunk GEMMs on two streams) and "vfe_inv_nb" (batched inverse blocks in the prelude):
# its tests, the in-process A/B at C5 with the three-way phase split, and a kernel table of the new default
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/grad_check.py 65536 > $O/grad_check_after_fix.jsonl 2> $O/grad_check_after_fix.err; echo "grad_check rc=$?"; cat $O/grad_check_after_fix.jsonl | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c4_gradient" > $O/pytest_call3_c4grad.log 2>&1; echo "pytest c4 gradient rc=$?"; tail -3 $O/pytest_call3_c4grad.log
timeout 900 python -m pytest tests -q -m gpu -x -k "vfe or c5 or sparse or approx or elbo or dtc" --durations=5 > $O/pytest_call3_vfe.log 2>&1; echo "pytest vfe rc=$?"
tail -8 $O/pytest_call3_vfe.log
timeout 600 python tools/c5_ab.py rounds=3 > $O/c5_ab.jsonl 2> $O/c5_ab.err; echo "c5_ab rc=$?"
cat $O/c5_ab.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_C5 -o t -- python $GRAFT_REPO_ROOT/tools/c5_profile.py reps=4 > $GRAFT_REPO_ROOT/$O/stats_C5.log 2>&1; echo "stats C5 rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/st_C5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/C5_kernel_stats.csv; rm -rf $O/st_C5
tail -2 $O/stats_C5.log | cut -c1-600
head -12 $O/C5_kernel_stats.csv | cut -c1-160
