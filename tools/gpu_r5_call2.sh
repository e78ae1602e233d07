#!/bin/bash
# Round 5, GPU call 2: inverse-diagonal-block forward solves (parity + the §8(f) rows with dib_nb = 0 / 2048 / 1024), then the counter evidence
# (tools/gpu_r5_pmc.sh) and rocprofv3 kernel stats of C5 and C2 on the round-5 engine.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu > $OUT/pytest_call2.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -6 $OUT/pytest_call2.log
timeout 300 python tools/r5_next.py 16384 -- dib_nb=0 dib_nb=2048 dib_nb=1024 > $OUT/next_c2.jsonl 2> $OUT/next_c2.err; echo "next c2 rc=$? ($(( $(date +%s) - t0 )) s)"; cut -c1-230 $OUT/next_c2.jsonl
timeout 600 python tools/r5_next.py 65536 -- dib_nb=0 dib_nb=2048 dib_nb=1024 > $OUT/next_c4.jsonl 2> $OUT/next_c4.err; echo "next c4 rc=$? ($(( $(date +%s) - t0 )) s)"; cut -c1-230 $OUT/next_c4.jsonl
bash tools/gpu_r5_pmc.sh; echo "pmc done ($(( $(date +%s) - t0 )) s)"
cd /tmp
for cfg in "C5 c5_profile.py reps=4" "C2 trace_fit.py 16384" "C3 trace_fit.py 32768"; do
  set -- $cfg; tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$tag -o t -- python $R/tools/"$@" > $OUT/stats_$tag.log 2>&1; echo "stats $tag rc=$?"
  f=$(find $OUT/st_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv && head -7 $OUT/${tag}_kernel_stats.csv | cut -c1-160
  tail -2 $OUT/stats_$tag.log | cut -c1-300
  rm -rf $OUT/st_$tag
done
echo "all done ($(( $(date +%s) - t0 )) s)"
