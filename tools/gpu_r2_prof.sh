#!/bin/bash
# Round 2 profiling: rocprofv3 kernel-trace stats for the bench (C4) and for C2 / C3 / C5, then separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / MFMA busy) over the bench and over C5.  Summaries are copied to profiles/r2/ by hand.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
stats() { # tag, command...
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o p -- "$@" > $OUT/$tag.log 2>&1
  echo "stats $tag rc=$?"
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv && python $R/tools/kstats.py $f 14
  rm -rf $OUT/$tag
}
stats bench_c4 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check
stats C2 python $R/tools/bench_configs.py C2
stats C3 python $R/tools/bench_configs.py C3
stats C5 python $R/tools/sweep_r2.py C5only16
pmc() { # tag, pattern, counters, command...
  tag=$1; pat=$2; grp=$3; shift 3
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$tag -o pmc --output-format csv -- "$@" > $OUT/pmc_$tag.log 2>&1
  echo "pmc $tag [$grp] rc=$?"
}
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  t=$(echo $g | cut -d' ' -f1)
  pmc bench_$t x "$g" python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check
  pmc c5_$t x "$g" python $R/tools/sweep_r2.py C5only16
  pmc c2_$t x "$g" python $R/tools/bench_configs.py C2
done
python $R/tools/pmc_summary2.py $OUT > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err; head -c 3000 $OUT/pmc_summary.json
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
