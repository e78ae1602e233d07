#!/bin/bash
# Round 4 measurement artefacts: the bench line as the driver runs it, rocprofv3 kernel stats of the same command, PMC traffic passes
# (FETCH_SIZE / WRITE_SIZE in separate --pmc runs, kernel-trace only), kernel stats + timelines of C2 / N = 4096.  Output under gpurun_out/r4/prof.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4/prof
rm -rf $OUT; mkdir -p $OUT/traces
cd $R && timeout 700 python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_c4.json').read().strip().splitlines()[-1]); rf=d['roofline']; print(d['ms_per_step'], rf['frac'], rf['kernel_frac'], rf.get('kernel_frac_lookahead_off'), {k:(round(v['ms_per_step'],2), round(v.get('frac', v.get('frac_fp32')),3)) for k,v in d.get('other_configs',{}).items() if k != 'next'}, d['cpu_baseline']['value']); print(json.dumps(d['other_configs']['next'])[:1200])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -o c4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-other-configs --no-comparator > $OUT/stats_c4.log 2>&1; echo "stats c4 rc=$?"
f=$(find $OUT/stats_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_c4_kernel_stats.csv && head -8 $OUT/bench_c4_kernel_stats.csv | cut -c1-200
rm -rf $OUT/stats_c4
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-other-configs --no-comparator > $OUT/p$i.log 2>&1
  echo "pmc pass $i [$grp] rc=$?"
done
python $R/tools/pmc_summary.py $OUT gemm_nt > $OUT/pmc_bench_summary.json 2> $OUT/pmc_summary.err; cat $OUT/pmc_bench_summary.json | head -20
tr() { tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python $R/tools/trace_fit.py "$@" > $OUT/$tag.log 2>&1
  echo "trace $tag rc=$?"; grep "^fit" $OUT/$tag.log | tail -1
  f=$(find $OUT/$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $R/tools/trace_analyze.py $f 24 > $OUT/traces/${tag}_summary.txt 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/$tag
  head -14 $OUT/traces/${tag}_summary.txt | cut -c1-150
}
tr c2_la1 16384 lookahead=1
tr c2_la0 16384 lookahead=0
tr c3_la1 32768 lookahead=1
tr m4096 4096 lookahead=0
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/p1 $OUT/p2; du -sh $OUT
