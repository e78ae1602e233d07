#!/bin/bash
# Round 3 measurement artefacts: the bench line as the driver runs it, rocprofv3 kernel stats of the same command, PMC traffic passes
# (FETCH_SIZE / WRITE_SIZE in separate --pmc runs, kernel-trace only), kernel stats of C2 / C3.  Output under gpurun_out/r3/prof.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3/prof
rm -rf $OUT; mkdir -p $OUT
cd $R && timeout 600 python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_c4.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_frac'], {k:(round(v['ms_per_step'],2), round(v.get('frac', v.get('frac_fp32')),3)) for k,v in d.get('other_configs',{}).items()}, d['cpu_baseline']['value'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -o c4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-other-configs > $OUT/stats_c4.log 2>&1; echo "stats c4 rc=$?"
f=$(find $OUT/stats_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_c4_kernel_stats.csv && head -8 $OUT/bench_c4_kernel_stats.csv
rm -rf $OUT/stats_c4
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-other-configs > $OUT/p$i.log 2>&1
  echo "pmc pass $i [$grp] rc=$?"
done
python $R/tools/pmc_summary.py $OUT gemm_nt > $OUT/pmc_bench_summary.json 2> $OUT/pmc_summary.err; cat $OUT/pmc_bench_summary.json | head -20
for cfg in C2 C3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python $R/tools/bench_configs.py $cfg > $OUT/stats_$cfg.log 2>&1; echo "stats $cfg rc=$?"
  f=$(find $OUT/stats_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${cfg}_kernel_stats.csv && head -6 $OUT/${cfg}_kernel_stats.csv
  grep '"config"' $OUT/stats_$cfg.log >> $OUT/configs.jsonl
  rm -rf $OUT/stats_$cfg
done
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/p1 $OUT/p2; du -sh $OUT
