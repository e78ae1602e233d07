#!/bin/bash
# End-of-round GPU call (ROUND=r6 bash tools/gpu_final.sh): the whole GPU suite as the driver runs it, smoke(), then the measurement artefacts of the final tree:
#   counter passes over `bench.py --steps 1` (FETCH_SIZE / WRITE_SIZE / SQ + GRBM, separate --pmc runs, kernel-trace only) -> pmc_bench_summary.json
#   (copied into profiles/$ROUND/ on the box so that the bench line replays THIS round's counters), the bench line as the driver runs it, rocprofv3
#   --kernel-trace --stats of the same command, kernel tables of C5 / C2 / C3 and their SQ counters (tools/gpu_pmc_configs.sh).
set -u
ROUND=${ROUND:-r6}
export ROUND
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$ROUND/final; rm -rf $OUT; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_final.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke_final.log
cd /tmp
BARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-check --no-other-configs --no-comparator"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_fetch -o pmc --output-format csv -- python $R/bench.py $BARGS > $OUT/p_fetch.log 2>&1; echo "pmc FETCH rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_write -o pmc --output-format csv -- python $R/bench.py $BARGS > $OUT/p_write.log 2>&1; echo "pmc WRITE rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OUT/p_sq -o pmc --output-format csv -- python $R/bench.py $BARGS > $OUT/p_sq.log 2>&1; echo "pmc SQ rc=$? ($(( $(date +%s) - t0 )) s)"
python $R/tools/pmc_bench_summary.py $OUT > $OUT/pmc_bench_summary.json 2> $OUT/pmc_bench_summary.err; head -40 $OUT/pmc_bench_summary.json
mkdir -p $R/profiles/$ROUND && [ -s $OUT/pmc_bench_summary.json ] && grep -q FETCH_SIZE $OUT/pmc_bench_summary.json && cp $OUT/pmc_bench_summary.json $R/profiles/$ROUND/pmc_bench_summary.json
rm -rf $OUT/p_fetch $OUT/p_write $OUT/p_sq
cd $R && timeout 900 python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4_progress.log; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; python -c "
import json; d=json.loads(open('$OUT/bench_c4.json').read().strip().splitlines()[-1]); rf=d['roofline']; print(d['ms_per_step'], rf['frac'], rf['kernel_frac'], rf.get('mfma_busy'), rf.get('traffic'), {k:(round(v['ms_per_step'],2), round(v.get('frac', v.get('frac_fp32')),3)) for k,v in d.get('other_configs',{}).items() if k != 'next'}, d['cpu_baseline']['value'], d.get('check_vs_oracle_digest')); nx=d['other_configs']['next']; print({k:(round(v['ms'],1), round(v['frac'],3)) for k,v in nx.items() if k!='value_and_gradient'}, {k:(round(v['ms'],1), round(v['frac'],3)) for k,v in nx['value_and_gradient'].items()})"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -o c4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-other-configs --no-comparator > $OUT/stats_c4.log 2>&1; echo "stats c4 rc=$?"
f=$(find $OUT/stats_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_c4_kernel_stats.csv && head -6 $OUT/bench_c4_kernel_stats.csv | cut -c1-200
tail -1 $OUT/stats_c4.log | cut -c1-400 > $OUT/bench_c4_stats_run_line.json; rm -rf $OUT/stats_c4
for cfg in "C5 c5_profile.py reps=4" "C2 trace_fit.py 16384" "C3 trace_fit.py 32768"; do
  set -- $cfg; tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$tag -o t -- python $R/tools/"$@" > $OUT/stats_$tag.log 2>&1; echo "stats $tag rc=$?"
  f=$(find $OUT/st_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/st_$tag
done
bash $R/tools/gpu_pmc_configs.sh > $OUT/pmc_configs.log 2>&1; cp $R/gpurun_out/$ROUND/pmc/pmc_sq_summary.json $OUT/pmc_sq_summary.json; cp $R/gpurun_out/$ROUND/pmc/pmc_sq_table.txt $OUT/pmc_sq_table.txt; grep "gemm_nt\|mfma_rate\|panel" $OUT/pmc_sq_table.txt | cut -c1-200
find $OUT -name "*.csv" -size +2M -delete; du -sh $OUT; echo "all done ($(( $(date +%s) - t0 )) s)"
