#!/bin/bash
# Round 5, GPU call 3: gradient with the triangular k range, fast exp in the Gram kernels, persistent rank threads (multi-device suites), fp32 GEMM
# residency A/B, over-fetch beside a copy stream, CU-masked chain probe, virtual-rank bench on the shipped rank configuration + its kernel stats.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_random.py tests/test_gpu_multi.py tests/test_gpu_multi_rccl.py -x -q -m gpu > $OUT/pytest_call3.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -6 $OUT/pytest_call3.log
timeout 300 python tools/r5_next.py 16384 mv cov grad -- base > $OUT/next3_c2.jsonl 2> $OUT/next3_c2.err; echo "next c2 rc=$?"; cut -c1-200 $OUT/next3_c2.jsonl
timeout 400 python tools/r5_next.py 65536 grad -- base > $OUT/next3_c4.jsonl 2> $OUT/next3_c4.err; echo "next c4 rc=$? ($(( $(date +%s) - t0 )) s)"; cut -c1-200 $OUT/next3_c4.jsonl
timeout 300 python tools/r5_sweep.py c5 pair:16384,65536 -- base gemm_pad_f32=0 base gemm_pad_f32=0 > $OUT/sweep3.jsonl 2> $OUT/sweep3.err; echo "sweep rc=$? ($(( $(date +%s) - t0 )) s)"; cut -c1-230 $OUT/sweep3.jsonl
timeout 120 tools/bin/overfetch_probe > $OUT/overfetch_probe.jsonl 2>&1; echo "overfetch rc=$?"; cut -c1-420 $OUT/overfetch_probe.jsonl
timeout 300 python tools/cumask_chain_probe.py > $OUT/cumask_chain_probe.jsonl 2> $OUT/cumask_chain_probe.err; echo "cumask rc=$? ($(( $(date +%s) - t0 )) s)"; cut -c1-230 $OUT/cumask_chain_probe.jsonl; tail -3 $OUT/cumask_chain_probe.err
for v in 0 1 2 4 8; do
  if [ $v = 0 ]; then a="--gpus 1"; else a="--vranks $v"; fi
  timeout 400 python bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-other-configs --no-comparator > $OUT/vbench_$v.json 2> $OUT/vbench_$v.err; echo "vbench $v rc=$? ($(( $(date +%s) - t0 )) s)"
  python -c "
import json; d=json.loads(open('$OUT/vbench_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['config']['parallelism'], d.get('multi_stats'), d['roofline']['kernel_frac'], d['roofline']['launches_per_step'])"
done
cd /tmp
for v in 1 8; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_v$v -o t -- python $R/bench.py --vranks $v --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-other-configs --no-comparator > $OUT/vstats_$v.log 2>&1; echo "vstats $v rc=$?"
  f=$(find $OUT/st_v$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/vranks${v}_kernel_stats.csv && head -9 $OUT/vranks${v}_kernel_stats.csv | cut -c1-150
  rm -rf $OUT/st_v$v
done
echo "all done ($(( $(date +%s) - t0 )) s)"
