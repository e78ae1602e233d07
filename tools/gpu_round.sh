#!/bin/bash
# One gpurun call: smoke, GPU tests, timing sweep, bench, rocprof kernel trace.  Everything is logged
# under gpurun_out/ (merged back).  Each stage has its own timeout so a hang cannot strike the box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${STAGES:-smoke tests perf bench prof}"
for st in $STAGES; do
  case $st in
    smoke) timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" ;;
    tests) timeout 1200 python -m pytest tests -m gpu -q --no-header -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/pytest_gpu.log ;;
    perf)  timeout 900 python tools/gpu_perf.py ${PERF_ARGS:-} > gpurun_out/perf.log 2>&1; echo "perf rc=$?"; tail -30 gpurun_out/perf.log ;;
    bench) timeout 1200 python bench.py ${BENCH_ARGS:---steps 2 --warmup 1} > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log ;;
    gemm)  timeout 600 python tools/gemm_bench.py ${GEMM_ARGS:-} > gpurun_out/gemm.log 2>&1; echo "gemm rc=$?"; cat gpurun_out/gemm.log ;;
    pmc)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVES} -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${PMC_TAG:-a} -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py ${PMC_ARGS:---shapes 16384x16384x2048 --reps 2} > $GRAFT_REPO_ROOT/gpurun_out/pmc_${PMC_TAG:-a}.log 2>&1); echo "pmc rc=$?"; find gpurun_out/pmc_${PMC_TAG:-a} -type f | head ;;
    prof)  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py ${PROF_ARGS:---steps 1 --warmup 1 --no-cpu-baseline --no-check} > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1); echo "prof rc=$?"; find gpurun_out/prof -name "*stats*" | head ;;
  esac
done
