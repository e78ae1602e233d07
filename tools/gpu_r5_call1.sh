#!/bin/bash
# Round 5, GPU call 1: the pipelined GEMM k loop — parity on the fast GPU files (with the new defaults-asserting fixture), then A/B against the
# round-2 loop on pairs (N = 4 096 … 65 536), C5 and isolated launches; kmat store probe; fp32 residency with the new loop.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu > $OUT/pytest_call1.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/pytest_call1.log
timeout 700 python tools/r5_sweep.py pair:4096,16384,32768,65536 c5 -- base gemm_pipe=0 > $OUT/sweep_pipe.jsonl 2> $OUT/sweep_pipe.err; echo "sweep_pipe rc=$? ($(( $(date +%s) - t0 )) s)"; cat $OUT/sweep_pipe.jsonl | cut -c1-260
timeout 300 python tools/r5_sweep.py c5 -- gemm_pad_f32=0 "gemm_pipe=0,gemm_pad_f32=0" base vfe_sk=1 > $OUT/sweep_c5.jsonl 2> $OUT/sweep_c5.err; echo "sweep_c5 rc=$?"; cat $OUT/sweep_c5.jsonl | cut -c1-200
timeout 200 python tools/r5_sweep.py pair:65536 -- kmat_nt=1 base > $OUT/sweep_kmat.jsonl 2> $OUT/sweep_kmat.err; echo "sweep_kmat rc=$?"; cat $OUT/sweep_kmat.jsonl | cut -c1-260
timeout 120 tools/bin/kmat_probe 32768 > $OUT/kmat_probe.jsonl 2>&1; echo "kmat_probe rc=$?"; cat $OUT/kmat_probe.jsonl
timeout 400 python tools/r5_sweep.py gemm -- base gemm_pipe=0 "gemm_pad_lds=20480" > $OUT/sweep_gemm.jsonl 2> $OUT/sweep_gemm.err; echo "sweep_gemm rc=$? ($(( $(date +%s) - t0 )) s)"; cat $OUT/sweep_gemm.jsonl | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not c4" > $OUT/pytest_call1b.log 2>&1; echo "pytest b rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/pytest_call1b.log
