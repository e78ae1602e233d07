#!/bin/bash
# Round 5, GPU call 4: the rank-4 pivot chain of the register-resident leaf (leaf_rank4) — values against the round-3 leaf, cycle stamps, the potrf unit
# tests and the parity suite on it, pairs with and without it; fp32 residency default re-check.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
t0=$(date +%s)
for v in 0 1; do
  timeout 120 tools/bin/leaf_check_r4$v 0 64 208 16384 > $OUT/leaf_check_r4$v.log 2>&1; echo "leaf_check r4=$v rc=$?"; grep -iE "128-col|max|fail|us" $OUT/leaf_check_r4$v.log | tail -14 | cut -c1-220
  timeout 60 tools/bin/leaf_stamps_r4$v 16384 0 64 8 > $OUT/leaf_stamps_r4$v.log 2>&1; echo "leaf_stamps r4=$v rc=$?"; tail -6 $OUT/leaf_stamps_r4$v.log | cut -c1-250
done
GPMI_PARAMS=leaf_rank4=1 GPMI_TEST_EXPECT=leaf_rank4=1 timeout 600 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu > $OUT/pytest_call4_rank4.log 2>&1; echo "pytest rank4 rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/pytest_call4_rank4.log
timeout 400 python tools/r5_sweep.py pair:4096,8192,16384,32768,65536 -- base leaf_rank4=1 base leaf_rank4=1 > $OUT/sweep4.jsonl 2> $OUT/sweep4.err; echo "sweep rc=$? ($(( $(date +%s) - t0 )) s)"; cut -c1-200 $OUT/sweep4.jsonl
timeout 200 python tools/r5_sweep.py c5 -- base gemm_pad_f32=20480 leaf_rank4=1 > $OUT/sweep4_c5.jsonl 2> $OUT/sweep4_c5.err; echo "sweep c5 rc=$?"; cut -c1-200 $OUT/sweep4_c5.jsonl
echo "all done ($(( $(date +%s) - t0 )) s)"
