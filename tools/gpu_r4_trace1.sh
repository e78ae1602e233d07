#!/bin/bash
# one kernel trace of a fit with given ctx parameters: tools/gpu_r4_trace1.sh <tag> <n> <param=value ...>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
OUT=$R/gpurun_out/r4/traces; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python $R/tools/trace_fit.py "$@" > $OUT/$tag.log 2>&1
echo "trace $tag rc=$?"; grep "^fit" $OUT/$tag.log | tail -1
f=$(find $OUT/$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/trace_analyze.py $f 30 > $OUT/${tag}_summary.txt 2>&1
rm -rf $OUT/$tag
head -34 $OUT/${tag}_summary.txt | cut -c1-150
