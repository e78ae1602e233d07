#!/bin/bash
# Kernel timelines of the mid-size fits (rocprofv3 --kernel-trace) + the CU-mask probe.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
$R/tools/bin/cumask_probe > $OUT/cumask_probe.txt 2>&1; echo "probe rc=$?"; head -50 $OUT/cumask_probe.txt
cd /tmp
tr() { tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o t -- python $R/tools/trace_fit.py "$@" > $OUT/$tag.log 2>&1
  echo "trace $tag rc=$?"; grep "^fit" $OUT/$tag.log
  f=$(find $OUT/$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${tag}_trace.csv && python $R/tools/trace_analyze.py $f 30 > $OUT/${tag}_summary.txt 2>&1
  rm -rf $OUT/$tag
  head -30 $OUT/${tag}_summary.txt
}
tr c2_la0 16384 lookahead=0
tr c2_la1 16384 lookahead=1
tr c2_la1_nb1024 16384 lookahead=1 nb=1024
tr c3_la1 32768 lookahead=1
tr m4096 4096 lookahead=0
du -sh $OUT
