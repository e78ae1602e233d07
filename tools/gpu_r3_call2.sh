#!/bin/bash
# Round 3, GPU call 2: discriminating variants of the fresh-context first-fit failure (5/32 in call 1 with stream-K + high-priority
# comm stream, 0/32 as soon as marker kernels sit around the events), and a kernel timeline of the CU-partitioned C2 fit.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
S="timeout 300 python tools/multi_fresh_stress.py"
$S 10 dims=1 check=0                 > $O/ff_d1_check0.log 2>&1; tail -1 $O/ff_d1_check0.log
$S 10 dims=1 check=4                 > $O/ff_d1_check4_nan.log 2>&1; tail -1 $O/ff_d1_check4_nan.log
$S 10 dims=1 check=2                 > $O/ff_d1_check2_verify.log 2>&1; tail -1 $O/ff_d1_check2_verify.log
$S 10 dims=1 check=0 copy_kernel=1   > $O/ff_d1_copykernel.log 2>&1; tail -1 $O/ff_d1_copykernel.log
$S 10 dims=1 check=0 sk=0 prio=1     > $O/ff_d1_sk0_prio1.log 2>&1; tail -1 $O/ff_d1_sk0_prio1.log
$S 10 dims=1 check=0 sk=1 prio=0     > $O/ff_d1_sk1_prio0.log 2>&1; tail -1 $O/ff_d1_sk1_prio0.log
$S 10 dims=1 check=0 sk=0 prio=0     > $O/ff_d1_sk0_prio0.log 2>&1; tail -1 $O/ff_d1_sk0_prio0.log
$S 10 dims=1 check=0 dsync=1         > $O/ff_d1_dsync1_exchange.log 2>&1; tail -1 $O/ff_d1_dsync1_exchange.log
$S 10 dims=1 check=0 dsync=8         > $O/ff_d1_dsync8_panel.log 2>&1; tail -1 $O/ff_d1_dsync8_panel.log
$S 10 dims=1 check=0 dsync=16        > $O/ff_d1_dsync16_bulk.log 2>&1; tail -1 $O/ff_d1_dsync16_bulk.log
$S 10 dims=1 check=0 dsync=4         > $O/ff_d1_dsync4_hostwait.log 2>&1; tail -1 $O/ff_d1_dsync4_hostwait.log
$S 10 dims=1 check=0 hwq=4           > $O/ff_d1_hwq4.log 2>&1; tail -1 $O/ff_d1_hwq4.log
$S 10 dims=1 check=0 hwq=32          > $O/ff_d1_hwq32.log 2>&1; tail -1 $O/ff_d1_hwq32.log
$S 10 dims=3 check=0                 > $O/ff_d3_check0.log 2>&1; tail -1 $O/ff_d3_check0.log
$S 10 dims=1 check=0 grids=2x2,4x1   > $O/ff_d1_smallgrids.log 2>&1; tail -1 $O/ff_d1_smallgrids.log
# kernel timelines of the C2 fit: unpartitioned nb=512, and CU-partitioned 32 / 64
R=$GRAFT_REPO_ROOT
cd /tmp
tr() { tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/$tag -o t -- python $R/tools/trace_fit.py "$@" > $R/$O/$tag.log 2>&1
  echo "trace $tag rc=$?"; grep "^fit" $R/$O/$tag.log
  f=$(find $R/$O/$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp $f $R/$O/${tag}_trace.csv && python $R/tools/trace_analyze.py $f 30 > $R/$O/${tag}_summary.txt 2>&1
  rm -rf $R/$O/$tag
  head -14 $R/$O/${tag}_summary.txt; grep "in flight" $R/$O/${tag}_summary.txt
}
tr c2_nb512 16384 nb=512
tr c2_split32 16384 cu_split=32 cu_split_nb=512 cu_split_tail=8192
tr c2_split64 16384 cu_split=64 cu_split_nb=512 cu_split_tail=4096
