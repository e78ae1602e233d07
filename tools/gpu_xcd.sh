#!/bin/bash
# XCD-aware super-tile order of the GEMM grids: time (C4 pair) and fabric traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/xcd
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for x in 0 1; do
  for m in 256 4096; do
    [ $x = 0 ] && [ $m = 4096 ] && continue
    GPMI_PARAMS="xcd_swizzle=$x,xcd_min_tiles=$m" timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd=$x min=$m', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_frac'])"
  done
done
for x in 0 1; do
  for cn in FETCH_SIZE WRITE_SIZE; do
    GPMI_PARAMS="xcd_swizzle=$x" timeout 600 rocprofv3 --kernel-trace --pmc $cn -d $OUT/pmc_x${x}_$cn -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $OUT/pmc_x${x}_$cn.log 2>&1
    echo "pmc xcd=$x $cn rc=$?"
  done
done
python $R/tools/pmc_summary2.py $OUT > $OUT/pmc_xcd_summary.json 2> $OUT/pmc_xcd_summary.err
python - <<PY
import json
d=json.load(open("$OUT/pmc_xcd_summary.json"))
for tag,v in d.items():
    for k,c in v.items():
        if 'gemm_nt_dma' in k:
            print(tag, k[:30], c['dispatches'], {n:round(x['avg']/1024/1024,3) for n,x in c.items() if isinstance(x,dict)}, 'GiB avg per launch (raw counter KiB->GiB)')
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete; du -sh $OUT
