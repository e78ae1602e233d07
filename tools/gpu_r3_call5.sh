#!/bin/bash
# Round 3, GPU call 5: which runtime mechanism?  First fits of fresh, UNPRIMED multi-device contexts (the configuration that fails at
# ~4 %) under runtime switches: CPU-side dependency resolution off, kernel arguments in host memory, unoptimised cache flushes.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
export GPMI_MULTI_PRIME=0
S="timeout 300 python tools/multi_fresh_stress.py 60 dims=1 check=0 grids=8x1,4x2"
run() { tag=$1; shift; env "$@" $S > $O/ff4_$tag.log 2>&1; echo "$tag: $(tail -1 $O/ff4_$tag.log)"; }
run base              X=1
run cpuwait0          ROC_CPU_WAIT_FOR_SIGNAL=0
run devkernarg0       HIP_FORCE_DEV_KERNARG=0
run optflush0         AMD_OPT_FLUSH=0
run hdpflushwa        DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run activewait0       ROC_ACTIVE_WAIT_TIMEOUT=0
run base2             X=2
grep -h "WRONG\|ERROR" $O/ff4_*.log | cut -c1-260 | head -30
