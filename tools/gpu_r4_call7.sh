#!/bin/bash
# Round 4, call 7: the whole GPU suite on the 128-column leaf + bench line + rocprofv3 stats / PMC / timelines (final artefacts of the round)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1300 python -m pytest tests -q -m gpu --timeout 600 --durations=12 > $O/pytest_gpu_final.log 2>&1; echo "suite rc=$?"; tail -22 $O/pytest_gpu_final.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke_final.log
bash tools/gpu_r4_prof.sh
