#!/bin/bash
# Scaling sweep for an 8-GPU MI355X node (NOT run by any round so far: no multi-GPU box was available — every number this would
# produce is missing from profiles/, and tools/grid_model.py is the only basis for the default grid).  One JSON line per run in
# gpurun_out/scale_sweep.jsonl:  device counts 1/2/4/8 × the grids of each count × distribution blocks 512/1024/2048.
#   bash tools/scale_sweep.sh [steps] [warmup]
# Before the sweep, on such a node: the multi-device tests over REAL devices (peer copies and real RCCL instead of virtual ranks
# and the stand-in library) —   GPMI_TEST_REAL_DEVICES=1 python -m pytest tests -q -m gpu -k "multi or conformance_on_a_multi"
STEPS=${1:-3}; WARM=${2:-1}
OUT=gpurun_out/scale_sweep.jsonl
mkdir -p gpurun_out; : > $OUT
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "skip --gpus $N: only $NGPU visible" >&2; continue; }
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-other-configs | tail -1 >> $OUT
    continue
  fi
  for GRID in $(python - <<PY
n=$N
print(" ".join(f"{p}x{n//p}" for p in range(1, n+1) if n % p == 0))
PY
); do
    for NB in 512 1024 2048; do
      echo "== --gpus $N --grid $GRID --nb $NB" >&2
      timeout 900 python bench.py --gpus $N --grid $GRID --nb $NB --steps $STEPS --warmup $WARM --no-cpu-baseline | tail -1 >> $OUT
    done
  done
done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/scale_sweep.jsonl") if l.startswith("{")]
for r in rows:
    print(r["n_gpus"], r["config"]["parallelism"][:60], f"{r['ms_per_step']:.1f} ms", f"{100 * r['roofline']['frac']:.1f} % of peak")
PY
