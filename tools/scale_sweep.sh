#!/bin/bash
# A/B sweep of the multi-device driver — the first thing to run on an 8-GPU MI355X node (NOT run by any round so far: no multi-GPU box was
# available; the default grid P×1 rests on tools/grid_model.py and on virtual-rank runs on one GPU, profiles/r6/multi_virtual_bench.jsonl).
# One bench line per
#     device count {1, 2, 4, 8} × grid {P×1, the 2-D grids of that count} × transport {rccl, p2p} × diagonal chain {unmasked, 16 masked CUs}
# plus, on the best grid of each count, the substitution solve ("multi_trsm_inv" = 0) and distribution blocks 512 / 2048, in gpurun_out/scale_sweep.jsonl.
#   bash tools/scale_sweep.sh [steps] [warmup]
#   VRANKS=1 bash tools/scale_sweep.sh          the same matrix with VIRTUAL ranks on GPU 0 (one-GPU box: aggregate GPU time, no transport axis)
# Before the sweep, on a multi-GPU node, the multi-device tests over REAL devices (peer copies and real RCCL instead of virtual ranks and the
# stand-in library):    GPMI_TEST_REAL_DEVICES=1 python -m pytest tests -q -m gpu -k "multi or conformance_on_a_multi"
STEPS=${1:-3}; WARM=${2:-1}
OUT=${OUT:-gpurun_out/scale_sweep.jsonl}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
COMMON="--steps $STEPS --warmup $WARM --no-cpu-baseline --no-other-configs --no-comparator"
run() {  # run <gpus> <extra bench args...>
  local n=$1; shift
  if [ -n "$VRANKS" ]; then
    echo "== --vranks $n $*" >&2
    timeout 1200 python bench.py --vranks "$n" $COMMON "$@" | tail -1 >> "$OUT"
  else
    echo "== --gpus $n $*" >&2
    timeout 1200 python bench.py --gpus "$n" $COMMON "$@" | tail -1 >> "$OUT"
  fi
}
python bench.py --gpus 1 $COMMON | tail -1 >> "$OUT"            # the single-device engine on the same box: the reference of every ratio
for N in 1 2 4 8; do
  if [ -z "$VRANKS" ] && [ "$N" -gt "$NGPU" ]; then echo "skip $N devices: only $NGPU visible" >&2; continue; fi
  case $N in 1) GRIDS="1x1";; 2) GRIDS="2x1 1x2";; 4) GRIDS="4x1 2x2";; 8) GRIDS="8x1 4x2 2x4";; esac
  if [ -n "$VRANKS" ] || [ "$N" = 1 ]; then COMMS="p2p"; else COMMS="rccl p2p"; fi
  for GRID in $GRIDS; do
    for COMM in $COMMS; do
      for CUS in 0 16; do
        [ "$N" = 1 ] && [ "$CUS" != 0 ] && continue
        GPMI_COMM=$COMM run $N --grid $GRID --params multi_chain_cus=$CUS
      done
    done
  done
  [ "$N" = 1 ] && continue
  G1=$(echo $GRIDS | cut -d' ' -f1)
  GPMI_COMM=$(echo $COMMS | cut -d' ' -f1) run $N --grid $G1 --params multi_trsm_inv=0
  for NB in 512 2048; do GPMI_COMM=$(echo $COMMS | cut -d' ' -f1) run $N --grid $G1 --nb $NB; done
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for r in rows:
    ms = r.get("multi_stats") or {}
    print(f"{r['ms_per_step']:9.1f} ms  {100 * r['roofline']['frac']:5.1f} % of peak  retries {ms.get('retries', '-')}  {r.get('check_vs_oracle_digest', '-'):4}  {r['config']['parallelism']}")
PY
