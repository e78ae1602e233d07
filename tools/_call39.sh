#!/bin/bash
cd $GRAFT_REPO_ROOT
for ch in 16384 65536 131072 262144; do
  echo "vfe_chunk=$ch"; GPMI_PARAMS=vfe_chunk=$ch timeout 300 python tools/train_sparse_example.py n=2000000 m=256 iters=30 2>/dev/null | cut -c1-260
done
GPMI_PARAMS=vfe_chunk=131072 timeout 300 python tools/train_sparse_example.py n=2000000 m=64 iters=30 2>/dev/null | cut -c1-260
timeout 300 python tools/train_sparse_example.py n=2000000 m=64 iters=30 2>/dev/null | cut -c1-260
