#!/bin/bash
# DeepLabV3+ B=16 512x512: eager vs replayed (lane executor) step time.  usage: tools/dl_modes.sh [policies...]
cd "$GRAFT_REPO_ROOT"
for pol in ${@:-fp32 half}; do
  for g in 0 1; do
    echo "graph=$g $(PSEG_PRECISION=$pol PSEG_GRAPH=$g python3 tools/bench_model.py deeplabv3plus 16 512 21 20 2>&1 | grep -a 'ms/step')"
  done
done
