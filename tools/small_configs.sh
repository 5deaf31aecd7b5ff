#!/bin/bash
# the launch-bound configurations, replayed (and eager): usage tools/small_configs.sh [policies...]
cd "$GRAFT_REPO_ROOT"
POLS=${@:-fp32 half}
for cfg in "hrnet 8 512 21 20" "unet 8 256 2 30"; do
  for pol in $POLS; do
    for g in 0 1; do
      echo "graph=$g $(PSEG_PRECISION=$pol PSEG_GRAPH=$g python3 tools/bench_model.py $cfg 2>&1 | grep -a 'ms/step\|lane executor')"
    done
  done
done
