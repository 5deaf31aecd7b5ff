#!/bin/bash
# launch-bound configurations, replayed: fewer launches by (a) one slab reduction per step, (b) the fused BatchNorm kernels
cd "$GRAFT_REPO_ROOT"
run() { echo "$1: $(env $2 PSEG_GRAPH=1 python3 tools/bench_model.py $3 2>&1 | grep -a 'ms/step\|lane exec' | tr '\n' ' ' | cut -c1-200)"; }
for pol in half fp32; do
  for cfg in "hrnet 8 512 21 20" "unet 8 256 2 30"; do
    run "base        $pol" "PSEG_PRECISION=$pol" "$cfg"
    run "defer slabs $pol" "PSEG_PRECISION=$pol PSEG_DEFER_SLABS=1" "$cfg"
    run "bn fused    $pol" "PSEG_PRECISION=$pol PSEG_BN_SMALL_GRAPH=1" "$cfg"
    run "both        $pol" "PSEG_PRECISION=$pol PSEG_DEFER_SLABS=1 PSEG_BN_SMALL_GRAPH=1" "$cfg"
  done
done
