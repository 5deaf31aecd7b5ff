#!/bin/bash
# layer-1 / layer-2 shapes of the DeepLabV3+ step (short contractions on big maps) under forced tiles: is the planner's choice the best?
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_sweep_l1.txt
: > $out
S="l1_1x1a l1_3x3 l1_1x1b l1_1x1c l2_1x1a l2_1x1b l2_1x1c l3_1x1c cls_conv stem7x7"
echo "== default plan" >> $out
python tools/bench_conv.py fp32 $S >> $out 2>&1
for t in "128 128" "128 64" "64 128" "128 32"; do
  set -- $t
  echo "== fwd/dgrad tile $1 x $2" >> $out
  PSEG_CONV_BM=$1 PSEG_CONV_BN=$2 python tools/bench_conv.py fp32 $S >> $out 2>&1
done
for t in "128 128" "64 128" "128 64"; do
  set -- $t
  echo "== wgrad tile $1 x $2" >> $out
  PSEG_WGRAD_BM=$1 PSEG_WGRAD_BN=$2 python tools/bench_conv.py fp32 $S >> $out 2>&1
done
for b in 1 2 3; do
  echo "== wgrad blocks per CU $b" >> $out
  PSEG_WGRAD_BPC=$b python tools/bench_conv.py fp32 $S >> $out 2>&1
done
