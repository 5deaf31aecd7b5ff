#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_pw.txt
: > $out
S="l1_1x1a l1_1x1b l1_1x1c l2_1x1a l2_1x1b l2_1x1c l3_1x1c l3_1x1b l4_1x1b l4_1x1a"
for v in "PSEG_CONV_PW=0" "PSEG_CONV_PW=1" "PSEG_CONV_PW=1 PSEG_CONV_PW_KT=64" "PSEG_CONV_PW=1 PSEG_CONV_PW_BPC=1"; do
  echo "== $v" >> $out
  env $v python tools/bench_conv.py fp32 $S >> $out 2>&1
done
for v in 0 1; do
  PSEG_CONV_PW=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --also "" --no-cpu-baseline --no-roofline > gpurun_out/r05_bench_pw$v.log 2>&1
done
