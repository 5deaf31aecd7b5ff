#!/bin/bash
# per-block phase split of the exact-fp32 LDS-DMA gather kernel on short / medium contractions (trace build)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_phases.txt
: > $out
export PSEG_LIB_PATH=$PWD/pytorch_segmentation_amd/libpseg_amd_trace.so
for a in "16 256 32 1024 1" "16 1024 32 256 1" "16 64 128 256 1" "16 256 128 64 1" "16 256 32 256 3" "16 512 32 512 3" "16 128 64 512 1"; do
  echo "== $a" >> $out
  python tools/conv_phases.py $a >> $out 2>&1
done
