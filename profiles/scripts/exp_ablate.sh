#!/bin/bash
# (needs the lab build: python -m pytorch_segmentation_amd.csrc.build --lab; PSEG_LIB_PATH=pytorch_segmentation_amd/libpseg_amd_lab.so)
# which part of the fp16 gather kernel costs what (PSEG_HCONV_ABLATE bits: 1 no stores, 2 no statistics, 4 no operand DMAs
# after the prologue, 8 no MFMAs, 16 no fragment reads); results of ablated runs are wrong by design
cd "$GRAFT_REPO_ROOT"
SH="aspp_d6 aspp_1x1 low_proj l1_3x3 l1_1x1b l2_3x3 l2_1x1b l3_3x3 l3_1x1b l3_1x1c l4_3x3d2 l4_1x1b l4_1x1a"
for a in 0 1 2 3 4 8 16 24 28 31 7; do
  PSEG_HCONV_ABLATE=$a python tools/bench_conv_half.py $SH > gpurun_out/r04_ablate_$a.log 2>&1
  echo "ablate=$a: $(tail -1 gpurun_out/r04_ablate_$a.log)"
done
