#!/bin/bash
# (needs the lab build: python -m pytorch_segmentation_amd.csrc.build --lab; PSEG_LIB_PATH=pytorch_segmentation_amd/libpseg_amd_lab.so)
# round 4: block / wave tile experiments of the fp16 gather kernel (PSEG_HCONV_TILE, see plan_gather_h)
# usage (on the GPU box): bash tools/exp_htile.sh  -> gpurun_out/r04_htile_*.log
SH="aspp_d6 aspp_1x1 aspp_proj low_proj l1_3x3 l1_1x1b l1_1x1c l2_3x3 l2_1x1b l2_1x1c l3_3x3 l3_1x1b l3_1x1c l4_3x3d2 l4_1x1b l4_1x1a"
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  env "$@" python tools/bench_conv_half.py $SH > gpurun_out/r04_htile_$name.log 2>&1
  echo "== $name: $(tail -1 gpurun_out/r04_htile_$name.log)"
}
run base
run t1_4w PSEG_HCONV_TILE=1
run t1_4w_s3 PSEG_HCONV_TILE=1 PSEG_HCONV_STAGES=3
run t2_256x128 PSEG_HCONV_TILE=2
run t2_256x128_s3 PSEG_HCONV_TILE=2 PSEG_HCONV_STAGES=3
run t2_256x128_kb32 PSEG_HCONV_TILE=2 PSEG_HCONV_KB=32 PSEG_HCONV_STAGES=3
run t2_256x128_kb32s4 PSEG_HCONV_TILE=2 PSEG_HCONV_KB=32 PSEG_HCONV_STAGES=4
run t3_256x256_s3 PSEG_HCONV_TILE=3 PSEG_HCONV_KB=32 PSEG_HCONV_STAGES=3
run t3_256x256_s4 PSEG_HCONV_TILE=3 PSEG_HCONV_KB=32 PSEG_HCONV_STAGES=4
