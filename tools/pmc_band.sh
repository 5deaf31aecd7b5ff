#!/bin/bash
# L2<->fabric fetch bytes of the ASPP convs with / without class-sorted rows (run on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for mode in default noband noskip; do
  case $mode in noband) export PSEG_CONV_NOBAND=1;; noskip) unset PSEG_CONV_NOBAND; export PSEG_CONV_NOSKIP=1;; esac
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcband_$mode -- python3 tools/bench_conv.py fp32 aspp_d6 aspp_d12 > gpurun_out/pmcband_$mode.log 2>&1
  echo "== $mode"; python3 tools/pmc_summarize.py gpurun_out/pmcband_$mode | grep -A1 "gather_f32_dma\|wgrad_f32"
done
