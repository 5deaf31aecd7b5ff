#!/bin/bash
# hardware counters of the conv kernels on representative shapes (run on the GPU box via gpurun)
# usage: tools/pmc_conv.sh <policy> <shape>...   (separate --pmc passes, kernel-trace only)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
POL=${1:-fp32}; shift
SHAPES=${@:-aspp_d6 l4_3x3d2 l1_1x1b low_proj}
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/bench_conv.py $POL $SHAPES > gpurun_out/pmc_$tag.log 2>&1
done
python3 tools/pmc_summarize.py gpurun_out/pmc_* > gpurun_out/pmc_summary.txt 2>&1
