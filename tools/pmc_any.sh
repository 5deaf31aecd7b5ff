#!/bin/bash
# hardware counters of the kernels of an arbitrary python command: tools/pmc_any.sh <tag> <script> [args...]
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
TAG=$1; shift
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  c=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmcany_${TAG}_$c
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmcany_${TAG}_$c -- python3 "$@" > gpurun_out/pmcany_${TAG}_$c.log 2>&1
done
python3 tools/pmc_summarize.py gpurun_out/pmcany_${TAG}_* > gpurun_out/pmcany_${TAG}.txt 2>&1
