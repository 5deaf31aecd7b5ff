#!/bin/bash
# hardware counters of the conv kernels on three representative shapes (run on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/bench_conv.py aspp_d6 l4_3x3d2 l1_1x1b low_proj > gpurun_out/pmc_$tag.log 2>&1
done
ls gpurun_out/pmc_*/*/ | head -30
