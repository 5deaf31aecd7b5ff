#!/bin/bash
# Hardware counters of one whole training step, per kernel class (run on the GPU box via gpurun):
#   tools/pmc_step.sh <tag> [policy ...]        e.g.  tools/pmc_step.sh r02 fp32 mixed
# Separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA-busy + GRBM), kernel-trace only, over bench.py itself
# (headline policy only, no CPU baseline, no metered step); tools/pmc_step.py folds them into
# gpurun_out/<tag>_pmc_traffic.json / .md, which are then copied to profiles/ (tracked) -- bench.py reads
# profiles/r*_pmc_traffic.json for roofline.traffic.
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
TAG=${1:-r02}; shift
POLS=${@:-fp32 mixed}
for pol in $POLS; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    c=$(echo $pass | cut -d' ' -f1)
    out=gpurun_out/pmc_${TAG}_${pol}_${c}
    rm -rf $out
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 \
        --no-cpu-baseline --no-roofline --precision $pol --also "" > $out.log 2>&1 || { echo "pass $pol/$c failed"; tail -5 $out.log; exit 1; }
    echo "pass $pol / $c done"
  done
done
python3 tools/pmc_step.py $TAG 3 $POLS
