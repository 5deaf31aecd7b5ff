#!/bin/bash
# part 2: the launch-bound configurations, counters per kernel class, the forced 1-rank reducer.  usage: tools/refresh_b.sh <tag>
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
TAG=${1:-r04}
O=gpurun_out
for cfg in "hrnet 8 512 21 20" "unet 8 256 2 30"; do
  for pol in fp32 half mixed limb; do
    for g in 0 1; do
      echo "graph=$g $(PSEG_PRECISION=$pol PSEG_GRAPH=$g python3 tools/bench_model.py $cfg 2>&1 | grep -a 'ms/step\|lane executor')"
    done
    # no environment variable: the Trainer's AUTO mode (what `python train.py ...` runs)
    echo "graph=auto $(PSEG_PRECISION=$pol python3 tools/bench_model.py $cfg 2>&1 | grep -a 'ms/step\|graph mode')"
  done
done > $O/${TAG}_small_configs.txt 2>&1
echo "small configs done"
tools/pmc_step.sh $TAG fp32 half > $O/${TAG}_pmc.log 2>&1 || { echo pmc failed; tail -5 $O/${TAG}_pmc.log; exit 1; }
echo "pmc done"
python3 tools/bench_forced_reducer.py > $O/${TAG}_forced.log 2>&1
echo "forced reducer done"
