#!/bin/bash
# End-of-round measurement set (run on the GPU box via gpurun): bench line, kernel tables (two streams / one stream),
# counters per kernel class, the launch-bound configurations.  usage: tools/refresh_profiles.sh <tag>
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
TAG=${1:-r03}
O=gpurun_out
python3 bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err || { echo bench failed; tail -5 $O/${TAG}_bench_n1.err; exit 1; }
echo "bench done"
prof() {   # name, env..., -- policy
  name=$1; pol=$2; shift 2
  rm -rf $O/prof_${TAG}_$name
  env "$@" true
  ( export "$@"; rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$name -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --precision $pol --also "" > $O/prof_${TAG}_$name.log 2>&1 ) || { echo "prof $name failed"; exit 1; }
  python3 tools/prof_summary.py $(find $O/prof_${TAG}_$name -name "*.db" | head -1) 7 $O/${TAG}_kernel_stats_$name.csv > $O/${TAG}_table_$name.md
  echo "prof $name done"
}
prof fp32 fp32 PSEG_OVERLAP_WGRAD=1
prof fp32_1s fp32 PSEG_OVERLAP_WGRAD=0
prof half half PSEG_OVERLAP_WGRAD=1
prof half_1s half PSEG_OVERLAP_WGRAD=0
prof mixed_1s mixed PSEG_OVERLAP_WGRAD=0
for cfg in "hrnet 8 512 21 20" "unet 8 256 2 30"; do
  for pol in fp32 half mixed limb; do
    for g in 0 1; do
      echo "graph=$g $(PSEG_PRECISION=$pol PSEG_GRAPH=$g python3 tools/bench_model.py $cfg 2>&1 | grep -a 'ms/step\|lane executor')"
    done
  done
done > $O/${TAG}_small_configs.txt 2>&1
echo "small configs done"
tools/pmc_step.sh $TAG fp32 half > $O/${TAG}_pmc.log 2>&1 || { echo pmc failed; tail -5 $O/${TAG}_pmc.log; exit 1; }
echo "pmc done"
