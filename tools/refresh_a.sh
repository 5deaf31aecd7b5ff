#!/bin/bash
# part 1 of the end-of-round measurement set: bench line + kernel tables.  usage: tools/refresh_a.sh <tag>
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
TAG=${1:-r04}
O=gpurun_out
python3 bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err || { echo bench failed; tail -5 $O/${TAG}_bench_n1.err; exit 1; }
echo "bench done"
prof() {   # name, policy, env...
  name=$1; pol=$2; shift 2
  rm -rf $O/prof_${TAG}_$name
  ( export "$@"; rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$name -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --precision $pol --also "" > $O/prof_${TAG}_$name.log 2>&1 ) || { echo "prof $name failed"; exit 1; }
  python3 tools/prof_summary.py $(find $O/prof_${TAG}_$name -name "*.db" | head -1) 7 $O/${TAG}_kernel_stats_$name.csv > $O/${TAG}_table_$name.md
  rm -rf $O/prof_${TAG}_$name
  echo "prof $name done"
}
prof fp32 fp32 PSEG_OVERLAP_WGRAD=1
prof fp32_1s fp32 PSEG_OVERLAP_WGRAD=0
prof half half PSEG_OVERLAP_WGRAD=1
prof half_1s half PSEG_OVERLAP_WGRAD=0
prof mixed_1s mixed PSEG_OVERLAP_WGRAD=0
