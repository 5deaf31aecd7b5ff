#!/bin/bash
# stream-order / priority experiments for the half step (two repetitions each)
cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; for rep in 1 2; do r=$(env "$@" python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"); echo "$name rep$rep: $r"; done; }
run base PSEG_NOOP=1
run after_dgrad PSEG_WGRAD_AFTER_DGRAD=1
run aux2 PSEG_AUX_STREAMS=2
run bpc2 PSEG_WGRAD_BPC=2
run dgrad_prio0 PSEG_DGRAD_PRIO=0
run dgrad_prio2 PSEG_DGRAD_PRIO=2
run one_stream PSEG_OVERLAP_WGRAD=0
run graph PSEG_GRAPH=1
