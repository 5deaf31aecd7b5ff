#!/bin/bash
# A/B of one environment variable on the bench step: usage tools/ab_env.sh VAR v0 v1 [reps]   -> fp32 headline and half ms/step
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do for v in $A $B; do echo "$VAR=$v: $(env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also half 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fp32 %.2f half %.2f' % (d['ms_per_step'], d['other_policies']['half']['ms_per_step']))")"; done; done
