for v in "0 3" "1 3" "1 1" "1 2" "0 3" "1 1"; do set -- $v; echo "fuse $1 where $2: $(PSEG_FUSE_BN_BWD=$1 PSEG_BNS_H_WHERE=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also half 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['other_policies']['half']['ms_per_step'], d['other_policies']['half']['step_mode'])")"; done
