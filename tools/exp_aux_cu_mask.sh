cd $GRAFT_REPO_ROOT
for pol in fp32 half; do for n in 0 16 32 64; do echo "policy=$pol aux_free_cus=$n: $(PSEG_AUX_FREE_CUS=$n python3 bench.py --no-cpu-baseline --no-roofline --precision $pol --also "" --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))")"; done; done
