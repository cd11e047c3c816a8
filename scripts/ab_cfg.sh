# A/B of library variants on ONE box for a scripts/bench_configs.py workload:  bash scripts/ab_cfg.sh "ARGS" name1 name2 ...
cd "$GRAFT_REPO_ROOT/deepgroebner_amd"
cp libbbx.so libbbx_orig.so
args=$1; shift
for r in 1 2; do for v in "$@"; do cp libbbx_$v.so libbbx.so; echo "$v: $(cd ..; python scripts/bench_configs.py $args --cpu-envs 0 --no-twin 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['gpu_steps_per_s']/1e6,3), 'M env-steps/s')")"; done; done
cp libbbx_orig.so libbbx.so
