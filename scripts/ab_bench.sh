# A/B of library variants on ONE box: deepgroebner_amd/libbbx_<name>.so for every name given, bench.py at the driver's
# arguments (--steps 20 --warmup 5) three times each, interleaved; prints value and long_launch.value in M env-steps/s
cd "$GRAFT_REPO_ROOT/deepgroebner_amd"
cp libbbx.so libbbx_orig.so
for r in 1 2 3; do for v in "$@"; do cp libbbx_$v.so libbbx.so; echo "$v: $(cd ..; python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']/1e6,1), round(d['long_launch']['value']/1e6,1), d.get('session_stats',{}).get('whole_run'))")"; done; done
cp libbbx_orig.so libbbx.so
