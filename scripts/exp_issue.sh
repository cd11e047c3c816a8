# Which resource binds the fast kernel?  Rebuild libbbx on the GPU box with N extra independent instructions per env-step
# (scalar adds/multiplies, vector adds/multiplies, s_nop) and time bench.py's 1024-step launches.
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
# the experiment code is not in the product headers: it is scripts/patches/issue_and_observation_experiments.patch
patch -p0 < scripts/patches/issue_and_observation_experiments.patch || exit 1
build() { python3 -c "import sys, __graft_entry__ as g; g.build(force=True, defines=[d[2:] for d in sys.argv[1:]])" $1 2>&1 | grep -i "error" ; true; }
run() { python bench.py --steps 1024 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']/1e6,1), 'M env-steps/s', round(d['roofline']['kernel_ms_per_launch'],4), 'ms/launch')"; }
for v in "" "-DBBX_EXP_SALU=50" "-DBBX_EXP_SALU=100" "-DBBX_EXP_VALU=50" "-DBBX_EXP_VALU=100" "-DBBX_EXP_NOP=50"; do build "$v"; run "baseline$v"; run "baseline$v"; done
patch -R -p0 < scripts/patches/issue_and_observation_experiments.patch; build ""
