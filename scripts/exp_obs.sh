# ablation of the observation writer of the HBM-resident binomial class: no stores / no gathers
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
# the experiment code is not in the product headers: it is scripts/patches/issue_and_observation_experiments.patch
patch -p0 < scripts/patches/issue_and_observation_experiments.patch || exit 1
build() { python3 -c "import sys, __graft_entry__ as g; g.build(force=True, defines=[d[2:] for d in sys.argv[1:]])" $1 2>&1 | grep -i "error" ; true; }
run() { timeout -k 10 200 python scripts/bench_configs.py 5-10-5-uniform --batch 4096 --steps 2048 --obs-rows 2048 --cpu-envs 0 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d['gpu_steps_per_s']/1e6,1))" "$1"; }
for v in "" "-DBBX_ABL_OBS=1" "-DBBX_ABL_OBS=2"; do build "$v"; run "[$v]"; done
run "[no obs at all]" --no-obs
patch -R -p0 < scripts/patches/issue_and_observation_experiments.patch; build ""
