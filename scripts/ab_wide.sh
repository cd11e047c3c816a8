# A/B of library variants on ONE box for the two cyclic-7 figures:  bash scripts/ab_wide.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT/deepgroebner_amd"
cp libbbx.so libbbx_orig.so
for v in "$@"; do cp libbbx_$v.so libbbx.so
  echo "$v: B=512 random $(cd ..; python scripts/bench_configs.py cyclic-7 --batch 512 --steps 512 --cpu-envs 0 --no-twin 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['gpu_steps_per_s']))") env-steps/s, one env Degree $(cd ..; python scripts/bench_configs.py cyclic-7 --batch 1 --agent degree --to-completion --cpu-envs 0 --obs-rows 4096 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['gpu_seconds'],2))") s"
done
cp libbbx_orig.so libbbx.so
