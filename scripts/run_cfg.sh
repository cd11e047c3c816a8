cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r3_cfg_$1.log
: > $L
timeout -k 10 200 python scripts/bench_configs.py 5-10-5-uniform --batch 4096 --steps 2048 --obs-rows 2048 --cpu-envs 0 2>/dev/null | tail -1 | cut -c1-330 >> $L
timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 512 --steps 512 --cpu-envs 0 2>/dev/null | tail -1 | cut -c1-330 >> $L
timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 1 --agent degree --to-completion --cpu-envs 0 --obs-rows 4096 2>/dev/null | tail -1 | cut -c1-330 >> $L
timeout -k 10 200 python scripts/bench_policy.py 2>/dev/null | tail -1 | cut -c1-330 >> $L
timeout -k 10 200 python scripts/bench_policy.py --per-step 2>/dev/null | tail -1 | cut -c1-330 >> $L
timeout -k 10 200 python scripts/bench_configs.py 5-4-4-1.0-uniform --batch 4096 --steps 64 --obs-rows 1024 --cpu-envs 0 2>/dev/null | tail -1 | cut -c1-330 >> $L
timeout -k 10 200 python scripts/bench_configs.py 3-5-4-0.5-uniform --batch 4096 --steps 512 --obs-rows 512 --cpu-envs 0 2>/dev/null | tail -1 | cut -c1-330 >> $L
cat $L
