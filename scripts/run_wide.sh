# wide-class iteration helper (GPU box): the wide-class parity tests, then the two cyclic-7 figures and the long general config
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/wide_$1.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cyclic7 or wide or long_polynomials or golden_trace" > gpurun_out/wide_$1_tests.log 2>&1 || { tail -30 gpurun_out/wide_$1_tests.log; exit 1; }
tail -1 gpurun_out/wide_$1_tests.log >> $L
timeout -k 10 300 python scripts/fuzz_wide.py 6 11 > gpurun_out/wide_$1_fuzz.log 2>&1 || { tail -30 gpurun_out/wide_$1_fuzz.log; exit 1; }
tail -1 gpurun_out/wide_$1_fuzz.log >> $L
timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 512 --steps 512 --cpu-envs 0 2>/dev/null | tail -1 | cut -c1-200 >> $L
timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 1 --agent degree --to-completion --cpu-envs 0 --obs-rows 4096 2>/dev/null | tail -1 | cut -c1-200 >> $L
timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 512 --agent degree --to-completion --cpu-envs 0 --obs-rows 4096 2>/dev/null | tail -1 | cut -c1-200 >> $L
timeout -k 10 200 python scripts/bench_configs.py 5-4-4-1.0-uniform --batch 4096 --steps 64 --obs-rows 1024 --cpu-envs 0 2>/dev/null | tail -1 | cut -c1-200 >> $L
cat $L
