# time bench.py (1024-step launches) for libbbx built with the given -D flags, one build per argument
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
build() { python3 -c "import sys, __graft_entry__ as g; g.build(force=True, defines=[d[2:] for d in sys.argv[1:]])" $1 2>&1 | grep -i "error" ; true; }
run() { python bench.py --steps 1024 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']/1e6,1), 'M env-steps/s', round(d['roofline']['kernel_ms_per_launch'],4), 'ms/launch')"; }
for v in "$@"; do build "$v"; run "[$v]"; run "[$v]"; done
