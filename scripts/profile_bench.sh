# rocprofv3 evidence for bench.py (3-20-10-weighted, B=4096): kernel-trace stats at the driver's arguments and at the
# defaults, then PMC passes (each its own rocprofv3 run: counters + kernel-trace only).  The PMC passes run the step kernel
# one kernel per launch (--no-persistent, 1024 steps per launch): counter collection serialises kernels, so the small
# kernels that raise a persistent session's step counter could not run beside the session's kernel, whose waves would sit
# polling — the instruction mix of a step is the same code either way (fast_body), the poll excepted.
#   bash scripts/profile_bench.sh r03      -> gpurun_out/prof_r03_*  (summarise with scripts/pmc_summary.py r03)
# The library is built BEFORE any profiled run (bench.py would otherwise start hipcc / make as children of the profiled
# process); the program itself stands directly behind `--`.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
tag=${1:-r04}
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
for kw in "20 5" "1024 64"; do
  set -- $kw
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_k$1 -o bench -- python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-long-launch > gpurun_out/prof_${tag}_k$1.json 2> gpurun_out/prof_${tag}_k$1.log || { echo "kernel-trace run K=$1 failed"; exit 1; }
done
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python3 bench.py --steps 1024 --warmup 64 --no-persistent --no-cpu-baseline --no-long-launch > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.log || { echo "pmc pass $i failed"; exit 1; }
done
