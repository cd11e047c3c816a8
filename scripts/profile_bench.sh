# rocprofv3 evidence for the bench (default workload): kernel-trace stats, then the PMC passes (separate runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r01}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_${tag}.log 2>&1 || echo "kernel-trace run failed"
bash scripts/pmc_passes.sh ${tag}
