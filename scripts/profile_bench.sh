# rocprofv3 evidence for bench.py (3-20-10-weighted, B=4096): kernel-trace stats at the driver's arguments and at the
# defaults, then PMC passes (each its own rocprofv3 run: counters + kernel-trace only), FETCH/WRITE at both launch
# lengths so that bench.py can state the HBM traffic of a K-step launch for any K.
#   bash scripts/profile_bench.sh r02      -> gpurun_out/prof_r02_*  (summarise with scripts/pmc_summary.py)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
tag=${1:-r02}
for kw in "20 5" "1024 64"; do
  set -- $kw
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_k$1 -o bench -- python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline > gpurun_out/prof_${tag}_k$1.log 2>&1 || echo "kernel-trace run K=$1 failed"
done
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for kw in "1024 64" "20 5"; do
    set2=($kw)
    if [ $i -le 2 ] && [ ${set2[0]} = 20 ]; then continue; fi
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_k${set2[0]}_$i -o p -- python3 bench.py --steps ${set2[0]} --warmup ${set2[1]} --no-cpu-baseline > gpurun_out/pmc_${tag}_k${set2[0]}_$i.log 2>&1 || echo "pmc $i K=${set2[0]} failed"
  done
done
