# PMC passes over the bench (each its own rocprofv3 run; counters + kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "$2" ] && [ $i -gt $2 ]; then break; fi
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pmc $i failed"
done
