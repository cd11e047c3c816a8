# experiment: phase shares of the wide kernel for the slowest cyclic-7 environment of the batch (profiling build of the library)
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/exp_wide_prof.log
cp deepgroebner_amd/libbbx.so /tmp/libbbx_keep.so && cp deepgroebner_amd/libbbx_prof.so deepgroebner_amd/libbbx.so
( timeout -k 10 200 python scripts/prof_wide.py cyclic-7 128 512 0 lean random 2>&1 | grep -v amdgpu.ids
  echo "--- one environment: agent seed 45 / 100"
  timeout -k 10 100 python scripts/prof_wide.py cyclic-7 1 512 ${1:-45} lean random 2>&1 | grep -v amdgpu.ids ) > $L
cp /tmp/libbbx_keep.so deepgroebner_amd/libbbx.so
cat $L
