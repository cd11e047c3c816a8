# experiment: phase shares of the wide kernel on the long-polynomial general configuration (profiling build of the library)
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/exp_wide_prof2.log
cp deepgroebner_amd/libbbx.so /tmp/libbbx_keep.so && cp deepgroebner_amd/libbbx_prof.so deepgroebner_amd/libbbx.so
( timeout -k 10 300 python scripts/prof_wide.py 5-4-4-1.0-uniform 4096 64 0 lean random 2>&1 | grep -v amdgpu.ids ) > $L
cp /tmp/libbbx_keep.so deepgroebner_amd/libbbx.so
cat $L
