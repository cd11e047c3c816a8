# Which resource binds the fast kernel?  Rebuild libbbx on the GPU box with N extra independent instructions per env-step
# (scalar adds/multiplies, vector adds/multiplies, s_nop) and time bench.py's 1024-step launches.
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
build() { (cd deepgroebner_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $1 -o ../libbbx.so bbx_kernels.hip bbx_api.cpp bbx_ideals.cpp 2>&1 | grep -i error); }
run() { python bench.py --steps 1024 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']/1e6,1), 'M env-steps/s', round(d['roofline']['kernel_ms_per_launch'],4), 'ms/launch')"; }
for v in "" "-DBBX_EXP_SALU=50" "-DBBX_EXP_SALU=100" "-DBBX_EXP_VALU=50" "-DBBX_EXP_VALU=100" "-DBBX_EXP_NOP=50"; do build "$v"; run "baseline$v"; run "baseline$v"; done
