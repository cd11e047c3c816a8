# Deeper policies (two / three hidden layers: bbx_pmlp2_act / bbx_pmlp3_act; four: torch ops, eager and replayed from a HIP graph): bench lines and
# the kernel-trace summary of the two-layer run.   bash scripts/profile_policy_deep.sh r03   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
tag=${1:-r03}
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
out=gpurun_out/${tag}_bench_policy_deep.jsonl
: > $out
timeout -k 10 120 python3 scripts/bench_policy.py --hidden 128,128 --steps 1000 2>/dev/null | tail -1 >> $out
timeout -k 10 120 python3 scripts/bench_policy.py --hidden 64,64 --steps 1000 2>/dev/null | tail -1 >> $out
timeout -k 10 120 python3 scripts/bench_policy.py --hidden 128,128,128 --steps 1000 2>/dev/null | tail -1 >> $out
timeout -k 10 120 python3 scripts/bench_policy.py --hidden 128,128,128,128 --steps 300 2>/dev/null | tail -1 >> $out
timeout -k 10 120 python3 scripts/bench_policy.py --hidden 128,128,128,128 --steps 300 --batch 256 2>/dev/null | tail -1 >> $out
timeout -k 10 120 python3 scripts/bench_policy.py --hidden 128,128,128,128 --steps 300 --batch 256 --graph 2>/dev/null | tail -1 >> $out
cat $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_policy_deep -o p -- python3 scripts/bench_policy.py --hidden 128,128 --steps 1000 > gpurun_out/prof_${tag}_policy_deep.log 2>&1
