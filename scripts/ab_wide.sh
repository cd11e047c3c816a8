# A/B of library variants on the wide-class figures: ab_wide.sh name1 name2 ...  (deepgroebner_amd/libbbx_<name>.so)
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/ab_wide.log
: > $L
cp deepgroebner_amd/libbbx.so /tmp/libbbx_keep.so
for v in "$@"; do
  cp deepgroebner_amd/libbbx_$v.so deepgroebner_amd/libbbx.so
  echo "== $v" >> $L
  timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 512 --steps 512 --cpu-envs 0 --no-twin 2>/dev/null | tail -1 | cut -c1-160 >> $L || exit 1
  timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 1 --agent degree --to-completion --cpu-envs 0 --obs-rows 4096 --no-twin 2>/dev/null | tail -1 | cut -c1-160 >> $L || exit 1
  timeout -k 10 200 python scripts/bench_configs.py cyclic-7 --batch 512 --agent degree --to-completion --cpu-envs 0 --obs-rows 4096 --no-twin 2>/dev/null | tail -1 | cut -c1-160 >> $L || exit 1
  timeout -k 10 200 python scripts/bench_configs.py 5-4-4-1.0-uniform --batch 4096 --steps 64 --obs-rows 1024 --cpu-envs 0 --no-twin 2>/dev/null | tail -1 | cut -c1-160 >> $L || exit 1
done
cp /tmp/libbbx_keep.so deepgroebner_amd/libbbx.so
cat $L
