# experiment: cyclic-7 B=512 x 512 with 4..8 waves per environment, 128-register kernel against the uncapped one at two workgroups per CU
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/exp_wide_waves.log
: > $L
for nw in 8 4 5 6; do
  for unc in 0 1; do
    echo "waves $nw uncapped $unc" >> $L
    if [ $unc = 1 ]; then export BBX_WIDE_UNCAPPED=1; else unset BBX_WIDE_UNCAPPED; fi
    timeout -k 10 120 python scripts/bench_configs.py cyclic-7 --batch 512 --steps 512 --cpu-envs 0 --no-twin --wide-waves $nw 2>/dev/null | tail -1 | cut -c1-200 >> $L || exit 1
  done
done
cat $L
