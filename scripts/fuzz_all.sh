# every fuzz script for a few rounds each with a fresh seed (GPU box): bash scripts/fuzz_all.sh [SEED]
cd "$GRAFT_REPO_ROOT"
seed=${1:-41}
L=gpurun_out/fuzz_all.log
: > $L
for f in parity gym policy strategies value wide generators sessions; do
  echo "== fuzz_$f" >> $L
  timeout -k 10 170 python scripts/fuzz_$f.py ${2:-12} $seed > gpurun_out/fuzz_$f.log 2>&1; rc=$?
  tail -2 gpurun_out/fuzz_$f.log | cut -c1-300 >> $L
  echo "rc=$rc" >> $L
done
cat $L
