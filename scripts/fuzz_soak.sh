# long randomised runs of every fuzzer (GPU box): bash scripts/fuzz_soak.sh SEED SECONDS_EACH
cd "$GRAFT_REPO_ROOT"
seed=${1:-1}; secs=${2:-120}
L=gpurun_out/fuzz_soak_$seed.log
: > $L
for f in parity gym policy strategies value wide generators sessions; do
  echo "== fuzz_$f" >> $L
  timeout -k 10 $secs python scripts/fuzz_$f.py 100000 $seed > gpurun_out/fuzz_soak_$f.log 2>&1; rc=$?
  grep -c "^ok" gpurun_out/fuzz_soak_$f.log >> $L
  grep -v "^ok" gpurun_out/fuzz_soak_$f.log | grep -v amdgpu | tail -5 | cut -c1-400 >> $L
  echo "rc=$rc (124 = time is up, no mismatch until then)" >> $L
done
cat $L
