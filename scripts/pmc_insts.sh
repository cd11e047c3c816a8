# Instruction mix of the headline step kernel for library variants deepgroebner_amd/libbbx_<name>.so (one PMC pass each, one kernel
# per launch, 40 launches of 1024 steps): instructions per env-step by class.  bash scripts/pmc_insts.sh base nbk4
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
cp deepgroebner_amd/libbbx.so deepgroebner_amd/libbbx_orig.so
for v in "$@"; do
  cp deepgroebner_amd/libbbx_$v.so deepgroebner_amd/libbbx.so
  rm -rf gpurun_out/pmci_$v
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d gpurun_out/pmci_$v -o p -- python3 bench.py --steps 1024 --warmup 64 --repeats 40 --no-persistent --no-cpu-baseline --no-long-launch > gpurun_out/pmci_$v.json 2> gpurun_out/pmci_$v.log || { echo "pmc pass $v failed"; tail -5 gpurun_out/pmci_$v.log; }
  python3 - "$v" <<'PY'
import csv, glob, json, sys, collections
v = sys.argv[1]
line = json.loads([l for l in open("gpurun_out/pmci_%s.json" % v) if l.startswith("{")][0])
steps = line["roofline"]["batch_steps_through_kernel"]
acc = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmci_%s/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbx_fast_headline_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print(v, {k: round(x / steps / 4096, 1) for k, x in sorted(acc.items())}, "total", round(sum(x for k, x in acc.items() if k.startswith("SQ_INSTS")) / steps / 4096, 1))
PY
done
cp deepgroebner_amd/libbbx_orig.so deepgroebner_amd/libbbx.so
