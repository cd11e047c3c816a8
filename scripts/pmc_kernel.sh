# rocprofv3 PMC passes (separate runs, counters + kernel-trace only) over an arbitrary command, summarised for one kernel
#   bash scripts/pmc_kernel.sh TAG KERNEL_SUBSTRING -- python3 scripts/bench_configs.py ...
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
tag=$1; kern=$2; shift 3
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - "$tag" "$kern" "$@" <<'PY'
import collections, csv, glob, json, sys
tag, kern = sys.argv[1], sys.argv[2]
out = {}; dur = []; nd = 0
for f in sorted(glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True)):
    acc = collections.defaultdict(float); d = {}
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); d[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out.update(acc)
    if d: dur.append(sum(d.values())); nd = len(d)
res = {"kernel": kern, "command": " ".join(sys.argv[3:]), "dispatches": nd, "counters_summed_over_dispatches": out, "kernel_ns_total_per_pass": dur}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    res["hbm_traffic_bytes"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
    res["hbm_traffic_bytes_per_launch"] = res["hbm_traffic_bytes"] / max(1, nd)
    res["hbm_traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE) KiB -> bytes, FETCH doubled per MI355X_MICROARCH.md"
json.dump(res, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
