"""Summarise the rocprofv3 runs of scripts/profile_bench.sh into profiles/<tag>_pmc_fast_kernel.json and copy the kernel-trace
stats.  The step kernel of the bench line is the kernel of a persistent session: a run has a handful of its dispatches
(pre-roll, warm-up, calibration, the 10 ms slices of the timed region), so counters are SUMMED over the dispatches of a
pass and divided by the batch steps the run pushed through the kernel (bench.py reports them:
roofline.batch_steps_through_kernel).  HBM traffic = (2 FETCH_SIZE + WRITE_SIZE) KiB (FETCH doubled, MI355X_MICROARCH.md);
scalar-issue bound = scalar-pipe instructions per cycle per CU against the one scalar unit of a CU.
    python scripts/pmc_summary.py r03"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
kern = sys.argv[2] if len(sys.argv) > 2 else "bbx_fast_headline_persistent_kernel"      # the kernel of the bench line (kernel-trace stats)
pmc_kern = "bbx_fast_headline_kernel"                    # the same step code, one kernel per launch: what the PMC passes run
B = 4096


def bench_line(path):
    lines = [ln for ln in open(path) if ln.startswith("{")]
    if len(lines) != 1:
        sys.exit("%s: expected the bench line, found %d lines" % (path, len(lines)))
    return json.loads(lines[0])


counters, per_pass = {}, []
for i in range(1, 5):
    files = sorted(glob.glob("gpurun_out/pmc_%s_%d/**/*counter_collection.csv" % (tag, i), recursive=True))
    if not files:
        sys.exit("pass %d of the PMC runs is missing (gpurun_out/pmc_%s_%d): not summarising stale numbers" % (i, tag, i))
    line = bench_line("gpurun_out/pmc_%s_%d.json" % (tag, i))
    steps = line["roofline"]["batch_steps_through_kernel"]
    acc = collections.defaultdict(float); dur = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if pmc_kern in r["Kernel_Name"] and "persistent" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not acc:
        sys.exit("pass %d: no dispatch of %s" % (i, pmc_kern))
    per_pass.append({"pass": i, "dispatches": len(dur), "kernel_ns_total": sum(dur.values()), "batch_steps": steps,
                     "us_per_batch_step_under_counters": sum(dur.values()) / steps / 1e3})
    for c, v in acc.items():
        counters[c] = v / steps                            # per batch step (4096 environment steps)
res = {"round": int(tag.lstrip("r") or 0), "kernel": pmc_kern,
       "workload": "bench.py --steps 1024 --warmup 64 --no-persistent: 3-20-10-weighted, 4096 envs, k=2, obs every step, ideals drawn on the device, "
                   "one kernel per launch (counter collection serialises kernels: see scripts/profile_bench.sh)",
       "passes": per_pass, "counters_per_batch_step": counters}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    res["hbm_traffic"] = {"bytes_per_batch_step": (2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0, "fixed_bytes_per_launch": 0.0,
                          "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB -> bytes over all dispatches of the kernel / batch steps through it, FETCH doubled per "
                                  "MI355X_MICROARCH.md; the records' load / store at the ends of a session's kernels is in the average"}
if "SQ_INSTS_SALU" in counters and "SQ_BUSY_CYCLES" in counters:
    scalar = counters["SQ_INSTS_SALU"] + counters.get("SQ_INSTS_BRANCH", 0) + counters.get("SQ_INSTS_SMEM", 0)
    cyc_cu = counters["SQ_BUSY_CYCLES"] / 32.0             # summed over the 32 shader engines
    res["per_env_step"] = {k: round(v / B, 2) for k, v in counters.items() if k.startswith("SQ_INSTS")}
    res["issue_bound"] = {"kind": "scalar-issue", "achieved": scalar / 256.0 / cyc_cu, "peak": 1.0, "unit": "scalar-pipe instructions / cycle / CU",
                          "valu_per_cycle_per_simd": counters.get("SQ_INSTS_VALU", 0) / 1024.0 / cyc_cu,
                          "note": "SALU + branch + SMEM instructions / 256 CUs / (SQ_BUSY_CYCLES / 32 shader engines), per batch step; one scalar unit per CU"}
    if "SQ_WAVE_CYCLES" in counters:
        wc = counters["SQ_WAVE_CYCLES"]
        res["wave_time_split"] = {"issuing (SQ_ACTIVE_INST_ANY)": round(counters.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                  "parked at s_waitcnt (SQ_WAIT_ANY)": round(counters.get("SQ_WAIT_ANY", 0) / wc, 3),
                                  "issue stalls (SQ_WAIT_INST_ANY)": round(counters.get("SQ_WAIT_INST_ANY", 0) / wc, 3)}
os.makedirs("profiles", exist_ok=True)
json.dump(res, open("profiles/%s_pmc_fast_kernel.json" % tag, "w"), indent=1)
for k in (20, 1024):
    found = glob.glob("gpurun_out/prof_%s_k%d/**/*kernel_stats.csv" % (tag, k), recursive=True)
    if not found:
        sys.exit("kernel-trace stats of K=%d are missing" % k)
    shutil.copy(found[0], "profiles/%s_bench_k%d_kernel_stats.csv" % (tag, k))
    shutil.copy("gpurun_out/prof_%s_k%d.json" % (tag, k), "profiles/%s_bench_k%d_line.json" % (tag, k))
    line = bench_line("gpurun_out/prof_%s_k%d.json" % (tag, k))
    for r in csv.DictReader(open(found[0])):
        if kern in r["Name"]:
            tot = float(r["TotalDurationNs"]); steps = line["roofline"]["batch_steps_through_kernel"]
            print("K=%d: %s x%s, %.3f ms in all / %d batch steps = %.3f us per batch step; bench line ms_per_step %.3f us, value %.1f M" % (
                k, kern, r["Calls"], tot / 1e6, steps, tot / steps / 1e3, line["ms_per_step"] * 1e3, line["value"] / 1e6))
print(json.dumps({k: res.get(k) for k in ("hbm_traffic", "issue_bound", "per_env_step", "wave_time_split")}, indent=1))
