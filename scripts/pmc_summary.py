"""Summarise the rocprofv3 runs of scripts/profile_bench.sh into profiles/<tag>_pmc_fast_kernel.json and copy the
kernel-trace stats: per counter the MEDIAN over the dispatches of the benchmarked kernel (the timed region is R equal
launches; pre-roll / warm-up launches are the minority), HBM traffic of a K-step launch as fixed + per-step parts from
the FETCH_SIZE / WRITE_SIZE passes at two launch lengths (FETCH doubled, MI355X_MICROARCH.md), and the scalar-issue
bound (scalar-pipe instructions per cycle per CU against the one scalar unit of a CU)."""
import collections, csv, glob, json, os, shutil, statistics, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
kern = sys.argv[2] if len(sys.argv) > 2 else "bbx_fast_headline_kernel"
B = 4096


def passes(k):
    out, durs = {}, []
    for f in sorted(glob.glob("gpurun_out/pmc_%s_k%d_*/**/*counter_collection.csv" % (tag, k), recursive=True)):
        by = collections.defaultdict(dict); dur = {}
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                by[r["Counter_Name"]][r["Dispatch_Id"]] = float(r["Counter_Value"])
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for c, v in by.items():
            out[c] = statistics.median(v.values())
        if dur:
            durs.append(statistics.median(dur.values()))
    return out, durs


c1024, d1024 = passes(1024)
c20, d20 = passes(20)
res = {"round": 2, "kernel": kern, "workload": "bench.py: 3-20-10-weighted, 4096 envs, k=2, obs every step, ideals drawn on the device; launches of 1024 and of 20 steps",
       "dispatch_ns_median": {"1024": d1024, "20": d20}, "counters_1024_step_launch": c1024, "counters_20_step_launch": c20}
if "FETCH_SIZE" in c1024 and "WRITE_SIZE" in c1024 and "FETCH_SIZE" in c20 and "WRITE_SIZE" in c20:
    t = {k: (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 for k, c in ((1024, c1024), (20, c20))}
    per_step = (t[1024] - t[20]) / (1024 - 20)
    res["hbm_traffic"] = {"bytes_per_batch_step": per_step, "fixed_bytes_per_launch": t[20] - 20 * per_step,
                          "measured_bytes_per_launch": {"1024": t[1024], "20": t[20]},
                          "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB -> bytes, FETCH doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                                  "coalesced reads; our access widths are not calibrated); per-step part = slope between the two launch lengths"}
if "SQ_INSTS_SALU" in c1024 and "SQ_BUSY_CYCLES" in c1024:
    es = B * 1024.0
    scalar = c1024["SQ_INSTS_SALU"] + c1024.get("SQ_INSTS_BRANCH", 0) + c1024.get("SQ_INSTS_SMEM", 0)
    cyc_cu = c1024["SQ_BUSY_CYCLES"] / 32.0             # summed over the 32 shader engines
    res["per_env_step"] = {k: round(v / es, 2) for k, v in c1024.items() if k.startswith("SQ_INSTS")}
    res["issue_bound"] = {"kind": "scalar-issue", "achieved": scalar / 256.0 / cyc_cu, "peak": 1.0, "unit": "scalar-pipe instructions / cycle / CU",
                          "valu_per_cycle_per_simd": c1024.get("SQ_INSTS_VALU", 0) / 1024.0 / cyc_cu,
                          "note": "SALU + branch + SMEM instructions of one 1024-step launch / 256 CUs / (SQ_BUSY_CYCLES / 32 shader engines); one scalar unit per CU"}
    if "SQ_WAVE_CYCLES" in c1024:
        wc = c1024["SQ_WAVE_CYCLES"]
        res["wave_time_split"] = {"issuing (SQ_ACTIVE_INST_ANY)": round(c1024.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                  "parked at s_waitcnt (SQ_WAIT_ANY)": round(c1024.get("SQ_WAIT_ANY", 0) / wc, 3),
                                  "issue stalls (SQ_WAIT_INST_ANY)": round(c1024.get("SQ_WAIT_INST_ANY", 0) / wc, 3)}
os.makedirs("profiles", exist_ok=True)
json.dump(res, open("profiles/%s_pmc_fast_kernel.json" % tag, "w"), indent=1)
for k in (20, 1024):
    for f in glob.glob("gpurun_out/prof_%s_k%d/**/*kernel_stats.csv" % (tag, k), recursive=True):
        shutil.copy(f, "profiles/%s_bench_k%d_kernel_stats.csv" % (tag, k))
print(json.dumps({k: res.get(k) for k in ("hbm_traffic", "issue_bound", "dispatch_ns_median")}, indent=1))
