"""Copy what scripts/profile_bench.sh / profile_configs.sh / the policy profile left under gpurun_out/ into profiles/ (tracked):
per workload the rocprofv3 kernel-trace stats, the PMC summary of the step kernel, and the bench_configs.py line with the
measured HBM traffic of the launch (PMC) put next to its algorithmic bytes.
    python scripts/profile_configs_summary.py r02"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
os.makedirs("profiles", exist_ok=True)
lines = []
for name, kern in (("cyclic7", "wide_kernel"), ("u5", "binom_kernel"), ("general", "general_kernel")):
    for f in glob.glob("gpurun_out/cfg_%s_%s_stats/**/*kernel_stats.csv" % (tag, name), recursive=True):
        shutil.copy(f, "profiles/%s_%s_kernel_stats.csv" % (tag, kern))
    pj = "gpurun_out/pmc_%s_%s.json" % (tag, name)
    pmc = None
    if os.path.exists(pj):
        pmc = json.load(open(pj))
        c = pmc.get("counters_summed_over_dispatches", {})
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            pmc["wave_time_split"] = {"issuing (SQ_ACTIVE_INST_ANY)": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                      "parked at s_waitcnt (SQ_WAIT_ANY)": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                                      "issue stalls (SQ_WAIT_INST_ANY)": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3)}
        if "TCC_HIT_sum" in c and c.get("TCC_REQ_sum"):
            pmc["l2_hit_rate"] = round(c["TCC_HIT_sum"] / c["TCC_REQ_sum"], 4)
        json.dump(pmc, open("profiles/%s_pmc_%s.json" % (tag, kern), "w"), indent=1)
    lj = "gpurun_out/cfg_%s_%s.json" % (tag, name)
    if os.path.exists(lj):
        for l in open(lj):
            if l.startswith("{"):
                d = json.loads(l)
                if pmc and "roofline" in d and "hbm_traffic_bytes_per_launch" in pmc:
                    d["roofline"]["traffic"] = pmc["hbm_traffic_bytes_per_launch"]
                    d["roofline"]["traffic_unit"] = "bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE KiB)"
                    d["roofline"]["traffic_source"] = "profiles/%s_pmc_%s.json" % (tag, kern)
                    d["roofline"]["traffic_over_algorithmic"] = pmc["hbm_traffic_bytes_per_launch"] / d["roofline"]["alg_bytes_total"]
                lines.append(d)
for f in glob.glob("gpurun_out/cfg_%s_general_long_stats/**/*kernel_stats.csv" % tag, recursive=True):
    shutil.copy(f, "profiles/%s_general_long_kernel_stats.csv" % tag)
for extra in ("cfg_%s_general_long.json" % tag, "cfg_%s_cyclic7_single.json" % tag):
    if os.path.exists("gpurun_out/" + extra):
        for l in open("gpurun_out/" + extra):
            if l.startswith("{"):
                lines.append(json.loads(l))
for src, dst in (("value_%s.json" % tag, "%s_bench_value.json" % tag), ("gym_%s.log" % tag, "%s_bench_gym.log" % tag), ("single_%s.log" % tag, "%s_bench_single.log" % tag)):
    if os.path.exists("gpurun_out/" + src):
        shutil.copy("gpurun_out/" + src, "profiles/" + dst)
for f in glob.glob("gpurun_out/prof_policy*/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, "profiles/%s_policy_%s_kernel_stats.csv" % (tag, "per_step" if "per_step" in f else "rollout"))
for f in sorted(glob.glob("gpurun_out/policy_%s_*.json" % tag)):
    for l in open(f):
        if l.startswith("{"):
            lines.append(json.loads(l))
if lines:
    with open("profiles/%s_bench_configs.jsonl" % tag, "w") as fh:
        for d in lines:
            fh.write(json.dumps(d) + "\n")
print("profiles/: " + " ".join(sorted(os.listdir("profiles"))))
