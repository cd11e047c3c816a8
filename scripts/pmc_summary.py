"""Summarise rocprofv3 --pmc passes written by scripts/pmc_passes.sh: counters of the longest step-kernel dispatch."""
import collections, csv, glob, json, sys
tag = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else None     # kernel-name substring (default: the longest step-kernel dispatch)
out = {}
kname = None
for f in sorted(glob.glob('gpurun_out/pmc_%s_*/p_counter_collection.csv' % tag)):
    by = collections.defaultdict(dict)
    names = {}
    dur = {}
    for r in csv.DictReader(open(f)):
        if want is not None and want not in r['Kernel_Name']:
            continue
        if 'bbx_' in r['Kernel_Name'] and any(t in r['Kernel_Name'] for t in ('step_kernel', 'binom_kernel', 'fast_kernel', 'fast_headline_kernel')):
            by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
            names[r['Dispatch_Id']] = r['Kernel_Name']
            dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if not by:
        continue
    d = max(dur, key=dur.get)
    out.update(by[d]); kname = names[d]; out.setdefault('_dispatch_ns', []).append(dur[d])
print(json.dumps({"kernel": kname, "counters": out}, indent=1))
