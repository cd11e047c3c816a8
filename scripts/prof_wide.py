"""Diagnostic (BBX_PROF_BUILD library only): per-phase cycle shares of the wide kernel.
usage: prof_wide.py DIST BATCH STEPS [AGENT_SEED0]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
dist, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
env = VecLeadMonomialsEnv(dist, batch=B, k=2)
env.seed_agent(np.arange(B) + seed0); env.reset()
if len(sys.argv) > 5 and sys.argv[5] == "lean":
    env.accounting(False)
lib = _ffi.lib()
acc = (C.c_ulonglong * 32)()
lib.bbx_wide_prof_read(acc, 1)
t0 = time.perf_counter()
env.rollout("random", T, auto_reset=True)
dt = time.perf_counter() - t0
lib.bbx_wide_prof_read(acc, 1)
a = np.array(list(acc), dtype=np.float64)
names = {0: "loop top/reset/table", 1: "select+removal+spoly setup", 2: "lead term + divisor scan", 3: "tail moves / loop exit", 4: "reducer metadata",
         5: "add_scaled -> LDS", 6: "add_scaled -> HBM", 7: "leader basis update", 8: "obs + bookkeeping"}
names.update({20: "  tier 1: reducer tail -> F (load + barrier)", 21: "  tier 1: chunk range search", 22: "  tier 1: merge", 5: "add_scaled -> LDS (rest)"})
tot = sum(a[i] for i in names)
st = env.stats()
print("seconds %.3f  steps %d additions %d  max additions/env %d (env %d)" % (dt, st[:, 0].sum(), st[:, 1].sum(), st[:, 1].max(), int(st[:, 1].argmax())))
for i, n in names.items():
    print("  %-28s %6.2f %%" % (n, 100 * a[i] / tot))
print("seconds per million: scans %.2f  tier-1 merges %.2f  tail loads %.2f" % (dt * a[2] / tot / max(1, a[10] + a[11] + a[12] + a[13]) * 1e6 / B, dt * a[22] / tot / max(1, a[14]) * 1e6 / B, dt * a[20] / tot / max(1, a[14]) * 1e6 / B))
print("H rewrites: tier1 %d tier2 %d tier3 %d  accumulator merges %d  tail-moves %d | chunks %d mean an %.0f mean bn %.0f | bring-backs %d spills %d" %
      (a[10], a[11], a[12], a[15], a[13], a[14], a[16] / max(1, a[10]), a[17] / max(1, a[10]), a[18], a[19]))
print("cycles per reduction round (all phases): %.0f" % (tot / max(1, a[10] + a[11] + a[12])))
