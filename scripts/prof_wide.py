"""Diagnostic (BBX_PROF_BUILD library only): per-phase cycle shares of the wide kernel.
usage: prof_wide.py DIST BATCH STEPS [AGENT_SEED0] [lean|acct] [AGENT]     (STEPS <= 0: to completion, no auto-reset)"""
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
agent = sys.argv[6] if len(sys.argv) > 6 else "random"
if T > 0:
    env.rollout(agent, T, auto_reset=True)
else:
    env.rollout(agent, 1 << 30, auto_reset=False)
dt = time.perf_counter() - t0
lib.bbx_wide_prof_read(acc, 1)
a = np.array(list(acc), dtype=np.float64)
names = {2: "lead term + divisor scan", 4: "reducer metadata", 20: "reducer tail -> F (load + barrier)", 22: "tier-1 merge",
         5: "rest of add_scaled", 8: "everything else"}
tot = sum(a[i] for i in names)
st = env.stats()
print("seconds %.3f  steps %d additions %d  max additions/env %d (env %d)" % (dt, st[:, 0].sum(), st[:, 1].sum(), st[:, 1].max(), int(st[:, 1].argmax())))
for i, n in names.items():
    print("  %-36s %6.2f %%" % (n, 100 * a[i] / tot))
scans = max(1.0, a[10] + a[13])
print("tier-1 rewrites %d  accumulator merges %d  tail-moves %d | scans: trips/scan %.2f  mean found index %.0f  mean |G| %.0f" %
      (a[10], a[15], a[13], a[23] / scans, a[24] / max(1.0, scans - a[13]), a[25] / scans))
print("cycles (100 MHz s_memtime ticks x 21) per reduction round: %.0f ticks" % (tot / scans))
mn = ["search of the diagonal", "boundary exchange + barrier", "sequential merge", "ballots + count exchange + barrier", "stores"]
mt = sum(a[26 + i] for i in range(5))
if mt > 0:
    print("inside the merges (%.1f %% of the phases above):" % (100 * mt / tot))
    for i in range(5):
        print("  %-36s %6.2f %%" % (mn[i], 100 * a[26 + i] / mt))
