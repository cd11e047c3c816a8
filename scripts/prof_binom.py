"""Diagnostic (BBX_PROF_BUILD library only): per-phase cycle shares of the HBM-resident binomial kernel.
usage: prof_binom.py DIST BATCH STEPS"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
dist, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
env = VecLeadMonomialsEnv(dist, batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
lib = _ffi.lib(); acc = (C.c_ulonglong * 32)()
obs = torch.empty((B, 2048, env.cols), dtype=torch.int32, device="cuda")
rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda"); rows = torch.zeros(B, dtype=torch.int32, device="cuda")
lib.bbx_bin_prof_read(acc, 1)
t0 = time.perf_counter()
env.rollout_device("random", T, True, torch.cuda.current_stream().cuda_stream, rew, done, rows, obs, 2048, False, True); env.sync()
dt = time.perf_counter() - t0
lib.bbx_bin_prof_read(acc, 1)
a = np.array(list(acc), dtype=np.float64)
names = {0: "loop top / reset", 1: "agent + pair removal", 2: "S-polynomial", 3: "reduce", 8: "update: old-pair filter", 9: "update: lcm pass", 10: "update: peel",
         11: "update: emit", 4: "reducer insert", 5: "bookkeeping", 6: "observation"}
tot = sum(a[i] for i in names)
print("%s B=%d T=%d: %.3f s = %.1f M env-steps/s" % (dist, B, T, dt, B * T / dt / 1e6))
for i, n in names.items():
    print("  %-26s %6.2f %%" % (n, 100 * a[i] / tot))
