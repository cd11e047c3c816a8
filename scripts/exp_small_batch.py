"""Experiment: host-stepped small batches (a Gym-style loop over VecLeadMonomialsEnv.step_ragged), microseconds per vector step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
for B in (1, 8, 16, 32, 64):
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
    env.seed(np.arange(B) + 5); env.reset(); env.accounting(False)
    acts = np.zeros(B, dtype=np.int32)
    for t in range(300):
        env.step_ragged(acts, auto_reset=True)
    t0 = time.perf_counter()
    n = 3000
    for t in range(n):
        flat, off, r, d = env.step_ragged(acts, auto_reset=True)
    t1 = time.perf_counter()
    print("B=%2d: %.1f us per vector step, %.0f k env-steps/s  %s" % (B, (t1 - t0) / n * 1e6, B * n / (t1 - t0) / 1e3, env.session_stats()), flush=True)
