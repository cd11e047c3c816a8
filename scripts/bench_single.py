"""Latency of the single-environment drop-in (what scripts/train.py of the reference drives): env.step(action) per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import CLeadMonomialsEnv
env = CLeadMonomialsEnv("3-20-10-weighted", k=2)
env.seed(123)
s = env.reset()
n = 0; t0 = time.perf_counter()
while n < 3000:
    s, r, d, _ = env.step(0)
    n += 1
    if d:
        s = env.reset()
t1 = time.perf_counter()
print("single-environment drop-in: %.1f us per step+observation, %.0f steps/s" % ((t1 - t0) / n * 1e6, n / (t1 - t0)))
