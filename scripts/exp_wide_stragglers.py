"""Experiment: is cyclic-7 B x 512 steps bound by its slowest environment?  Time against batch size and the spread of work per environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
torch.cuda.init()
T = 512
for B in (32, 128, 256, 512, 1024):
    env = VecLeadMonomialsEnv("cyclic-7", batch=B, k=2, caps={"queue_slots": 80})
    env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
    st0 = env.stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout("random", T, auto_reset=True)
    t1 = time.perf_counter()
    d = env.stats() - st0
    adds = d[:, 1].astype(np.float64)
    print("B=%4d: %.2f s, %.1f k env-steps/s; additions per env: mean %.0f, max %.0f (%.2fx), min %.0f; p90 %.0f" % (
        B, t1 - t0, B * T / (t1 - t0) / 1e3, adds.mean(), adds.max(), adds.max() / adds.mean(), adds.min(), np.percentile(adds, 90)), flush=True)
    del env
