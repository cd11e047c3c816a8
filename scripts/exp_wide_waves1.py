"""Experiment: one workgroup per CU (B = 32), cyclic-7 random agent 512 steps, and the Degree run of one environment, against waves per environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
torch.cuda.init()
for nw in (8, 6, 4, 3, 2):
    env = VecLeadMonomialsEnv("cyclic-7", batch=32, k=2, caps={"queue_slots": 80, "wide_waves": nw})
    env.seed_agent(np.arange(32)); env.reset(); env.accounting(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout("random", 512, auto_reset=True)
    t1 = time.perf_counter()
    adds = (env.stats()[:, 1]).astype(np.float64)
    env1 = VecLeadMonomialsEnv("cyclic-7", batch=1, k=2, caps={"wide_waves": nw})
    env1.reset(); env1.accounting(False)
    t2 = time.perf_counter()
    env1.rollout("degree", 1 << 30, auto_reset=False)
    t3 = time.perf_counter()
    print("waves %d: B=32 random x512 %.2f s (max additions/env %.0f -> %.2f us per addition of the slowest); Degree run %.2f s" % (
        nw, t1 - t0, adds.max(), (t1 - t0) / adds.max() * 1e6, t3 - t2), flush=True)
