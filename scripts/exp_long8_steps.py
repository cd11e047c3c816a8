"""Experiment: where the 8-variable long-polynomial corner spends its time — per-step seconds, additions and record growth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
env = VecLeadMonomialsEnv("8-3-4-0.5-uniform", batch=1, k=2, caps=None if len(sys.argv) < 2 else {"max_poly_terms": int(sys.argv[1])})
env.seed(np.array([1037])); env.seed_agent(np.array([37])); env.reset()
tot = 0.0
prev = env.stats()[0].copy(); g0 = env.capacities()["grown"]
for t in range(64):
    t0 = time.perf_counter(); env.rollout("random", 1, auto_reset=True); dt = time.perf_counter() - t0
    st = env.stats()[0]; g = env.capacities()["grown"]
    tot += dt
    if dt > 0.05 or g != g0:
        print("step %2d: %.3f s, %6d additions, basis %d, pairs %d, grown %d %s" % (t, dt, st[1] - prev[1], st[7], env.rows[0], g, env.capacities()), flush=True)
    prev = st.copy(); g0 = g
print("total %.2f s" % tot)
