"""Probe: which binomial distributions reach pair sets of more than 1024 rows (for the policy kernels' row-limit test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
for dist in sys.argv[1:]:
    env = VecLeadMonomialsEnv(dist, batch=256, k=1)
    env.seed(np.arange(256) + 300); env.seed_agent(np.arange(256)); env.reset(); env.accounting(False)
    peak, when = 0, 0
    for c in range(30):
        env.rollout("random", 100, auto_reset=True)
        m = int(env.rows.max())
        if m > peak: peak, when = m, (c + 1) * 100
        if m > 1200: break
    print(dist, "peak rows at a chunk end", peak, "after", when, "steps; env", int(env.rows.argmax()), flush=True)
