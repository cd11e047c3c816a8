"""Child process of tests/test_multirank_gpu.py: one rank of a world, its shard of the global batch on GPU `device`.
Prints one JSON line {rank, stats: [[steps, additions, episodes, zero_reductions, basis_size] per environment]}."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world, B, T, device = (int(x) for x in sys.argv[1:6])
from deepgroebner_amd import VecLeadMonomialsEnv      # noqa: E402
from deepgroebner_amd.shard import plan                # noqa: E402

pl = plan(rank, world, B)
env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2, device=device)
env.seed(pl["ideal_seeds"]); env.seed_agent(pl["agent_seeds"]); env.reset()
env.rollout("random", T, auto_reset=True)
st = env.stats()
print(json.dumps({"rank": rank, "ids": pl["ids"].tolist(), "stats": st[:, [0, 1, 2, 3, 7]].tolist()}))
