"""End-to-end env-steps/s with a policy in the loop: every vector step is [fused PMLP policy kernel:
log-softmax over the rows of the padded observation block + inverse-CDF draw] -> [bbx_step_device_autoreset: one step of
every environment, next padded block] on the device; the host only enqueues kernels.

    python scripts/bench_policy.py [--batch 4096] [--steps 2000] [--hidden 128] [--store]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from deepgroebner_amd.rollout import DeviceTrajectoryBuffer, PMLPPolicy, run_rollout

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="3-20-10-weighted")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--k", type=int, default=2)
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--obs-rows", type=int, default=128)
ap.add_argument("--store", action="store_true", help="also record the trajectory (actions, rewards, log-probabilities, dones) on the device")
ap.add_argument("--graph", action="store_true", help="capture the two-kernel step in a hipGraph")
a = ap.parse_args()
torch.manual_seed(0)
B = a.batch
env = VecLeadMonomialsEnv(a.dist, batch=B, k=a.k)
env.seed(np.arange(B) + 1000); env.reset()
env.accounting(False)
policy = PMLPPolicy(env.cols, [a.hidden]).cuda()
run_rollout(env, policy, 300, obs_rows=a.obs_rows)                 # steady state + warm-up
st0 = env.stats()
buf = DeviceTrajectoryBuffer(a.steps, B, obs_shape=None) if a.store else None
torch.cuda.synchronize()
t0 = time.perf_counter()
total, episodes = run_rollout(env, policy, a.steps, buffer=buf, obs_rows=a.obs_rows)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = env.stats()
d = st - st0
assert (d[:, 0] == a.steps).all() and (st[:, 4] == 0).all()
assert int(episodes.sum()) == int(d[:, 2].sum()) and float(total.sum()) == -float(d[:, 1].sum())
print(json.dumps({"dist": a.dist, "batch": B, "steps": a.steps, "policy": "PMLP([%d]) fused act kernel" % a.hidden, "store": bool(a.store),
                  "env_steps_per_s": B * a.steps / dt, "us_per_vector_step": dt / a.steps * 1e6,
                  "mean_return_per_episode": float(total.sum()) / max(1, int(episodes.sum())), "episodes": int(episodes.sum())}))
