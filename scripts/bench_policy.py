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
from deepgroebner_amd.rollout import DeviceTrajectoryBuffer, PMLPPolicy, run_rollout, run_rollout_fused

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="3-20-10-weighted")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--k", type=int, default=2)
ap.add_argument("--hidden", default="128", help="hidden layer sizes, e.g. 128 or 128,128 (more than one layer: always --per-step; two or three layers "
                                              "of <= 128 units: bbx_pmlp2_act / bbx_pmlp3_act in front of bbx_step_device_autoreset, else: torch ops)")
ap.add_argument("--graph", action="store_true", help="deeper policies: replay the vector step from a HIP graph (run_rollout(graph=True))")
ap.add_argument("--obs-rows", type=int, default=256)
ap.add_argument("--store", action="store_true", help="also record the trajectory (actions, rewards, log-probabilities, dones) on the device")
ap.add_argument("--per-step", action="store_true", help="one library call per vector step (bbx_policy_step_device) instead of the policy rollout "
                                                       "kernel (bbx_policy_rollout_device: --chunk steps per launch, policy inside the step loop)")
ap.add_argument("--chunk", type=int, default=256)
ap.add_argument("--store-states", action="store_true", help="--store plus the observation block of every step")
ap.add_argument("--no-persistent", action="store_true", help="--per-step: one kernel per call instead of calls that join a persistent session")
a = ap.parse_args()
hidden = [int(h) for h in str(a.hidden).split(",")]
if len(hidden) > 1:
    a.per_step = True; a.no_persistent = True
torch.manual_seed(0)
B = a.batch
env = VecLeadMonomialsEnv(a.dist, batch=B, k=a.k)
env.seed(np.arange(B) + 1000); env.reset()
env.accounting(False)
policy = PMLPPolicy(env.cols, hidden).cuda()
if a.per_step and not a.no_persistent and not a.store and not a.store_states:
    env.persistent(True)      # (per-step outputs are then only final at sync(): nothing in this mode reads them in between)
def go(nsteps, buf):
    if a.per_step:
        return run_rollout(env, policy, nsteps, buffer=buf, obs_rows=a.obs_rows, graph=a.graph)
    return run_rollout_fused(env, policy, nsteps, buffer=buf, obs_rows=a.obs_rows, chunk=a.chunk)
go(300, None)                                                      # steady state + warm-up
st0 = env.stats()
store = a.store or a.store_states
buf = DeviceTrajectoryBuffer(a.steps, B, obs_shape=(a.obs_rows, env.cols) if a.store_states else None) if store else None
torch.cuda.synchronize()
t0 = time.perf_counter()
total, episodes = go(a.steps, buf)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = env.stats()
d = st - st0
assert (d[:, 0] == a.steps).all() and (st[:, 4] == 0).all()
assert int(episodes.sum()) == int(d[:, 2].sum()) and float(total.sum()) == -float(d[:, 1].sum())
print(json.dumps({"dist": a.dist, "batch": B, "steps": a.steps, "policy": "PMLP(%s)" % hidden,
                  "mode": ("policy kernel (bbx_pmlp%d_act) + bbx_step_device_autoreset per step" % len(hidden) if policy.deep_ok(env.cols) else
                           "torch ops + bbx_step_device_autoreset per step" + (", replayed from a HIP graph" if a.graph else "")) if len(hidden) > 1 else ("one call per step (bbx_policy_step_device)" + (", calls joining persistent sessions" if a.per_step and not a.no_persistent and not a.store and not a.store_states else "")) if a.per_step else "policy rollout kernel, %d steps per launch" % a.chunk,
                  "store": "trajectory + states" if a.store_states else ("trajectory" if a.store else "nothing"),
                  "env_steps_per_s": B * a.steps / dt, "us_per_vector_step": dt / a.steps * 1e6,
                  "mean_return_per_episode": float(total.sum()) / max(1, int(episodes.sum())), "episodes": int(episodes.sum())}))
