"""Deeper policies (ParallelMultilayerPerceptron(hidden_layers=[h1, h2, ...]), networks.py:562-571) have no fused kernel: the
policy runs as torch ops on the observation block, then one bbx_step_device call.  This script measures that path eagerly
and with the whole vector step (policy ops + the step kernels) captured once in a HIP graph and replayed.
    python scripts/exp_graph_policy.py [--hidden 128,128] [--batch 4096] [--steps 500]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from deepgroebner_amd.rollout import PMLPPolicy, run_rollout

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="3-20-10-weighted")
ap.add_argument("--hidden", default="128,128")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--obs-rows", type=int, default=256)
a = ap.parse_args()
hidden = [int(h) for h in a.hidden.split(",")]
B = a.batch
torch.manual_seed(0)


def fresh():
    env = VecLeadMonomialsEnv(a.dist, batch=B, k=2)
    env.seed(np.arange(B) + 1000); env.reset(); env.accounting(False)
    return env


env = fresh()
policy = PMLPPolicy(env.cols, hidden).cuda()
out = {"dist": a.dist, "batch": B, "steps": a.steps, "policy": "PMLP(%s)" % hidden}
for graph in (False, True):
    env = fresh()
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    run_rollout(env, policy, 32, obs_rows=a.obs_rows, generator=g, graph=graph)          # warm-up (and the capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total, episodes = run_rollout(env, policy, a.steps, obs_rows=a.obs_rows, generator=g, graph=graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["graph" if graph else "eager"] = {"env_steps_per_s": B * a.steps / dt, "us_per_vector_step": dt / a.steps * 1e6,
                                          "total_reward": float(total.sum().item()), "episodes": int(episodes.sum().item())}
out["same_result"] = out["graph"]["total_reward"] == out["eager"]["total_reward"] and out["graph"]["episodes"] == out["eager"]["episodes"]
print(json.dumps(out))
