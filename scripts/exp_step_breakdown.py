"""Diagnostic: what a ONE-step launch of the fast class costs and why (3-20-10-weighted, 4096 environments, external actions):
launches right after a reset (no environment ends an episode inside them) against launches in the steady state (~3 % of the
environments draw a new ideal — ten insertions — inside any given step), and an empty launch (nsteps = 0: records in and out only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgroebner_amd import VecLeadMonomialsEnv
B, R = 4096, 512
env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.accounting(False)
obs = torch.empty((B, R, env.cols), dtype=torch.int32, device="cuda")
rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
rows = torch.zeros(B, dtype=torch.int32, device="cuda"); act = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream()
def one(n=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n):
        env.step_device(act, rew, done, rows, obs, R, 0, s.cuda_stream, auto_reset=True)
    e1.record(s); env.sync(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, int(done.sum().item())
fresh = []
for rep in range(20):
    env.reset(); env.sync()
    one(); fresh.append(one())                              # second step after a reset: nobody is done yet
env.rollout_device("random", 300, True, s.cuda_stream, rew, done, rows, obs, R, False, True); env.sync()
steady = [one() for _ in range(40)]
chain = one(200)
print("one-step launch right after a reset : %.1f us (episodes ending inside: %.1f)" % (np.mean([f[0] for f in fresh]), np.mean([f[1] for f in fresh])))
print("one-step launch in the steady state : %.1f us (episodes ending inside: %.1f of %d)" % (np.mean([f[0] for f in steady]), np.mean([f[1] for f in steady]), B))
print("200 one-step launches back to back  : %.1f us per launch" % chain[0])
