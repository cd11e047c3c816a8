"""Diagnostic: cost of a K-step launch of the fast class as intercept + slope (3-20-10-weighted, 4096 environments),
for the headline kernel (hash agent, no fill) and the generic kernel (external actions, incremental padding)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgroebner_amd import VecLeadMonomialsEnv
B, R = 4096, 256
env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
obs = torch.empty((B, R, env.cols), dtype=torch.int32, device="cuda")
rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
rows = torch.zeros(B, dtype=torch.int32, device="cuda"); act = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream()
env.rollout_device("random", 300, True, s.cuda_stream, rew, done, rows, obs, R, True, True); env.sync()
def timed(fn, n=200):
    fn(); env.sync()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    env.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for K in (1, 2, 4, 8, 16, 64):
    t = timed(lambda: env.rollout_device("random", K, True, s.cuda_stream, rew, done, rows, obs, R, False, True))
    print("headline kernel  K=%3d  %.1f us per launch  %.2f us per step" % (K, t, t / K))
for fill in (0, 1, 2):
    t = timed(lambda: env.step_device(act, rew, done, rows, obs, R, fill, s.cuda_stream, auto_reset=True))
    print("generic kernel, external action (always row 0), obs_fill=%d: %.1f us per step launch" % (fill, t))
t = timed(lambda: env.step_device(act, rew, done, rows, None, R, 0, s.cuda_stream, auto_reset=True))
print("generic kernel, no observation: %.1f us" % t)
