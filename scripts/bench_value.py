"""values/s of LeadMonomialsEnv.value(strategy, gamma) (buchberger.cpp:332-351; the README's training recipe calls it every
step, pg.py:461-462) for a whole batch at once on the device, next to the compiled reference's value() on one host core
(oracle/_ref; our C restatement where that library did not travel) on a sample of the same states.
    python scripts/bench_value.py [--dist 3-20-10-weighted] [--batch 4096] [--steps 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="3-20-10-weighted")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10, help="random-agent steps before the values are taken (a mix of episode phases)")
ap.add_argument("--cpu-envs", type=int, default=256, help="states the CPU baseline evaluates (sample: a sixteenth of that)")
ap.add_argument("--repeats", type=int, default=3)
a = ap.parse_args()
B = a.batch
env = VecLeadMonomialsEnv(a.dist, batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
env.rollout("random", a.steps, auto_reset=True)
kind = "reference" if ffi.available("ref") else "port"
lib = ffi.load("ref" if kind == "reference" else "bo")
cpu_envs = []
for e in range(min(B, a.cpu_envs)):
    o = lib.env(a.dist); o.seed(1000 + e); o.reset()
    for t in range(a.steps):
        o.step(ffi.agent_action(e, t, o.nP))
        if o.nP == 0:
            o.reset()
    cpu_envs.append(o)
out = {"dist": a.dist, "batch": B, "cpu_baseline_kind": kind, "strategies": {}}
for strategy in ("degree", "first", "normal", "sample"):
    env.values(strategy, 0.99)                                      # warm-up (allocations, code objects)
    t0 = time.perf_counter()
    for _ in range(a.repeats):
        got = env.values(strategy, 0.99)
    dt = (time.perf_counter() - t0) / a.repeats
    n_cpu = len(cpu_envs) // (16 if strategy == "sample" else 1)
    t0 = time.perf_counter()
    want = [cpu_envs[e].value(strategy, 0.99) for e in range(n_cpu)]
    dc = time.perf_counter() - t0
    if strategy != "sample":
        assert got[:n_cpu].tolist() == want, strategy               # (sample draws its own seeds on either side)
    out["strategies"][strategy] = {"gpu_values_per_s": B / dt, "gpu_ms_per_batch": dt * 1e3, "cpu_values_per_s": n_cpu / dc, "cpu_sample": n_cpu,
                                   "speedup_vs_one_core": (B / dt) / (n_cpu / dc)}
print(json.dumps(out))
