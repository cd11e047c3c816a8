"""Clones per second of the tree-search primitives (mcts.py:89,96,147 / az.py:82 call env.copy() once per search node):
bbx_copy (a whole batch into a new handle) and bbx_clone_envs (environments copied over others inside one batch: the
node pool of a batched search), next to the copy constructor of the compiled reference's LeadMonomialsEnv on one host core
(oracle/_ref; our C restatement where that library did not travel).  Mid-episode states of the distribution.
    python scripts/bench_clone.py [--dist 3-20-10-weighted] [--batch 4096] [--steps 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="3-20-10-weighted")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10, help="random-agent steps before the copies are taken")
ap.add_argument("--cpu-envs", type=int, default=256)
ap.add_argument("--repeats", type=int, default=20)
a = ap.parse_args()
B = a.batch
env = VecLeadMonomialsEnv(a.dist, batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
env.rollout("random", a.steps, auto_reset=True)
st0 = env.stats().copy()

# whole-batch copies (each is a new handle: allocation + device-to-device copy of every record)
c = env.copy(); del c
t0 = time.perf_counter()
for _ in range(a.repeats):
    c = env.copy()
    c.sync()
    del c
dt_copy = (time.perf_counter() - t0) / a.repeats

# in-batch clones: the first B/64 environments are the roots, each copied over 63 others (one call)
roots = B // 64 if B >= 64 else 1
src = np.repeat(np.arange(roots), (B - roots) // roots).astype(np.int32)
dst = (roots + np.arange(len(src))).astype(np.int32)
env.clone_envs(src, dst); env.sync()
t0 = time.perf_counter()
for _ in range(a.repeats):
    env.clone_envs(src, dst)
env.sync()
dt_clone = (time.perf_counter() - t0) / a.repeats
st = env.stats()
assert np.array_equal(st[dst][:, [7]], st[src][:, [7]])              # (basis sizes of the clones are their roots')
# the drop-in form: ONE environment, copy() as mcts.py calls it per node (a new one-environment handle each time)
from deepgroebner_amd import CLeadMonomialsEnv
one = CLeadMonomialsEnv(a.dist, k=2); one.seed(1000); one.reset()
for t in range(a.steps):
    _, _, done, _ = one.step(0)                       # (first pair: any mid-episode state will do)
    if done:
        one.reset()
c1 = one.copy(); del c1
t0 = time.perf_counter()
for _ in range(200):
    c1 = one.copy()
    del c1
dt_one = (time.perf_counter() - t0) / 200

kind = "reference" if ffi.available("ref") else "port"
lib = ffi.load("ref" if kind == "reference" else "bo")
cpu = []
for e in range(min(B, a.cpu_envs)):
    o = lib.env(a.dist); o.seed(1000 + e); o.reset()
    for t in range(a.steps):
        o.step(ffi.agent_action(e, t, o.nP))
        if o.nP == 0:
            o.reset()
    cpu.append(o)
t0 = time.perf_counter()
copies = [o.copy() for o in cpu for _ in range(4)]
dc = time.perf_counter() - t0
print(json.dumps({"dist": a.dist, "batch": B, "steps_before": a.steps, "cpu_baseline_kind": kind,
                  "bbx_copy": {"ms_per_batch": dt_copy * 1e3, "clones_per_s": B / dt_copy},
                  "single_env_copy": {"us_per_copy": dt_one * 1e6, "clones_per_s": 1.0 / dt_one,
                                      "note": "CLeadMonomialsEnv.copy(): a new one-environment handle per copy (allocation + record copy)"},
                  "bbx_clone_envs": {"clones_per_call": int(len(src)), "ms_per_call": dt_clone * 1e3, "clones_per_s": len(src) / dt_clone},
                  "cpu_copy_constructor": {"clones_per_s": len(copies) / dc, "sample": len(copies), "cores": 1,
                                           "note": "ctypes call overhead included (about 1 us per call)"}}))
