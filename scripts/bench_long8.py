"""The long-polynomial corner of the reference's N = 8 rings: `8-3-4-0.5-uniform`, environment seed 1037
builds intermediate polynomials of more than 32 768 terms within 64 steps.  Times that environment alone and a batch around
it on the GPU (general class -> wide class hand-off for 32-byte monomials) and, with --cpu, the compiled reference on one core.
    python scripts/bench_long8.py [--batch 64] [--steps 64] [--cpu]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi
ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="8-3-4-0.5-uniform")
ap.add_argument("--seed", type=int, default=1037)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--no-wide", action="store_true", help="wave-per-environment kernel only (caps wide_waves = -1)")
a = ap.parse_args()
out = {"dist": a.dist, "seed": a.seed, "steps": a.steps}
for B in sorted({1, a.batch}):
    env = VecLeadMonomialsEnv(a.dist, batch=B, k=2, caps={"wide_waves": -1} if a.no_wide else None)
    env.seed(np.arange(B) + a.seed); env.seed_agent(np.arange(B) + a.seed - 1000); env.reset()
    t0 = time.perf_counter()
    env.rollout("random", a.steps, auto_reset=True)
    dt = time.perf_counter() - t0
    st = env.stats()
    out["gpu_B%d" % B] = {"seconds": dt, "additions": int(st[:, 1].sum()), "max_additions_one_env": int(st[:, 1].max()), "grown": env.capacities()["grown"]}
if a.cpu:
    lib = ffi.load("ref" if ffi.available("ref") else "bo")
    r = lib.bench_random(a.dist, 2, 1, a.steps, a.seed, a.seed - 1000)
    out["cpu_one_env"] = {"seconds": r["seconds"], "additions": r["additions"], "kind": lib.kind}
print(json.dumps(out))
