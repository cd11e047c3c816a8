"""Reproduction of a fuzz_gym failure: 5-3-3-0.5-uniform, LCM elimination, seeds 867467.., host-stepped with the oracle beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi
bo = ffi.load("bo")
dist, elim, rew, seed0, B, T, k = "5-3-3-0.5-uniform", "lcm", "additions", 867467, 5, 30, 2
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1          # step only this environment on the device (others: action on a finished env is ignored)
arng = np.random.default_rng(seed0)
env = VecLeadMonomialsEnv(dist, B, elim, rew, False, True, k, 0, None, "python")
env.seed(np.arange(B) + seed0)
os_ = []
for e in range(B):
    o = bo.env(dist, elimination=elim, rewards=rew); o.seed(seed0 + e); o.reset(); os_.append(o)
obs = env.reset()
for t in range(T):
    acts = np.array([arng.integers(0, max(1, os_[e].nP)) for e in range(B)], dtype=np.int32)
    live = [os_[e].nP > 0 for e in range(B)]
    if not any(live):
        break
    t0 = time.perf_counter()
    try:
        obs, r, d, _ = env.step(acts)
    except Exception as ex:
        print("step", t, "FAILED:", str(ex)[:200], env.capacities(), "oracle nG/nP", [(o.nG, o.nP) for o in os_]); break
    dt = time.perf_counter() - t0
    mask = np.zeros(B, dtype=np.uint8)
    want = []
    for e in range(B):
        if not live[e]:
            want.append(0.0); continue
        want.append(os_[e].step(int(acts[e])))
        if os_[e].nP == 0 and arng.random() < 0.7:
            mask[e] = 1
    print("step %2d: %.2f s device; rewards device %s oracle %s mask %s %s" % (t, dt, [float(x) for x in r], want, list(mask), env.capacities()), flush=True)
    if [float(x) for x in r] != want:
        print("MISMATCH"); break
    if mask.any():
        obs = env.reset(mask)
        for e in np.flatnonzero(mask):
            os_[e].reset()
