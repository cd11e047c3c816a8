"""Randomised parity sweep (test infrastructure, GPU box): random distribution strings, batch sizes, horizons, k and kernel
capacities; every environment's counters and final state against the CPU restatement (oracle/, counter-hash agent).
    python scripts/fuzz_parity.py [ROUNDS] [SEED]          (FUZZ_LARGE=1: batches of 1024 / 4096 environments; FUZZ_LONG=1: 10x horizons)
Prints one line per case; exits non-zero at the first mismatch — or capacity error: records grow on demand, an environment
the oracle could finish must finish here."""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi
from test_gpu_parity import fnv64, _state_words

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bo = ffi.load("bo")
torch.cuda.init()
t_start = time.time()
for it in range(rounds):
    n = rng.choice([2, 3, 3, 3, 4, 5, 5, 6, 7, 8])
    kind = rng.choice(["binom", "binom", "binom", "poly"])
    if kind == "binom":
        d = rng.randint(2, 12 if n <= 3 else 6); s = rng.randint(2, 10 if n <= 3 else 5)
        dist = "%d-%d-%d-%s" % (n, d, s, rng.choice(["uniform", "weighted", "maximum"]))
        T = rng.choice([30, 80, 200]) if n <= 3 else rng.choice([20, 60])
    else:
        d = rng.randint(2, 4); s = rng.randint(2, 4); lam = rng.choice([0.3, 0.5, 1.0])
        dist = "%d-%d-%d-%s-%s" % (n, d, s, lam, rng.choice(["uniform", "weighted"]))
        T = rng.choice([10, 25])
    for flag in ("consts", "homog", "pure"):
        if rng.random() < 0.15:
            dist += "-" + flag
    if os.environ.get("FUZZ_LONG"):
        T *= 10                                             # FUZZ_LONG=1: ten times the horizon
    k = rng.choice([1, 2, 2, 3])
    B = rng.choice([1024, 4096]) if os.environ.get("FUZZ_LARGE") else rng.choice([1, 3, 8, 33, 200])   # FUZZ_LARGE=1: full-size batches
    caps = rng.choice([None, None, {"lds_max_basis": 16}, {"lds_max_basis": -1}, {"general_class": 1}])
    lean = rng.random() < 0.5
    seed0 = rng.randint(0, 10 ** 6)
    try:
        want = bo.run_random_many(dist, k, range(seed0, seed0 + B), range(B), T, True, 0)
    except ValueError:
        print("skip %s (oracle rejects it)" % dist); continue
    try:
        env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps)
    except Exception as ex:
        print("skip %s: %s" % (dist, str(ex)[:80])); continue
    try:
        env.seed(np.arange(B) + seed0); env.seed_agent(np.arange(B)); env.reset()
        if lean:
            env.accounting(False)
        env.rollout("random", T, auto_reset=True)
    except Exception as ex:
        msg = str(ex)
        if "error -3" in msg:                              # capacities grow on demand (bbx_caps.no_growth = 0): the reference would
            print("CAPACITY %s k=%d B=%d T=%d caps=%s seed0=%d: %s" % (dist, k, B, T, caps, seed0, msg[:160]))   # have continued, so this is a failure
            sys.exit(1)
        if "error -4" in msg:                              # the generator fails where the reference throws
            print("generator %s: %s" % (dist, msg[:100])); continue
        print("ERROR %s k=%d B=%d T=%d caps=%s lean=%s seed0=%d: %s" % (dist, k, B, T, caps, lean, seed0, msg[:200]))
        sys.exit(1)
    st = env.stats()
    for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
        w = np.array([r[key] for r in want])
        if not np.array_equal(st[:, col], w):
            e = int(np.flatnonzero(st[:, col] != w)[0])
            print("MISMATCH %s k=%d B=%d T=%d caps=%s lean=%s seed0=%d: %s of env %d: device %d oracle %d" % (dist, k, B, T, caps, lean, seed0, key, e, st[e, col], w[e]))
            sys.exit(1)
    for e in sorted(set(range(0, B, 7)) | {B - 1}):
        basis, pairs, order = env.state(e)
        if fnv64(_state_words(basis, pairs, order)) != want[e]["state_hash"]:
            print("MISMATCH %s k=%d B=%d T=%d caps=%s lean=%s seed0=%d: final state of env %d" % (dist, k, B, T, caps, lean, seed0, e))
            sys.exit(1)
    print("ok %-28s k=%d B=%-3d T=%-3d caps=%-22s lean=%d  steps %d additions %d episodes %d" % (dist, k, B, T, caps, lean, st[:, 0].sum(), st[:, 1].sum(), st[:, 2].sum()))
    del env
print("fuzz: %d rounds, %.0f s, no mismatch" % (rounds, time.time() - t_start))
