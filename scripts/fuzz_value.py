"""Randomised parity sweep of value() and copy() (test infrastructure, GPU box): environments stepped to a random point of
an episode with random actions; value(strategy, gamma) of every environment (device rollouts from clones) equals the CPU
restatement's double for the deterministic strategies; a copy taken there and the original then continue identically, and
in-batch clones behave like their sources.      python scripts/fuzz_value.py [ROUNDS] [SEED]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bo = ffi.load("bo")
t0 = time.time()
for it in range(rounds):
    n = rng.choice([2, 3, 3, 3, 4, 5])
    if rng.random() < 0.75:
        dist = "%d-%d-%d-%s" % (n, rng.randint(2, 8 if n <= 3 else 4), rng.randint(2, 7 if n <= 3 else 4), rng.choice(["uniform", "weighted", "maximum"]))
    else:
        dist = "%d-%d-%d-%s-uniform" % (min(n, 4), rng.randint(2, 3), rng.randint(2, 3), rng.choice(["0.3", "0.5"]))
    k = rng.choice([1, 2])
    B = rng.choice([1, 4, 12])
    caps = rng.choice([None, None, {"lds_max_basis": 16}, {"lds_max_basis": -1}, {"general_class": 1}])
    pre = rng.randint(0, 25)
    seed0 = rng.randint(0, 10 ** 6)
    lean = rng.random() < 0.5                            # (lean kernels: the host steps of small batches then run through mailbox sessions)
    arng = np.random.default_rng(seed0)
    tag = "%s k=%d B=%d pre=%d caps=%s lean=%d seed0=%d" % (dist, k, B, pre, caps, lean, seed0)
    try:
        env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps)
        env.seed(np.arange(B) + seed0)
        if lean:
            env.accounting(False)
        oracles = []
        for e in range(B):
            o = bo.env(dist); o.seed(seed0 + e); o.reset(); oracles.append(o)
        env.reset()
        for t in range(pre):
            acts = np.array([arng.integers(0, max(1, oracles[e].nP)) for e in range(B)], dtype=np.int32)
            if not any(o.nP > 0 for o in oracles):
                break
            env.step(acts)
            for e in range(B):
                if oracles[e].nP > 0:
                    oracles[e].step(int(acts[e]))
        for strategy in rng.sample(["first", "degree", "normal", "sugar"], 2):
            gamma = rng.choice([0.99, 0.9, 1.0, 0.5])
            got = env.values(strategy, gamma)
            for e in range(B):
                want = oracles[e].value(strategy, gamma)
                if got[e] != want:
                    print("MISMATCH %s: value(%s, %s) of env %d: device %r oracle %r" % (tag, strategy, gamma, e, got[e], want)); sys.exit(1)
        # in-batch clones (a tree search's node pool): environment src[i] over dst[i]; both continue alike afterwards
        if B >= 4 and rng.random() < 0.6:
            nc = rng.randint(1, B // 2)
            perm = list(range(B)); rng.shuffle(perm)
            src, dst = perm[:nc], perm[nc:2 * nc]
            env.clone_envs(src, dst)
            for a, b2 in zip(src, dst):
                oracles[b2] = oracles[a].copy()
        # value() must not disturb the environments; a copy continues like the original
        twin = env.copy()
        for t in range(8):
            acts = np.array([arng.integers(0, max(1, oracles[e].nP)) for e in range(B)], dtype=np.int32)
            if not any(o.nP > 0 for o in oracles):
                break
            o1, r1, d1, _ = env.step(acts); o2, r2, d2, _ = twin.step(acts)
            for e in range(B):
                if oracles[e].nP == 0:
                    continue
                wr = oracles[e].step(int(acts[e]))
                if r1[e] != wr or r2[e] != wr or not np.array_equal(o1[e], oracles[e].obs(k)) or not np.array_equal(o2[e], o1[e]):
                    print("MISMATCH %s: after value()/copy(), env %d step %d" % (tag, e, t)); sys.exit(1)
    except SystemExit:
        raise
    except Exception as ex:
        msg = str(ex)
        if "error -3" in msg or "error -4" in msg:
            print("skip %s: %s" % (tag, msg[:80])); continue
        print("ERROR %s: %s" % (tag, msg[:300])); sys.exit(1)
    print("ok " + tag)
    del env, twin
print("fuzz_value: %d rounds, %.0f s, no mismatch" % (rounds, time.time() - t0))
