"""Randomised parity sweep of the host-stepped (Gym-style) surface (test infrastructure, GPU box): VecLeadMonomialsEnv.reset /
step / masked reset / auto-reset and the single-environment classes with uniformly random actions, every observation matrix,
reward and done flag against the CPU restatement's environment objects.   python scripts/fuzz_gym.py [ROUNDS] [SEED]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import CLeadMonomialsEnv, LeadMonomialsEnv, VecLeadMonomialsEnv, _ffi
from oracle import ffi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bo = ffi.load("bo")
t0 = time.time()


def fail(msg):
    print("MISMATCH " + msg); sys.exit(1)


for it in range(rounds):
    n = rng.choice([2, 3, 3, 3, 4, 5, 6, 8])
    if rng.random() < 0.7:
        dist = "%d-%d-%d-%s" % (n, rng.randint(2, 10 if n <= 3 else 5), rng.randint(2, 8 if n <= 3 else 4), rng.choice(["uniform", "weighted", "maximum"]))
    else:
        dist = "%d-%d-%d-%s-uniform" % (min(n, 5), rng.randint(2, 3), rng.randint(2, 3), rng.choice(["0.3", "0.5"]))
    k = rng.choice([1, 2, 3])
    B = rng.choice([1, 1, 2, 5, 9, 40])                  # (<= 8: the zero-copy path with status-word polling)
    T = rng.choice([20, 60, 150]) if dist.count("-") == 3 and n <= 3 else rng.choice([10, 30])
    elim = rng.choice(["gebauermoeller", "gebauermoeller", "lcm", "none"])
    rewards = rng.choice(["additions", "reductions"])
    auto = rng.random() < 0.5
    caps = rng.choice([None, None, {"lds_max_basis": 16}, {"lds_max_basis": -1}, {"general_class": 1}])
    seed0 = rng.randint(0, 10 ** 6)
    lean = rng.random() < 0.5                            # (lean kernels; with them small batches of the register/LDS class step through host mailbox sessions)
    arng = np.random.default_rng(seed0)
    tag = "%s k=%d B=%d T=%d elim=%s rewards=%s auto=%d caps=%s lean=%d seed0=%d" % (dist, k, B, T, elim, rewards, auto, caps, lean, seed0)
    try:
        env = VecLeadMonomialsEnv(dist, B, elim, rewards, False, True, k, 0, caps, "python")
        env.seed(np.arange(B) + seed0)
        if lean:
            env.accounting(False)
        oracles = []
        for e in range(B):
            o = bo.env(dist, elimination=elim, rewards=rewards); o.seed(seed0 + e); o.reset(); oracles.append(o)
        obs = env.reset()
        how = "reset"
        hist = []
        for t in range(T):
            for e in range(B):
                if not np.array_equal(obs[e], oracles[e].obs(k)):
                    w = oracles[e].obs(k)
                    fail(tag + ": observation of env %d at step %d (last step through %s): device %s %s oracle %s %s rows %s" % (
                        e, t, how, np.asarray(obs[e]).shape, np.asarray(obs[e])[:2].tolist(), w.shape, w[:2].tolist(), env.rows.tolist()))
            if n <= 3 and B <= 5 and elim == "gebauermoeller" and dist.count("-") == 3 and rng.random() < 0.08:
                # value() between the steps of a loop (pg.py:461-465 with --value_model degree): ends a mailbox session, runs clones
                got = env.values("degree", 0.99)
                for e in range(B):
                    if oracles[e].nP > 0 and got[e] != oracles[e].value("degree", 0.99):
                        fail(tag + ": value() of env %d at step %d: device %r oracle %r" % (e, t, got[e], oracles[e].value("degree", 0.99)))
            acts = np.array([arng.integers(0, max(1, oracles[e].nP)) for e in range(B)], dtype=np.int32)
            live = [oracles[e].nP > 0 for e in range(B)]
            if not any(live):
                break
            how = "env.step"
            hist.append(0)
            if rng.random() < 0.3:                             # the C ABI's two-call form: bbx_step(_autoreset), then bbx_obs (padded block)
                r = np.zeros(B); dn = np.zeros(B, dtype=np.uint8)
                fn = _ffi.lib().bbx_step_autoreset if auto else _ffi.lib().bbx_step
                _ffi.check(fn(env._h, _ffi.ptr(acts), _ffi.ptr(r), _ffi.ptr(dn), _ffi.ptr(env.rows)))
                d = dn.astype(bool)
                obs = env.observations()
                how = "bbx_step + bbx_obs"
                hist[-1] = 1
            elif auto:
                obs, r, d, _ = env.step(acts, auto_reset=True)
            else:
                obs, r, d, _ = env.step(acts)
            mask = np.zeros(B, dtype=np.uint8)
            for e in range(B):
                if not live[e]:
                    continue
                want_r = oracles[e].step(int(acts[e]))
                done = oracles[e].nP == 0
                if r[e] != want_r or bool(d[e]) != done:
                    fail(tag + ": reward/done of env %d at step %d (through %s; the steps before: %s): device (%s, %s) oracle (%s, %s); session stats %s" % (e, t, how, hist[-8:], r[e], d[e], want_r, done, env.session_stats()))
                if done:
                    if auto:
                        oracles[e].reset()
                    elif arng.random() < 0.7:
                        mask[e] = 1
            if not auto and mask.any():
                obs = env.reset(mask)
                for e in np.flatnonzero(mask):
                    oracles[e].reset()
    except SystemExit:
        raise
    except Exception as ex:
        msg = str(ex)
        if "error -3" in msg:
            print("CAPACITY %s: %s" % (tag, msg[:160])); sys.exit(1)   # (records grow on demand: a failure)
        if "error -4" in msg or "error -5" in msg or "bad distribution" in msg:
            print("generator %s: %s" % (dist, msg[:90])); continue
        print("ERROR %s: %s" % (tag, msg[:300])); sys.exit(1)
    print("ok " + tag)
    del env
print("fuzz_gym: %d rounds, %.0f s, no mismatch" % (rounds, time.time() - t0))
