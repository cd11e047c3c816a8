"""TEST INFRASTRUCTURE — golden vectors from the reference's PYTHON twin.

Imports /root/reference/deepgroebner/{buchberger,ideals}.py where they lie (nothing is copied; an empty `IPython`
module stands in for an unused import, and `np.product` — removed in NumPy 2 — is aliased to `np.prod`), runs the
episodes of the reference's own tests/test_buchberger.py:246-362 plus a few seeded random ideals through

    BuchbergerEnv.reset() -> (G, P), BuchbergerEnv.step((i, j)) -> ((G, P), reward, done, {})   buchberger.py:330-375
    BuchbergerAgent / select                                                                     buchberger.py:397-439
    LeadMonomialsEnv / LeadMonomialsAgent                                                         buchberger.py:448-567

and records inputs and outputs as data in tests/golden/py_reference.json.gz: the ideal (terms), per step the chosen pair,
the reward, the pair list, and the monic new basis element; for the LeadMonomialsEnv runs the state matrix of every
step.  The random ideals are recorded too (the Python generators use NumPy's PCG64, the C++ ones minstd_rand0: the
seeded streams differ, so the ideal itself is the fixture).

    python oracle/make_py_golden.py        (only in the build container: needs /root/reference)
"""
import gzip
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "py_reference.json.gz")


def load_reference():
    sys.modules.setdefault("IPython", types.ModuleType("IPython"))
    if not hasattr(np, "product"):
        np.product = np.prod
    sys.path.insert(0, REF)
    import deepgroebner.buchberger as rb
    import deepgroebner.ideals as ri
    return rb, ri


def terms(f):
    """sympy PolyElement -> [[coef, [exps...]], ...] in the ring's term order (descending)."""
    return [[int(c) % 32003, [int(e) for e in m]] for m, c in f.terms()]


def run_buchberger_env(rb, env, agent, max_steps=100000):
    G, P = env.reset()
    rec = {"ideal": [terms(g) for g in G], "init_pairs": [list(p) for p in P], "steps": []}
    done, total = False, 0.0
    state = (G, P)
    while not done and len(rec["steps"]) < max_steps:
        action = agent.act(state)
        nG = len(state[0])
        state, reward, done, _ = env.step(action)
        total += reward
        G, P = state
        rec["steps"].append({"action": list(action), "reward": reward, "pairs": [list(p) for p in P],
                             "new": terms(G[-1]) if len(G) > nG else None})
    rec["total_reward"] = total
    return rec


def run_lead_env(rb, env, agent, nsteps):
    state = env.reset()
    rec = {"ideal_monic": [terms(g) for g in env.env.G], "init_state": state.tolist(), "steps": []}
    done = False
    while not done and len(rec["steps"]) < nsteps:
        action = int(agent.act(state))
        state, reward, done, _ = env.step(action)
        rec["steps"].append({"action": action, "reward": reward, "state": state.tolist(), "done": bool(done)})
    return rec


def main():
    import sympy as sp
    rb, ri = load_reference()
    out = {"source": "deepgroebner/buchberger.py + ideals.py of the reference, imported in the build container",
           "sympy": sp.__version__, "numpy": np.__version__, "buchberger_env": [], "lead_env": [], "select": [], "lmv": []}

    # ---- sort_reducers on / off on a fixed ideal (the shape of tests/test_buchberger.py:246-257, whose own rings are
    # lex-ordered over QQ and therefore outside the device path: grevlex over GF(32003) here), full episodes
    R, a, b, c, d = sp.ring("a,b,c,d", sp.FF(32003), "grevlex")
    F0 = [a**2*b*d - c**2, a*d - b*c**2 - d, a - c]
    for sr in (True, False):
        env = rb.BuchbergerEnv(ri.FixedIdealGenerator(F0), sort_reducers=sr, elimination="lcm")
        # (sort_reducers=False: the reference's step() stores the polynomial list as the lead-monomial list after the
        # first non-zero reduction, buchberger.py:372-373, and fails on the reduction after it — one step is what its own
        # test takes, test_buchberger.py:253-257)
        rec = run_buchberger_env(rb, env, rb.BuchbergerAgent(selection="first"), max_steps=100000 if sr else 1)
        rec.update({"name": "env0_sort_reducers_%d" % sr, "nvars": 4, "kwargs": {"sort_reducers": sr, "elimination": "lcm"}, "selection": "first"})
        out["buchberger_env"].append(rec)

    # ---- episodes of tests/test_buchberger.py:270-312 (grevlex ones; the grlex case is outside the device path)
    R, a, b, c, d, e = sp.ring("a,b,c,d,e", sp.FF(32003), "grevlex")
    F1 = [a + 2*b + 2*c + 2*d + 2*e - 1, a**2 + 2*b**2 + 2*c**2 + 2*d**2 + 2*e**2 - a, 2*a*b + 2*b*c + 2*c*d + 2*d*e - b,
          b**2 + 2*a*c + 2*b*d + 2*c*e - c, 2*b*c + 2*a*d + 2*b*e - d]
    for s in ("first", ["degree", "first"], ["normal", "first"]):
        env = rb.BuchbergerEnv(ri.FixedIdealGenerator(F1), rewards="reductions")
        rec = run_buchberger_env(rb, env, rb.BuchbergerAgent(selection=s))
        rec.update({"name": "episode0_%s" % (s if isinstance(s, str) else "+".join(s)), "nvars": 5,
                    "kwargs": {"rewards": "reductions"}, "selection": s})
        assert rec["total_reward"] == -28, rec["total_reward"]
        out["buchberger_env"].append(rec)
    R, a, b, c, d = sp.ring("a,b,c,d", sp.FF(32003), "grevlex")
    F2 = [a + b + c + d, a*b + b*c + c*d + d*a, a*b*c + b*c*d + c*d*a + d*a*b, a*b*c*d - 1]
    for el, want in (("none", -45), ("lcm", -35), ("gebauermoeller", -11)):
        env = rb.BuchbergerEnv(ri.FixedIdealGenerator(F2), elimination=el, rewards="reductions")
        rec = run_buchberger_env(rb, env, rb.BuchbergerAgent(selection=["normal", "first"]))
        rec.update({"name": "episode1_%s" % el, "nvars": 4, "kwargs": {"elimination": el, "rewards": "reductions"},
                    "selection": ["normal", "first"]})
        assert rec["total_reward"] == want, rec["total_reward"]
        out["buchberger_env"].append(rec)

    # ---- seeded random ideals of the Python generators, additions rewards, full episodes
    for dist, seed, sel in (("3-20-10-weighted", 123, ["degree", "first"]), ("3-20-10-uniform", 7, ["normal", "first"]),
                            ("4-5-4-uniform", 5, "first")):
        env = rb.BuchbergerEnv(dist)
        env.seed(seed)
        rec = run_buchberger_env(rb, env, rb.BuchbergerAgent(selection=sel), max_steps=160)
        rec.update({"name": "random_%s_seed%d" % (dist, seed), "nvars": int(dist.split("-")[0]), "kwargs": {}, "selection": sel})
        out["buchberger_env"].append(rec)

    # ---- LeadMonomialsEnv + LeadMonomialsAgent (buchberger.py:448-567), k = 1 and 2
    for dist, seed, k, sel in (("3-20-10-weighted", 123, 2, "degree"), ("3-20-10-uniform", 123, 1, "first"), ("5-10-5-uniform", 9, 2, "degree")):
        env = rb.LeadMonomialsEnv(dist, k=k)
        env.seed(seed)
        rec = run_lead_env(rb, env, rb.LeadMonomialsAgent(selection=sel, k=k), 100)
        rec.update({"name": "lead_%s_seed%d_k%d_%s" % (dist, seed, k, sel), "nvars": int(dist.split("-")[0]), "k": k, "selection": sel})
        out["lead_env"].append(rec)
    R3, x, y, z = sp.ring("x,y,z", sp.FF(32003), "grevlex")
    for el in ("none", "gebauermoeller"):
        env = rb.LeadMonomialsEnv(ri.FixedIdealGenerator([y - x**2, z - x**3]), elimination=el)
        rec = run_lead_env(rb, env, rb.LeadMonomialsAgent(selection="first"), 20)
        rec.update({"name": "lead_twisted_cubic_%s" % el, "nvars": 3, "k": 1, "selection": "first", "kwargs": {"elimination": el}})
        out["lead_env"].append(rec)

    # ---- select() on recorded states, all strategies incl. tie-break lists (buchberger.py:415-439)
    env = rb.BuchbergerEnv("3-20-10-weighted"); env.seed(11)
    G, P = env.reset()
    for t in range(12):
        row = {"basis": [terms(g) for g in G], "pairs": [list(p) for p in P], "picks": {}}
        for s in ("first", "normal", "degree", ["degree", "first"], ["degree", "normal"], ["normal", "first"]):
            row["picks"][s if isinstance(s, str) else "+".join(s)] = list(rb.select(G, P, strategy=s))
        out["select"].append(row)
        (G, P), _, done, _ = env.step(rb.select(G, P, strategy=["degree", "normal"]))
        if done:
            break

    # ---- lead_monomials_vector known answers (tests/test_buchberger.py:315-330)
    R1, x, y, z = sp.ring("x,y,z", sp.FF(32003), "grevlex")
    R2, a, b, c, d = sp.ring("a,b,c,d", sp.FF(32003), "grevlex")
    for f, ring, k in ((R1.one, R1, 1), (R2.zero, R2, 2), (x*y, R1, 1), (x*y, R1, 3), (x*y**2*z + x**3 + z + 1, R1, 1),
                       (x*y**2*z + x**3 + z + 1, R1, 2), (x*y**2*z + x**3 + z + 1, R1, 4), (b*d**5 + a**3, R2, 1), (b*d**5 + a**3, R2, 3)):
        out["lmv"].append({"poly": terms(f), "nvars": ring.ngens, "k": k, "vector": rb.lead_monomials_vector(f, ring, k=k).tolist()})

    with gzip.GzipFile(OUT, "wb", mtime=0) as fh:
        fh.write(json.dumps(out, separators=(",", ":")).encode())
    print("wrote %s (%d bytes): %d BuchbergerEnv runs, %d LeadMonomialsEnv runs" % (OUT, os.path.getsize(OUT), len(out["buchberger_env"]), len(out["lead_env"])))


if __name__ == "__main__":
    main()
