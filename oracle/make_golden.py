"""TEST INFRASTRUCTURE — regenerate tests/golden/ from the COMPILED REFERENCE (oracle/_ref).

Run in the dev container only (needs /root/reference to build oracle/_ref):
    python -m oracle.make_golden
The outputs are data (inputs + expected outputs); no reference source is stored.
"""
import json
import os
import sys

import numpy as np

from . import ffi
from .trace import flat_ideal, run_trace

GOLD = os.path.join(os.path.dirname(ffi.HERE), "tests", "golden")

# name -> (dist, env kwargs, k, nenvs, env seed0, agent seed0, policy, nsteps, until_done)
TRACES = {
    "w3_degree_single": ("3-20-10-weighted", {}, 2, 1, 123, 0, "degree", 0, True),
    "w3_hash_b16": ("3-20-10-weighted", {}, 2, 16, 1000, 0, "hash", 256, False),
    "u3_hash_b4": ("3-20-10-uniform", {}, 1, 4, 2000, 100, "hash", 200, False),
    "u5_hash_b3": ("5-10-5-uniform", {}, 2, 3, 1000, 0, "hash", 400, False),
    "cyc5_hash_b2": ("cyclic-5", {}, 2, 2, 0, 7, "hash", 120, False),
    "cyc6_degree": ("cyclic-6", {}, 2, 1, 0, 0, "degree", 150, False),
    "cyc7_hash_b1": ("cyclic-7", {}, 2, 1, 0, 3, "hash", 60, False),
    "r3_hash_b6": ("3-5-4-0.5-uniform", {}, 2, 6, 500, 0, "hash", 128, False),
    "r4_homog_consts": ("4-6-5-1.5-weighted-consts-homog", {}, 3, 3, 42, 9, "hash", 96, False),
    "w3_lcm": ("3-20-10-weighted", {"elimination": "lcm"}, 2, 3, 300, 1, "hash", 80, False),
    "w3_none_reductions": ("3-20-10-weighted", {"elimination": "none", "rewards": "reductions"}, 2, 3, 310, 2, "hash", 60, False),
    "w3_sortinput_nosortred": ("3-20-10-weighted", {"sort_input": True, "sort_reducers": False}, 2, 3, 320, 3, "hash", 120, False),
    "m3_pure_homog": ("3-8-6-maximum-pure-homog", {}, 2, 3, 5, 5, "hash", 100, False),
    # Poisson mean >= 12 (libstdc++'s rejection branch): polynomials of ~18 terms
    "p3_poisson16": ("3-5-3-16.0-uniform", {}, 2, 2, 77, 5, "hash", 40, False),
    # 8-variable rings (the reference's N, polynomials.h:29)
    "b8_hash_b3": ("8-4-5-uniform", {}, 1, 3, 900, 2, "hash", 150, False),
    "r8_hash_b2": ("8-3-4-1.5-weighted", {}, 2, 2, 31, 4, "hash", 60, False),
    # sort_input with sorted reducers (BuchbergerEnv::reset sorts the drawn generators, buchberger.cpp:299-303)
    "w3_sortinput": ("3-20-10-weighted", {"sort_input": True}, 2, 4, 640, 6, "hash", 160, False),
    "u5_sortinput": ("5-10-5-uniform", {"sort_input": True}, 2, 2, 650, 7, "hash", 200, False),
    # non-binomial random ideals in >= 5 variables whose intermediate polynomials outgrow the device's starting
    # max_poly_terms = 4096 (8630 and 13029 terms: oracle statistic bo_stat_max_terms): the records grow on demand
    "r5_long": ("5-4-4-1.0-uniform", {}, 2, 1, 1177, 177, "hash", 100, False),
    "r8_long": ("8-4-4-0.5-uniform", {}, 2, 1, 1092, 92, "hash", 100, False),
}

GENERATORS = [
    ("3-20-10-weighted", [123, 1000, 1001]),
    ("3-20-10-uniform", [123, 7]),
    ("5-10-5-uniform", [123, 1000]),
    ("3-20-10-maximum", [1]),
    ("3-5-5-uniform", [123]),
    ("3-5-5-0.5-uniform", [123]),
    ("4-6-5-1.5-weighted-consts-homog", [42]),
    ("3-8-6-maximum-pure-homog", [5]),
    ("2-6-4-weighted-consts", [11]),
    ("6-4-7-2.0-maximum", [17]),
    ("cyclic-4", [0]),
    ("cyclic-7", [0]),
    ("3-6-4-15.5-uniform", [123, 9]),
    ("4-5-3-30.0-weighted-homog", [4]),
    ("2-9-3-12.0-maximum-consts", [8]),
    ("8-4-5-uniform", [123]),
    ("8-3-4-1.5-weighted", [6]),
]


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("the reference tree is required to (re)generate goldens")
    ffi.build()
    ref = ffi.load("ref")
    os.makedirs(GOLD, exist_ok=True)
    manifest = {}
    only_meta = "--meta-only" in sys.argv
    for name, (dist, kw, k, nenvs, seed0, aseed0, policy, nsteps, until_done) in TRACES.items():
        manifest[name] = {"dist": dist, "kwargs": kw, "k": k, "nenvs": nenvs, "seed0": seed0,
                          "agent_seed0": aseed0, "policy": policy, "nsteps": nsteps, "until_done": until_done}
        if only_meta:
            continue
        arrays = {}
        for e in range(nenvs):
            env = ref.env(dist, **kw)
            env.seed(seed0 + e)
            tr = run_trace(env, k, nsteps, policy, agent_seed=aseed0 + e, until_done=until_done)
            for key, val in tr.items():
                arrays["e%d_%s" % (e, key)] = val
        np.savez_compressed(os.path.join(GOLD, "trace_%s.npz" % name), **arrays)
        print("trace", name, "steps/env", len(arrays["e0_action"]))
    gens = {}
    for dist, seeds in ([] if only_meta else GENERATORS):
        for s in seeds:
            g = ref.generator(dist)
            g.seed(s)
            for draw in range(3):
                gens["%s|%d|%d" % (dist, s, draw)] = flat_ideal(g.next())
            gens["%s|nvars" % dist] = np.array([g.nvars()], dtype=np.int32)
    if not only_meta:
        np.savez_compressed(os.path.join(GOLD, "generators.npz"), **gens)
    # value() known answers on reset states and after a few steps
    vals = {}
    # NB: First/reversed selections explode on cyclic-n (n >= 5); they are only sampled where cheap.
    cheap = ("first", "degree", "normal", "sugar", "env", "bogus")
    for dist, seed, strats in (("3-20-10-weighted", 123, cheap), ("3-20-10-uniform", 5, cheap),
                               ("3-5-4-0.5-uniform", 9, cheap), ("cyclic-4", 0, cheap),
                               ("cyclic-5", 0, ("degree", "normal", "sugar"))):
        env = ref.env(dist)
        env.seed(seed)
        env.reset()
        for t in range(3):
            for strat in strats:
                vals["%s|%d|%d|%s" % (dist, seed, t, strat)] = env.value(strat, 0.99)
            vals["%s|%d|%d|degree|g0.9" % (dist, seed, t)] = env.value("degree", 0.9)
            if env.nP == 0:
                break
            env.step(ffi.agent_hash(seed, t) % env.nP)
            if env.nP == 0:
                break
    # full Buchberger statistics (cyclic known answers)
    stats = {}
    from .trace import fnv64
    for n, sels in ((3, tuple(ffi.SELECTION)), (4, ("degree", "normal", "sugar", "first", "last", "codegree", "strange", "spice")),
                    (5, ("degree", "normal", "sugar")), (6, ("degree",)), (7, ("degree",))):
        for sel in sels:
            if sel == "random":
                G, st = ref.buchberger(ref.cyclic(n), selection=sel, seed=77)
            else:
                G, st = ref.buchberger(ref.cyclic(n), selection=sel)
            st["basis_size"] = len(G)
            st["basis_hash"] = int(fnv64(flat_ideal(G)))
            stats["cyclic-%d|%s" % (n, sel)] = st
            print("buchberger cyclic-%d %s" % (n, sel), st, flush=True)
    with open(os.path.join(GOLD, "values.json"), "w") as f:
        json.dump({"values": vals, "buchberger": stats, "traces": manifest}, f, indent=1, sort_keys=True)
    print("wrote goldens to", GOLD)


if __name__ == "__main__":
    main()
