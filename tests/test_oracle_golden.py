"""The C restatement (oracle/bbx_oracle.c) must reproduce every golden vector that
oracle/make_golden.py recorded from the compiled reference."""
import numpy as np
import pytest

from oracle import ffi
from oracle.trace import flat_ideal, fnv64, run_trace
from tests.golden_util import assert_trace_equal, load_trace, meta, trace_names, GOLD


@pytest.mark.parametrize("name", trace_names())
def test_trace(bo, name):
    m = meta()["traces"][name]
    gold = load_trace(name)
    for e in range(m["nenvs"]):
        env = bo.env(m["dist"], **m["kwargs"])
        env.seed(m["seed0"] + e)
        got = run_trace(env, m["k"], m["nsteps"], m["policy"], agent_seed=m["agent_seed0"] + e, until_done=m["until_done"])
        assert_trace_equal(got, gold, e, name)


def test_generators(bo):
    import os
    gold = np.load(os.path.join(GOLD, "generators.npz"))
    keys = [k for k in gold.files if not k.endswith("|nvars")]
    seen = {}
    for key in sorted(keys, key=lambda s: (s.split("|")[0], int(s.split("|")[1]), int(s.split("|")[2]))):
        dist, seed, draw = key.split("|")
        gk = (dist, seed)
        if gk not in seen:
            g = bo.generator(dist)
            g.seed(int(seed))
            seen[gk] = g
            assert g.nvars() == int(gold["%s|nvars" % dist][0])
        got = flat_ideal(seen[gk].next())
        assert np.array_equal(got, gold[key]), key


def test_values(bo):
    vals = meta()["values"]
    envs = {}
    # replay the same walk make_golden.py did
    for key in sorted(vals):
        parts = key.split("|")
        dist, seed, t, strat = parts[0], int(parts[1]), int(parts[2]), parts[3]
        gamma = 0.9 if len(parts) == 5 else 0.99
        ek = (dist, seed, t)
        if ek not in envs:
            env = bo.env(dist)
            env.seed(seed)
            env.reset()
            for s in range(t):
                env.step(ffi.agent_hash(seed, s) % env.nP)
            envs[ek] = env
        assert envs[ek].value(strat, gamma) == vals[key], key


@pytest.mark.parametrize("key", sorted(meta()["buchberger"].keys()))
def test_buchberger_stats(bo, key):
    want = meta()["buchberger"][key]
    dist, sel = key.split("|")
    n = int(dist.split("-")[1])
    G, st = bo.buchberger(bo.cyclic(n), selection=sel, seed=77 if sel == "random" else None)
    for f in ("zero_reductions", "nonzero_reductions", "polynomial_additions", "total_reward", "discounted_return"):
        assert st[f] == want[f], (key, f)
    assert len(G) == want["basis_size"]
    assert int(fnv64(flat_ideal(G))) == want["basis_hash"]
