"""The Python-twin surface (deepgroebner/buchberger.py:243-567 of the reference) against fixtures
recorded from the reference's own Python code (oracle/make_py_golden.py -> tests/golden/py_reference.json.gz):
select / BuchbergerAgent / LeadMonomialsAgent / lead_monomials_vector on the host (CPU tests), and the
BuchbergerEnv.reset() -> (G, P) / step((i, j)) and LeadMonomialsEnv episodes replayed on the HIP path (GPU tests)."""
import gzip
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "py_reference.json.gz")


@pytest.fixture(scope="module")
def gold():
    with gzip.open(GOLD) as fh:
        return json.loads(fh.read().decode())


def _poly(ts):
    return [(int(c), tuple(int(x) for x in e)) for c, e in ts]


def _sel(s):
    return s if isinstance(s, str) else list(s)


# ---- host logic (no GPU) ------------------------------------------------------------------------------------------------
def test_python_generators_numpy_streams_known_answers():
    """The reference's Python generators draw from numpy's default_rng (ideals.py:214, 250, 302), not from the C++ generators'
    engine; stream="numpy" reproduces those streams.  Expected ideals: the reference's own tests/test_ideals.py:48-69
    (seed 123), as data — term lists (coefficient, exponents), lead term first; and its degree distributions :33-45."""
    from deepgroebner_amd import ideals
    g = ideals.RandomBinomialIdealGenerator(3, 5, 5, stream="numpy")
    g.seed(123)
    assert next(g) == [[(1, (2, 0, 1)), (495, (0, 0, 1))],
                       [(1, (0, 1, 4)), (5901, (1, 1, 1))],
                       [(1, (5, 0, 0)), (14384, (3, 0, 2))],
                       [(1, (3, 1, 1)), (16417, (0, 2, 1))],
                       [(1, (3, 1, 1)), (13109, (0, 3, 2))]]
    g = ideals.RandomIdealGenerator(3, 5, 5, 0.5, stream="numpy")
    g.seed(123)
    assert next(g) == [[(1, (3, 0, 2)), (10689, (2, 1, 0)), (12547, (0, 1, 2))],
                       [(1, (0, 1, 4)), (15388, (0, 2, 1)), (22355, (0, 1, 2))],
                       [(1, (1, 1, 2)), (4665, (3, 0, 0)), (15800, (2, 1, 0))],
                       [(1, (3, 0, 2)), (8782, (0, 2, 3)), (15890, (0, 1, 2))],
                       [(1, (0, 2, 1)), (30687, (0, 2, 0))]]
    assert np.array_equal(ideals.RandomBinomialIdealGenerator(3, 3, 1, "uniform", stream="numpy").degree_dist, np.array([0, 3, 6, 10]) / 19.0)
    assert np.array_equal(ideals.RandomBinomialIdealGenerator(3, 3, 1, "weighted", stream="numpy").degree_dist, np.array([0, 1, 1, 1]) / 3.0)
    # the same seed gives the same ideals again; homogeneous / pure variants keep their promises
    a = ideals.RandomBinomialIdealGenerator(4, 6, 7, "weighted", homogeneous=True, pure=True, stream="numpy"); a.seed(5)
    b = ideals.RandomBinomialIdealGenerator(4, 6, 7, "weighted", homogeneous=True, pure=True, stream="numpy"); b.seed(5)
    F = next(a)
    assert F == next(b) and len(F) == 7
    for f in F:
        assert f[1][0] == 32002 and sum(f[0][1]) == sum(f[1][1]) and ideals._grevlex_key(f[0][1]) > ideals._grevlex_key(f[1][1])
    with pytest.raises(ValueError):
        ideals.RandomIdealGenerator(3, 5, 5, 0.5, stream="pcg")


def test_select_matches_reference_on_recorded_states(gold):
    from deepgroebner_amd.buchberger import BuchbergerAgent, select
    assert len(gold["select"]) >= 5
    for row in gold["select"]:
        G = [_poly(f) for f in row["basis"]]
        P = [tuple(p) for p in row["pairs"]]
        for name, want in row["picks"].items():
            strategy = name.split("+") if "+" in name else name
            assert list(select(G, P, strategy=strategy)) == want, name
            assert list(BuchbergerAgent(selection=strategy).act((G, P))) == want
    with pytest.raises(ValueError):
        select(G, P, strategy="bogus")


def test_lead_monomials_vector_known_answers(gold):
    from deepgroebner_amd.buchberger import lead_monomials_vector
    for row in gold["lmv"]:
        got = lead_monomials_vector(_poly(row["poly"]), row["nvars"], k=row["k"])
        assert got.dtype == np.int32 and got.tolist() == row["vector"], row


def test_lead_monomials_agent_on_recorded_matrices(gold):
    from deepgroebner_amd.buchberger import LeadMonomialsAgent
    for run in gold["lead_env"]:
        agent = LeadMonomialsAgent(selection=run["selection"], k=run["k"])
        state = np.array(run["init_state"], dtype=np.int32)
        for st in run["steps"]:
            assert int(agent.act(state)) == st["action"], run["name"]
            state = np.array(st["state"], dtype=np.int32).reshape(-1, 2 * run["nvars"] * run["k"])


# ---- the HIP path ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_buchberger_env_replays_python_reference_episodes(gold):
    """reset() -> (G, P) and step((i, j)) -> ((G, P), reward, done, {}): pair list, monic new element, reward and the
    episode return of every recorded run (tests/test_buchberger.py:270-299 answers -28 and -45 / -35 / -11 among them);
    our BuchbergerAgent picks the recorded pair from the device-backed state at every step."""
    from deepgroebner_amd import BuchbergerAgent, BuchbergerEnv, FixedIdealGenerator
    want_total = {"episode0_first": -28, "episode0_degree+first": -28, "episode0_normal+first": -28,
                  "episode1_none": -45, "episode1_lcm": -35, "episode1_gebauermoeller": -11}
    for run in gold["buchberger_env"]:
        F = [_poly(f) for f in run["ideal"]]
        env = BuchbergerEnv(FixedIdealGenerator(F), **run["kwargs"])
        agent = BuchbergerAgent(selection=_sel(run["selection"]))
        G, P = env.reset()
        assert G == F, run["name"]                                   # (the recorded generators are monic already)
        assert [list(p) for p in P] == run["init_pairs"], run["name"]
        total, done = 0.0, False
        for st in run["steps"]:
            assert not done
            action = tuple(st["action"])
            assert tuple(agent.act((G, P))) == action, run["name"]
            nG = len(G)
            (G, P), reward, done, info = env.step(action)
            total += reward
            assert reward == st["reward"] and info == {}, run["name"]
            assert [list(p) for p in P] == st["pairs"], run["name"]
            if st["new"] is None:
                assert len(G) == nG
            else:
                assert len(G) == nG + 1 and G[-1] == _poly(st["new"]), run["name"]
        if run["name"] in want_total:
            assert done and total == want_total[run["name"]] == run["total_reward"]
        with pytest.raises(ValueError):
            env.step((99, 100))                                      # like list.remove on a pair that is not in P


@pytest.mark.gpu
def test_lead_monomials_env_replays_python_reference_runs(gold):
    """LeadMonomialsEnv (python flavour: observation width = ring variables) over the recorded ideals: the state
    matrix, reward and done flag of every step of the reference's runs, k = 1 and 2."""
    from deepgroebner_amd import FixedIdealGenerator, LeadMonomialsAgent, LeadMonomialsEnv
    for run in gold["lead_env"]:
        F = [_poly(f) for f in run["ideal_monic"]]
        env = LeadMonomialsEnv(FixedIdealGenerator(F), k=run["k"], **run.get("kwargs", {}))
        agent = LeadMonomialsAgent(selection=run["selection"], k=run["k"])
        state = env.reset()
        cols = 2 * run["nvars"] * run["k"]
        assert state.dtype == np.int32 and np.array_equal(state, np.array(run["init_state"], dtype=np.int32).reshape(-1, cols))
        for st in run["steps"]:
            assert int(agent.act(state)) == st["action"]
            state, reward, done, info = env.step(st["action"])
            assert reward == st["reward"] and done == st["done"] and info == {}, run["name"]
            assert np.array_equal(state, np.array(st["state"], dtype=np.int32).reshape(-1, cols)), run["name"]
