"""Parity of the HIP path (through the C ABI) against the golden vectors recorded from the
compiled reference and against the CPU oracle.  Bit-exact: every action, reward, |P|, |G|,
observation matrix, pair list and new basis element of every step, plus the full final state."""
import os

import numpy as np
import pytest

from oracle import ffi
from oracle.trace import degree_action, fnv64, poly_words, run_trace
from tests.golden_util import load_trace, meta, trace_names

pytestmark = pytest.mark.gpu

POLICY = {"hash": "random", "degree": "degree", "first": "first"}


def make_env(m, batch, **extra):
    from deepgroebner_amd import VecLeadMonomialsEnv
    return VecLeadMonomialsEnv(m["dist"], batch=batch, k=m["k"], **m["kwargs"], **extra)


def compare_with_trace(env, e, want, T, label):
    got = env.trace_read(e, 0, T)
    for field, key in (("action", "action"), ("reward", "reward"), ("rows", "nP"), ("basis_size", "nG"), ("done", "done"),
                       ("obs_hash", "obs_hash"), ("pairs_hash", "pairs_hash"), ("newpoly_hash", "newpoly_hash")):
        w = np.asarray(want[key])
        g = got[field].astype(w.dtype)
        if not np.array_equal(g, w):
            bad = int(np.flatnonzero(g != w)[0])
            raise AssertionError("%s env %d: %s differs first at step %d: %r != %r" % (label, e, field, bad, g[bad], w[bad]))
    basis, pairs, order = env.state(e)
    assert np.array_equal(pairs, want["final_pairs"]), label
    assert np.array_equal(order, want["final_order"]), label
    words = [poly_words(c, x) for c, x in basis]
    assert np.array_equal(np.concatenate(words) if words else np.zeros(0, np.int32), want["final_basis"]), label


# residency classes of the step kernel: default (LDS-resident where the class allows), HBM-resident only,
# and an LDS class so small that environments keep spilling into the HBM-resident follow-up pass
RESIDENCY = {"default": None, "hbm": {"lds_max_basis": -1}, "spill": {"lds_max_basis": 16},
             "general": {"general_class": 1}, "general_hbm": {"general_class": 1, "lds_max_basis": -1},
             "general_spill": {"general_class": 1, "lds_max_basis": 16},
             # wide class (fixed ideals): an LDS window so small that reducer tails stream through it in chunks and the
             # polynomial being reduced keeps moving to the HBM-resident buffers; three waves instead of eight
             "wide_small": {"wide_lds_terms": 64}, "wide_small_3waves": {"wide_lds_terms": 40, "wide_waves": 3},
             "wide_off": {"wide_waves": -1},
             # ..._lean: accounting off = the variant whose reducer tails collect in an LDS accumulator
             "wide_lean": None, "wide_small_lean": {"wide_lds_terms": 64}, "wide_small_3waves_lean": {"wide_lds_terms": 40, "wide_waves": 3}}


@pytest.mark.parametrize("residency", sorted(RESIDENCY))
@pytest.mark.parametrize("name", trace_names())
def test_golden_trace(name, residency):
    m = meta()["traces"][name]
    binomial = "." not in m["dist"] and not m["dist"].startswith("cyclic")
    three_var = m["dist"].startswith("3-") or m["dist"].startswith("2-")
    cyc, wide_res = m["dist"].startswith("cyclic"), residency.startswith("wide")
    if wide_res != cyc and (wide_res or residency != "default"):
        pytest.skip("the wide class serves fixed ideals only")
    if not cyc and residency != "default" and not binomial:
        pytest.skip("kernel classes only differ for binomial distributions")
    if residency in ("spill", "general_spill", "hbm") and not three_var:
        pytest.skip("only <=3-variable binomial distributions have an LDS-resident class")
    gold = load_trace(name)
    B = m["nenvs"]
    T = len(gold["e0_action"])
    env = make_env(m, B, caps=RESIDENCY[residency])
    if residency.endswith("_lean"):
        env.accounting(False)
    env.seed(np.arange(B) + m["seed0"])
    env.seed_agent(np.arange(B) + m["agent_seed0"])
    env.trace_enable(max(T, 1))
    obs0 = env.reset()
    for e in range(B):
        assert (len(env.state(e)[0]), int(env.rows[e])) == tuple(gold["e%d_init" % e]), name
        assert fnv64(obs0[e]) == int(gold["e%d_init_hash" % e][0]), name
    env.rollout(POLICY[m["policy"]], T, auto_reset=not m["until_done"])
    final_obs = env.observations()
    for e in range(B):
        want = {k[len("e%d_" % e):]: gold[k] for k in gold.files if k.startswith("e%d_" % e)}
        assert len(want["action"]) == T
        compare_with_trace(env, e, want, T, name)
        assert np.array_equal(final_obs[e], want["final_obs"]), name
    if name.endswith("_long"):                        # these runs outgrow the starting scratch: the records were enlarged
        assert env.capacities()["max_poly_terms"] > 4096 and env.capacities()["grown"] > 0, env.capacities()


def test_gym_surface_single_env_vs_oracle():
    """CLeadMonomialsEnv.reset/step with host-chosen actions, observation compared matrix by matrix."""
    from deepgroebner_amd import CLeadMonomialsEnv
    bo = ffi.load("bo")
    env = CLeadMonomialsEnv("3-20-10-weighted", k=2)
    env.seed(123)
    o = bo.env("3-20-10-weighted")
    o.seed(123)
    for episode in range(3):
        state = env.reset()
        o.reset()
        assert state.dtype == np.int32 and np.array_equal(state, o.obs(2))
        if episode == 0:
            assert state.shape == (19, 12) and state[0].tolist() == [11, 6, 3, 9, 7, 2, 7, 0, 5, 2, 0, 2]
        t = 0
        done = False
        while not done:
            a = ffi.agent_hash(7, t) % len(state)
            state, r, done, info = env.step(np.int64(a))
            assert r == o.step(a) and info == {}
            assert np.array_equal(state, o.obs(2))
            assert done == (o.nP == 0) and (len(state) == 0) == done
            t += 1


def test_short_lived_handles_reuse_pinned_memory_safely():
    """Many small handles created, stepped and destroyed one after the other (what a hyper-parameter sweep or a test-suite
    does): the host-stepped path of batches <= 8 watches status words in pinned memory, and pinned memory comes back from
    the allocator with the words of the handle that owned it before — every step of every handle against the oracle."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    for rnd in range(40):
        B, k = 1 + rnd % 3, 1 + rnd % 2
        dist = ("3-6-3-uniform", "3-7-4-uniform", "3-20-10-weighted")[rnd % 3]
        env = VecLeadMonomialsEnv(dist, batch=B, k=k)
        env.seed(np.arange(B) + 500 + rnd)
        oracles = []
        for e in range(B):
            o = bo.env(dist); o.seed(500 + rnd + e); o.reset(); oracles.append(o)
        obs = env.reset()
        for t in range(25):
            for e in range(B):
                assert np.array_equal(obs[e], oracles[e].obs(k)), (rnd, t, e)
            acts = np.array([ffi.agent_hash(e + rnd, t) % max(1, oracles[e].nP) for e in range(B)], dtype=np.int32)
            obs, r, d, _ = env.step(acts, auto_reset=True)
            for e in range(B):
                assert r[e] == oracles[e].step(int(acts[e])) and bool(d[e]) == (oracles[e].nP == 0), (rnd, t, e)
                if oracles[e].nP == 0:
                    oracles[e].reset()
        del env


def test_vec_step_matches_oracle_and_masked_reset():
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, k = 5, 1
    env = VecLeadMonomialsEnv("3-20-10-uniform", batch=B, k=k)
    env.seed(np.arange(B) + 77)
    oracles = []
    for e in range(B):
        o = bo.env("3-20-10-uniform"); o.seed(77 + e); o.reset(); oracles.append(o)
    obs = env.reset()
    for t in range(150):
        for e in range(B):
            assert np.array_equal(obs[e], oracles[e].obs(k)), (t, e)
        acts = np.array([ffi.agent_hash(e, t) % max(1, oracles[e].nP) for e in range(B)], dtype=np.int32)
        obs, r, d, _ = env.step(acts)
        mask = np.zeros(B, dtype=np.uint8)
        for e in range(B):
            if oracles[e].nP == 0:       # finished earlier and not reset: untouched
                continue
            assert r[e] == oracles[e].step(int(acts[e])), (t, e)
            assert bool(d[e]) == (oracles[e].nP == 0)
            if oracles[e].nP == 0 and (t + e) % 2 == 0:
                mask[e] = 1
        if mask.any():
            obs = env.reset(mask)
            for e in np.flatnonzero(mask):
                oracles[e].reset()


def test_fixed_ideal_python_flavour_and_eliminations():
    """tests/test_buchberger.py:328-362 of the reference (LeadMonomialsEnv over a FixedIdealGenerator)."""
    from deepgroebner_amd import FixedIdealGenerator, LeadMonomialsEnv
    F = [[(1, (0, 1, 0)), (-1, (2, 0, 0))], [(1, (0, 0, 1)), (-1, (3, 0, 0))]]     # y - x^2, z - x^3
    env = LeadMonomialsEnv(FixedIdealGenerator(F))
    state = env.reset()
    assert np.array_equal(state, [[2, 0, 0, 3, 0, 0]])
    state, _, done, _ = env.step(0)
    assert np.array_equal(state, [[2, 0, 0, 1, 1, 0]]) and not done
    state, _, done, _ = env.step(0)
    assert np.array_equal(state, [[1, 1, 0, 0, 2, 0]]) and not done
    state, _, done, _ = env.step(0)
    assert done
    env = LeadMonomialsEnv(FixedIdealGenerator(F), elimination="none")
    state = env.reset()
    state, _, done, _ = env.step(0)
    assert np.array_equal(state, [[2, 0, 0, 1, 1, 0], [3, 0, 0, 1, 1, 0]])


def test_copy_is_deep():
    from deepgroebner_amd import CLeadMonomialsEnv
    env = CLeadMonomialsEnv("3-20-10-weighted", k=2)
    env.seed(5)
    s0 = env.reset()
    cp = env.copy()
    s1, r1, _, _ = env.step(0)
    s1c, r1c, _, _ = cp.step(0)
    assert np.array_equal(s1, s1c) and r1 == r1c
    env.step(0)
    assert np.array_equal(cp.reset(), env.reset())    # generator state travelled with the copy


def test_capacity_overflow_is_reported_not_wrapped():
    """caps['no_growth']: the configured capacities are hard limits and exceeding one is BBX_E_CAPACITY."""
    from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=4, k=1, caps={"max_basis": 12, "max_pairs": 16, "no_growth": 1})
    env.seed(np.arange(4))
    env.reset()
    with pytest.raises(_ffi.BbxError) as ei:
        env.rollout("random", 200, auto_reset=True)
    assert ei.value.code == -3


def _assert_equals_oracle_run(env, want, sample_every=1):
    st = env.stats()
    assert (st[:, 4] == 0).all(), st[:, 4]
    for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
        w = np.array([r[key] for r in want])
        assert np.array_equal(st[:, col], w), (key, int(np.flatnonzero(st[:, col] != w)[0]))
    B = len(want)
    for e in sorted(set(range(0, B, sample_every)) | {B - 1}):
        basis, pairs, order = env.state(e)
        assert fnv64(_state_words(basis, pairs, order)) == want[e]["state_hash"], e


@pytest.mark.parametrize("dist,B,T,caps,cls", [
    # binomial classes: the basis / pair arrays of the HBM record start far too small
    ("3-20-10-weighted", 9, 200, {"max_basis": 12, "max_pairs": 16}, "fast + binomial continuation"),
    ("3-20-10-weighted", 9, 200, {"max_basis": 12, "max_pairs": 16, "lds_max_basis": -1}, "binomial, HBM-resident"),
    ("5-10-5-uniform", 5, 300, {"max_basis": 16, "max_pairs": 32}, "binomial, 16-byte monomials"),
    # general class: every array starts tiny (scratch of 8 terms, arena of 64): dozens of doublings
    ("3-20-10-weighted", 9, 120, {"general_class": 1, "lds_max_basis": -1, "max_basis": 8, "max_pairs": 8, "arena_terms": 32, "max_poly_terms": 4}, "general on binomials"),
    ("3-5-4-0.5-uniform", 9, 100, {"max_basis": 8, "max_pairs": 8, "arena_terms": 64, "max_poly_terms": 8}, "general"),
    ("5-4-4-1.0-uniform", 17, 60, {"max_basis": 8, "max_pairs": 16, "arena_terms": 64, "max_poly_terms": 16}, "general, long polynomials"),
    ("8-3-4-1.5-weighted", 5, 40, {"max_basis": 8, "max_pairs": 16, "arena_terms": 64, "max_poly_terms": 16}, "general, 32-byte monomials"),
])
def test_capacity_is_a_cliff_not_a_failure(dist, B, T, caps, cls):
    """Every per-environment array of the record starts far too small: the kernels stop BEFORE the step that does not fit,
    the host doubles what was full (bbx_api.cpp grow_records) and the step is taken then — counters of every environment
    and every final state equal the oracle's, as if the capacities had been ample (reference: heap vectors,
    polynomials.h:71-94, buchberger.cpp:24-99)."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    want = bo.run_random_many(dist, 2, range(500, 500 + B), range(B), T, True, 0)
    env = VecLeadMonomialsEnv(dist, batch=B, k=2, caps=caps)
    env.seed(np.arange(B) + 500); env.seed_agent(np.arange(B)); env.reset()
    env.rollout("random", T, auto_reset=True)
    _assert_equals_oracle_run(env, want)
    assert env.capacities()["grown"] > 0, cls


@pytest.mark.parametrize("persistent", [0, 1])
@pytest.mark.parametrize("dist,B,T,reach", [("3-20-40-weighted", 96, 120, 64), ("3-20-140-weighted", 32, 160, 128),
                                            ("3-20-200-weighted", 24, 120, 192), ("3-20-260-weighted", 16, 100, 256)])
def test_fast_class_second_reducer_bank_lean(persistent, dist, B, T, reach):
    """3-20-40-weighted starts every episode with 40 generators, so bases pass 64 elements within a few steps: the lean fast
    kernels (the reduction loop written in assembly, bbx_fast.h) find divisors in the second bank of reducer registers
    (reducers 64..127); with 140 / 200 generators the third and fourth banks (reducers 128..255) are in use; with 260 every
    environment outgrows the class's 256 elements, continues in the HBM-resident class and comes back — counters of every
    environment and sampled final states against the oracle, with one kernel per launch and through a persistent session."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    want = bo.run_random_many(dist, 2, range(700, 700 + B), range(B), T, True, 0)
    assert max(r["nG"] for r in want) > reach
    env = VecLeadMonomialsEnv(dist, batch=B, k=2)
    env.seed(np.arange(B) + 700); env.seed_agent(np.arange(B)); env.reset()
    env.accounting(False)
    R = 1024
    d_obs = torch.empty((B, R, env.cols), dtype=torch.int32, device="cuda")
    d_rew = torch.empty(B, dtype=torch.float64, device="cuda"); d_done = torch.empty(B, dtype=torch.uint8, device="cuda")
    d_rows = torch.empty(B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    if persistent:
        env.persistent(True)
    for _ in range(T // 20):
        env.rollout_device("random", 20, True, stream, d_rew, d_done, d_rows, d_obs, R, False, True)
    env.sync(); torch.cuda.synchronize()
    _assert_equals_oracle_run(env, want, sample_every=8)
    rows = d_rows.cpu().numpy()
    host = env.observations(max_rows=R, fill=False)                  # the block the last step left is the final state's observation
    got = d_obs.cpu().numpy()
    for e in range(B):
        assert np.array_equal(got[e, :rows[e]], host[e, :rows[e]]), e


@pytest.mark.parametrize("caps,lean", [(None, 0), (None, 1), ({"wide_lds_terms": 64}, 1), ({"wide_waves": 3}, 0), ({"wide_waves": -1}, 0)])
def test_exponents_beyond_a_byte_wide_and_general(caps, lean):
    """A fixed ideal whose exponents pass 255 (degrees up to 350 and beyond during the reductions): the wide class cannot pack
    such monomials into its 8-byte sort keys, so the polynomial being reduced lives in HBM as plain monomials, lead terms are
    fetched one by one and merges run on HBM-resident views (bbx_wide.h, tier 3 and the unkeyed remainder path); wide_waves =
    -1 is the wave-per-environment general kernel on the same ideal.  Every step's reward, the counters and the complete
    final state against the oracle."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.ideals import FixedIdealGenerator
    bo = ffi.load("bo")
    F = [[(1, (300, 0, 0, 0)), (1, (0, 2, 1, 0)), (5, (0, 0, 0, 0))],
         [(1, (0, 200, 150, 0)), (1, (1, 0, 0, 1)), (7, (0, 0, 0, 0))],
         [(1, (120, 130, 0, 10)), (1, (0, 0, 3, 0)), (11, (0, 0, 0, 0))],
         [(1, (1, 1, 1, 1)), (2, (0, 0, 0, 2)), (3, (0, 0, 0, 0))]]
    B, T = 3, 60
    env = VecLeadMonomialsEnv(FixedIdealGenerator(F), batch=B, k=2, caps=caps)
    if lean:
        env.accounting(False)
    env.seed_agent(np.arange(B) + 3); env.reset()
    oracles = []
    for e in range(B):
        o = bo.env(fixed=F); o.reset(); oracles.append(o)
    for t in range(T):
        rew, done, rows = env.rollout("random", 1, auto_reset=False)
        for e, o in enumerate(oracles):
            if o.nP == 0:
                continue
            r = o.step(ffi.agent_action(3 + e, t, o.nP))
            assert rew[e] == r and rows[e] == o.nP, (t, e, rew[e], r)
    for e, o in enumerate(oracles):
        basis, pairs, order = env.state(e)
        assert max(max(sum(x) for x in ex) for _, ex in basis) > 255          # (the point of the test)
        assert np.array_equal(_state_words(basis, pairs, order), _state_words(o.basis(), o.pairs(), o.reducer_order())), e


def test_capacity_growth_keeps_owed_steps_across_async_launches():
    """Three asynchronous launches queued behind each other, none synchronised: an environment that stops for room in the
    first keeps adding the later launches' steps to what it owes and takes them all once bbx_sync has enlarged the records."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    dist, B, T = "3-5-4-0.5-uniform", 33, 40
    want = bo.run_random_many(dist, 2, range(500, 500 + B), range(B), 3 * T, True, 0)
    env = VecLeadMonomialsEnv(dist, batch=B, k=2, caps={"max_basis": 8, "max_pairs": 8, "arena_terms": 64, "max_poly_terms": 8})
    env.seed(np.arange(B) + 500); env.seed_agent(np.arange(B)); env.reset()
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(3):
        env.rollout_device("random", T, True, torch.cuda.current_stream().cuda_stream, rows=rows)
    env.sync()
    _assert_equals_oracle_run(env, want)
    assert np.array_equal(rows.cpu().numpy(), np.array([r["nP"] for r in want]))


def test_capacity_growth_wide_class_and_strategies():
    """Fixed ideals on the wide (one workgroup per environment) class with a tiny arena and scratch: cyclic-5 under the
    seeded std::random selection — whose engine state must rewind when a step is taken again — and Degree, to completion."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.ideals import FixedIdealGenerator, cyclic
    bo = ffi.load("bo")
    for sel, seed in (("degree", None), ("random", 77), ("sugar", None)):
        env = VecLeadMonomialsEnv(FixedIdealGenerator(cyclic(5)), batch=3, k=1, caps={"max_basis": 8, "max_pairs": 16, "arena_terms": 64, "max_poly_terms": 16})
        if seed is not None:
            env.seed_strategy(seed)
        env.reset()
        env.rollout("random_std" if seed is not None else sel, 1 << 30, auto_reset=False)
        st = env.stats()
        _, w = bo.buchberger(bo.cyclic(5), selection=sel, want_basis=False, seed=seed)
        for e in range(3):
            assert (st[e, 3], st[e, 0] - st[e, 3], st[e, 1]) == (w["zero_reductions"], w["nonzero_reductions"], w["polynomial_additions"]), (sel, e)
        assert env.capacities()["grown"] > 0


def test_full_size_non_binomial_8_variables_vs_oracle():
    """8-4-4-0.5-uniform at B=4096 x T=25 on the general class at DEFAULT capacities (round 2: BBX_E_CAPACITY in environment
    3245): counters of all environments, the complete final state of every 16th."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    dist, B, T = "8-4-4-0.5-uniform", 4096, 25
    want = bo.run_random_many(dist, 2, range(1000, 1000 + B), range(B), T, True, 0)
    env = VecLeadMonomialsEnv(dist, batch=B, k=2)
    env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
    env.accounting(False)
    env.rollout("random", T, auto_reset=True)
    _assert_equals_oracle_run(env, want, 16)


def test_algorithmic_byte_counter_matches_oracle():
    """The roofline numerator counted on the device equals the oracle's count (SURVEY 8d formula)."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T, k = 4, 200, 2
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=k)
    env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
    env.rollout("random", T, auto_reset=True)
    st = env.stats()
    for e in range(B):
        o = bo.env("3-20-10-weighted"); o.seed(1000 + e); o.reset()
        total = adds = 0
        for t in range(T):
            r = o.step(ffi.agent_action(e, t, o.nP))
            total += o.last_step_bytes() + 4 * o.nP * 2 * 3 * k
            adds += int(-r)
            if o.nP == 0:
                o.reset()
        assert (st[e, 0], st[e, 1], st[e, 6]) == (T, adds, total)


def _state_words(basis, pairs, order):
    words = [poly_words(c, x) for c, x in basis]
    return np.concatenate(words + [np.asarray(pairs, np.int32).ravel(), np.asarray(order, np.int32)])


def test_full_size_headline_untraced_vs_oracle():
    """BASELINE configs[1] at full size (B=4096, T=256, the bench workload) on the production
    (non-tracing) kernel: per environment the step/addition/byte counters and the complete final
    state (basis term for term, pair list, reducer order) must equal the oracle's."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T, k = 4096, 256, 2
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=k, caps={"queue_slots": T // 4 + 16})
    env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
    env.rollout("random", T, auto_reset=True)
    st = env.stats()
    assert (st[:, 0] == T).all() and (st[:, 4] == 0).all()
    check_full = set(range(0, B, 64)) | {B - 1}
    for e in range(B):
        o = bo.env("3-20-10-weighted"); o.seed(1000 + e); o.reset()
        adds = total = 0
        for t in range(T):
            r = o.step(ffi.agent_action(e, t, o.nP))
            adds += int(-r)
            total += o.last_step_bytes() + 4 * o.nP * 2 * 3 * k
            if o.nP == 0:
                o.reset()
        assert (st[e, 1], st[e, 6], st[e, 7]) == (adds, total, o.nG), e
        assert int(env.rows[e]) == o.nP, e
        if e in check_full:
            basis, pairs, order = env.state(e)
            want = _state_words(o.basis(), o.pairs(), o.reducer_order())
            assert np.array_equal(_state_words(basis, pairs, order), want), e


def test_value_matches_reference_known_answers():
    """env.value(strategy, gamma) (buchberger.cpp:332-351): device rollouts from a clone, discounted return in double.
    Golden answers recorded from the compiled reference (tests/golden/values.json), compared with ==; unknown
    strategy names select First like the reference's std::map lookup ('env' is what train.py actually passes)."""
    from deepgroebner_amd import CLeadMonomialsEnv
    vals = meta()["values"]
    envs = {}
    for key in sorted(vals):
        parts = key.split("|")
        dist, seed, t, strat = parts[0], int(parts[1]), int(parts[2]), parts[3]
        gamma = 0.9 if len(parts) == 5 else 0.99
        ek = (dist, seed)
        if ek not in envs:
            env = CLeadMonomialsEnv(dist, k=1)
            env.seed(seed)
            envs[ek] = [env, env.reset(), 0]
        env, state, at = envs[ek]
        while at < t:                                   # the walk oracle/make_golden.py did
            state, _, _, _ = env.step(ffi.agent_hash(seed, at) % len(state))
            at += 1
        envs[ek][1], envs[ek][2] = state, at
        assert env.value(strat, gamma) == vals[key], key


def test_values_batch_and_state_untouched():
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B = 6
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
    env.seed(np.arange(B) + 50); obs0 = env.reset()
    want = []
    for e in range(B):
        o = bo.env("3-20-10-weighted"); o.seed(50 + e); o.reset()
        want.append([o.value(s, 0.99) for s in ("degree", "normal", "sugar", "first")])
    for c, s in enumerate(("degree", "normal", "sugar", "first")):
        got = env.values(s, 0.99)
        assert got.tolist() == [w[c] for w in want], s
    obs1 = env.observations()
    assert all(np.array_equal(a, b) for a, b in zip(obs0, obs1))      # value() works on clones
    v = env.value(2, "sample", 0.99)
    assert v >= want[2][0]                                             # best of degree + 100 random rollouts


@pytest.mark.parametrize("dist,caps", [("3-20-10-weighted", None), ("3-20-10-weighted", {"lds_max_basis": 16}), ("3-20-10-weighted", {"lds_max_basis": -1}),
                                       ("5-10-5-uniform", None), ("3-5-4-0.5-uniform", None)])
def test_value_random_and_sample_with_explicit_seeds(dist, caps):
    """value('random') / value('sample') roll out under buchberger(G, P, SelectionType::Random, ..., seed) (buchberger.cpp:
    200-203, 244, 332-351).  With the seeds given, every value equals the oracle's seeded rollout from the same state, ==
    on doubles; 'sample' is the best of the Degree rollout and the 100 seeded Random ones.  Register/LDS-resident class,
    its HBM-resident continuation, the HBM-resident binomial class and the general class."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T0 = 12, 5
    env = VecLeadMonomialsEnv(dist, batch=B, k=2, caps=caps)
    env.seed(np.arange(B) + 70); env.seed_agent(np.arange(B)); env.reset()
    env.rollout("random", T0, auto_reset=True)                      # somewhere inside an episode
    states = []
    for e in range(B):
        o = bo.env(dist); o.seed(70 + e); o.reset()
        for t in range(T0):
            o.step(ffi.agent_action(e, t, o.nP))
            if o.nP == 0:
                o.reset()
        G = [[(int(c), tuple(int(x) for x in ex)) for c, ex in zip(cs, es)] for cs, es in o.basis()]
        states.append((G, [tuple(int(x) for x in p) for p in o.pairs()]))
    rng = np.random.default_rng(11)
    seeds = rng.integers(-2 ** 31, 2 ** 31 - 1, size=B)
    seeds[:3] = (0, 2147483647, -1)                                 # x = seed mod (2^31 - 1), 0 becomes 1
    def ret(e, selection, seed=None):
        G, P = states[e]
        if not P:
            return 0.0
        return bo.buchberger(G, P, selection=selection, seed=None if seed is None else int(seed), want_basis=False)[1]["discounted_return"]
    got = env.values("random", 0.99, seeds=seeds)
    assert got.tolist() == [ret(e, "random", seeds[e]) for e in range(B)]
    seeds2 = rng.integers(0, 2 ** 31 - 1, size=(B, 100))
    got = env.values("sample", 0.99, seeds=seeds2)
    want = [max([ret(e, "degree")] + [ret(e, "random", s) for s in seeds2[e]]) for e in range(B)]
    assert got.tolist() == want
    assert (env.values("sample", 0.99) >= env.values("degree", 0.99)).all()      # (seeds drawn by the handle)


def test_bad_action_is_an_error_not_ub():
    from deepgroebner_amd import CLeadMonomialsEnv, _ffi
    env = CLeadMonomialsEnv("3-20-10-weighted", k=1)
    env.seed(3)
    state = env.reset()
    with pytest.raises(_ffi.BbxError) as ei:
        env.step(len(state))                       # the reference indexes P out of bounds here (buchberger.cpp:399)
    assert ei.value.code == -6
    state = env.reset()                            # a reset clears the error
    _, r, _, _ = env.step(0)
    assert r <= -1.0


def test_lead_monomial_padding_and_k_sweep():
    """lead_monomials_vector zero-padding when a polynomial has fewer than k terms (buchberger.cpp:363-368),
    for k = 1..4 on binomial ideals (two terms) against the oracle."""
    from deepgroebner_amd import CLeadMonomialsEnv
    bo = ffi.load("bo")
    for k in (1, 2, 3, 4):
        env = CLeadMonomialsEnv("3-20-10-uniform", k=k); env.seed(11)
        o = bo.env("3-20-10-uniform"); o.seed(11); o.reset()
        state = env.reset()
        for t in range(25):
            assert state.shape[1] == 2 * 3 * k and np.array_equal(state, o.obs(k))
            if len(state) == 0:
                break
            a = t % len(state)
            state, r, done, _ = env.step(a)
            assert r == o.step(a)


@pytest.mark.parametrize("dist", ["2-8-4-weighted", "4-5-4-weighted", "6-4-4-uniform", "7-3-4-weighted", "8-3-4-weighted",
                                  "2-5-4-0.8-uniform", "6-3-4-0.5-weighted"])
def test_observation_width_sweep(dist):
    """Every row width the observation writer knows how to store (n = 2..7 variables: dword / dwordx2 / dwordx4
    combinations), binomial and general classes, k = 1 and 3, against the oracle for a whole episode prefix."""
    from deepgroebner_amd import CLeadMonomialsEnv
    bo = ffi.load("bo")
    n = int(dist.split("-")[0])
    for k in (1, 3):
        env = CLeadMonomialsEnv(dist, k=k); env.seed(5)
        o = bo.env(dist); o.seed(5); o.reset()
        state = env.reset()
        for t in range(40):
            assert state.shape[1] == 2 * n * k and np.array_equal(state, o.obs(k)), (dist, k, t)
            if len(state) == 0:
                break
            a = (3 * t + 1) % len(state)
            state, r, done, _ = env.step(a)
            assert r == o.step(a)


@pytest.mark.parametrize("sort_input", [False, True])
@pytest.mark.parametrize("dist", ["3-20-10-weighted", "3-20-10-uniform-consts", "3-8-6-maximum-pure-homog", "5-10-5-uniform",
                                  "7-4-4-weighted-homog", "3-6-5-0.5-uniform", "4-4-4-1.5-weighted-consts", "3-5-4-2.0-maximum-homog",
                                  "8-4-5-uniform", "8-2-3-0.5-weighted"])
def test_device_drawn_ideals_equal_host_drawn(dist, sort_input, monkeypatch):
    """Random distributions — binomial and polynomial (Poisson term counts, sums of single terms, 1/LC scaling) — are
    drawn inside the kernels (minstd_rand0 + libstdc++'s distributions restated on the device); BBX_HOST_GEN=1 selects the
    host generators + ideal queue instead.  Same seeds, same rollout: identical
    counters and final states, episode after episode (each reset consumes the next ideal of the stream)."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    B, T = 6, {"5-10-5-uniform": 6000, "8-4-5-uniform": 6000}.get(dist, 400)
    envs = []
    for host in (False, True):
        if host:
            monkeypatch.setenv("BBX_HOST_GEN", "1")
        env = VecLeadMonomialsEnv(dist, batch=B, k=1, sort_input=sort_input)   # (sort_input: sorted on the device / by the host's std::sort)
        env.seed(np.arange(B) * 7 + 3); env.seed_agent(np.arange(B) + 50); env.reset()
        env.rollout("random", T, auto_reset=True)
        envs.append(env)
    a, b = envs
    sa, sb = a.stats(), b.stats()
    assert (sa[:, 2] >= 2).all(), "too few episodes to say anything about later ideals of the stream"
    assert np.array_equal(sa[:, :5], sb[:, :5]) and np.array_equal(sa[:, 7], sb[:, 7])
    for e in range(B):
        ba, pa, oa = a.state(e); bb, pb, ob = b.state(e)
        assert np.array_equal(_state_words(ba, pa, oa), _state_words(bb, pb, ob)), e


def test_generator_failure_surfaces_when_the_ideal_is_needed():
    """1-variable binomials: the generator soon fails to draw two distinct monomials (the reference throws,
    ideals.cpp:190-192).  Ideals are drawn ahead of time here (a ring of 8 per environment), but the error must appear
    at the reset that consumes the failing draw — where the reference raises — and not earlier; the resets before it
    match the oracle."""
    from deepgroebner_amd import CLeadMonomialsEnv, _ffi
    bo = ffi.load("bo")
    g = bo.generator("1-6-3-uniform"); g.seed(5)
    first_bad = next(i for i in range(10000) if len(g.next()) < 3)     # a failed draw leaves the ideal incomplete
    assert 1 <= first_bad < 200
    env = CLeadMonomialsEnv("1-6-3-uniform", k=1); env.seed(5)
    o = bo.env("1-6-3-uniform"); o.seed(5)
    for episode in range(first_bad):
        o.reset()
        state = env.reset()
        assert np.array_equal(state, o.obs(1)), episode
        assert len(state) > 0                                          # (no redraws: one ideal per reset)
    with pytest.raises(_ffi.BbxError, match="distinct"):
        env.reset()


def test_padded_observation_block():
    """[batch, max_rows, cols] with -1 fill: the layout the reference's agents build on the host (pg.py:217-226)."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=3, k=2)
    env.seed([1, 2, 3]); ragged = env.reset()
    mr = int(env.rows.max()) + 5
    block = env.observations(max_rows=mr, fill=True)
    assert block.shape == (3, mr, 12)
    for e in range(3):
        assert np.array_equal(block[e, :env.rows[e]], ragged[e]) and (block[e, env.rows[e]:] == -1).all()


def test_rollout_device_with_torch_buffers():
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T, k = 8, 40, 2
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=k)
    env.seed(np.arange(B) + 9); env.seed_agent(np.arange(B) + 100); env.reset()
    obs = torch.full((B, 96, env.cols), -7, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda")
    done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream()
    env.rollout_device("random", T, True, s.cuda_stream, rew, done, rows, obs, 96, True, True)
    env.sync()
    torch.cuda.synchronize()
    for e in range(B):
        o = bo.env("3-20-10-weighted"); o.seed(9 + e); o.reset()
        r = 0.0
        for t in range(T):
            r = o.step(ffi.agent_action(100 + e, t, o.nP))
            d = o.nP == 0
            if d:
                o.reset()
        assert float(rew[e]) == r and bool(done[e]) == d and int(rows[e]) == o.nP
        got = obs[e].cpu().numpy()
        assert np.array_equal(got[:o.nP], o.obs(k)) and (got[o.nP:] == -1).all()


@pytest.mark.parametrize("dist,lean", [("5-10-5-uniform", 1), ("5-10-5-uniform", 0), ("4-8-6-weighted", 1), ("3-20-10-weighted", 1)])
def test_observation_block_written_incrementally_equals_the_full_observation(dist, lean):
    """The HBM-resident binomial class rewrites, at every step of a launch, only the rows of the caller's block from the first
    pair that left the pair set on (bbx_binom.h: the rows in front of it stand as the previous step wrote them).  The block after
    launches of 1, 2, 3, 17 and 60 steps must be the oracle's full observation (buchberger.cpp:354-370, 391-394) of the state
    reached, for every environment; the block is handed over dirty (-7 everywhere) and rows beyond |P| are the caller's."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, k, R = 24, 2, 1024
    env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps={"lds_max_basis": -1})       # (3 variables: HBM-resident class forced)
    env.seed(np.arange(B) + 31); env.seed_agent(np.arange(B) + 5); env.reset()
    if lean:
        env.accounting(False)
    obs = torch.full((B, R, env.cols), -7, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    oracles = []
    for e in range(B):
        o = bo.env(dist); o.seed(31 + e); o.reset(); oracles.append(o)
    t = 0
    for n in (1, 2, 3, 17, 60, 1, 40):
        env.rollout_device("random", n, True, s, rew, done, rows, obs, R, False, True)
        env.sync(); torch.cuda.synchronize()
        got, nrows = obs.cpu().numpy(), rows.cpu().numpy()
        for e, o in enumerate(oracles):
            for tt in range(t, t + n):
                o.step(ffi.agent_action(5 + e, tt, o.nP))
                if o.nP == 0:
                    o.reset()
            assert nrows[e] == o.nP, (dist, e, t)
            assert np.array_equal(got[e, :o.nP], o.obs(k)), (dist, e, t, n)
        t += n


def test_headline_kernel_variant_vs_oracle():
    """The launch shape bench.py times — counter-hash agent, 3 variables, k = 2, observation written after every step
    without fill, auto-reset — takes the compile-time specialised kernel; a few launches of it (episodes end and restart
    inside them) against the oracle: last reward / done / rows, the observation block, the step and addition counters
    and the complete final state."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, k = 48, 2
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=k)
    env.seed(np.arange(B) + 300); env.seed_agent(np.arange(B) + 17); env.reset()
    obs = torch.zeros((B, 128, env.cols), dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda")
    done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream()
    oracles, tcount, adds = [], [0] * B, [0] * B
    for e in range(B):
        o = bo.env("3-20-10-weighted"); o.seed(300 + e); o.reset(); oracles.append(o)
    for T in (1, 63, 130, 200):
        env.rollout_device("random", T, True, s.cuda_stream, rew, done, rows, obs, 128, False, True)
        env.sync(); torch.cuda.synchronize()
        for e, o in enumerate(oracles):
            r = 0.0
            for _ in range(T):
                r = o.step(ffi.agent_action(17 + e, tcount[e], o.nP)); tcount[e] += 1; adds[e] += int(-r)
                d = o.nP == 0
                if d:
                    o.reset()
            assert float(rew[e]) == r and bool(done[e]) == d and int(rows[e]) == o.nP, (T, e)
            assert np.array_equal(obs[e, :o.nP].cpu().numpy(), o.obs(k)), (T, e)
    st = env.stats()
    assert st[:, 0].tolist() == tcount and st[:, 1].tolist() == adds
    for e in (0, B // 2, B - 1):
        basis, pairs, order = env.state(e)
        assert np.array_equal(_state_words(basis, pairs, order),
                              _state_words(oracles[e].basis(), oracles[e].pairs(), oracles[e].reducer_order())), e


@pytest.mark.parametrize("wide,lds_terms,lean", [(0, 0, 0), (-1, 0, 0), (3, 0, 0), (8, 256, 0), (5, 1000, 0),
                                                 (0, 0, 1), (8, 256, 1), (4, 64, 1), (5, 1000, 1),
                                                 (0, 3000, 0), (0, 4096, 1)])     # (more than 160 KB of LDS asked for: clamped)
def test_long_polynomials_cyclic7_all_merge_paths(wide, lds_terms, lean):
    """cyclic-7 far enough into an episode that polynomials have hundreds to thousands of terms: exercises the
    merge-path tiled merge of the general class (one wave per environment, wide = -1) and the wide class (one
    workgroup per environment: polynomial being reduced in LDS, reducer tails streamed through the LDS window in chunks,
    polynomials that outgrow LDS continued on the HBM-resident buffers) against the oracle: per-step rewards via
    counters, and the complete final state."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T, k = 3, 110, 2
    env = VecLeadMonomialsEnv("cyclic-7", batch=B, k=k, caps={"wide_waves": wide, "wide_lds_terms": lds_terms, "arena_terms": 1 << 19})
    if lean:
        env.accounting(False)            # the variant with the LDS accumulator (no algorithmic-byte counting)
    env.seed_agent(np.arange(B) + 40); env.reset()
    env.rollout("random", T, auto_reset=False)
    st = env.stats()
    for e in range(B):
        o = bo.env("cyclic-7"); o.reset()
        adds = 0
        for t in range(T):
            if o.nP == 0:
                break
            adds += int(-o.step(ffi.agent_action(40 + e, t, o.nP)))
        assert st[e, 1] == adds and st[e, 7] == o.nG and int(env.rows[e]) == o.nP
        basis, pairs, order = env.state(e)
        assert max(len(c) for c, _ in basis) > 600                     # long enough to take the cooperative path
        want = _state_words(o.basis(), o.pairs(), o.reducer_order())
        assert np.array_equal(_state_words(basis, pairs, order), want), e


@pytest.mark.parametrize("waves", [3, 5, 8])
@pytest.mark.parametrize("lean", [0, 1])
def test_wide_class_lockstep_stress(waves, lean):
    """The wide class keeps the waves of a workgroup in lockstep through hand-placed barriers and parity-buffered exchange
    slots (bbx_wide.h): a missing barrier or a branch on a value that is not workgroup-uniform shows as a wrong reduction only
    for particular combinations of workgroup width, LDS capacities (which decide the merge tier of every round) and
    polynomial lengths.  Sweep: widths 3 / 5 / 8 waves x LDS windows of 40 .. 1000 terms x the random agent (lazy accumulator
    when lean) and the Degree strategy (eager merges, four-term run scan) on cyclic-6, 150 steps, against the oracle's
    counters and complete final states."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T, k = 2, 150, 1
    want_hash = bo.run_random_many("cyclic-6", k, [0] * B, range(B), T, True, 0)
    for lds_terms in (40, 64, 128, 256, 1000):
        for agent in ("random", "degree"):
            env = VecLeadMonomialsEnv("cyclic-6", batch=B, k=k, caps={"wide_waves": waves, "wide_lds_terms": lds_terms})
            env.seed_agent(np.arange(B)); env.reset()
            if lean:
                env.accounting(False)
            env.rollout(agent, T, auto_reset=True)
            st = env.stats()
            assert (st[:, 4] == 0).all()
            if agent == "random":
                for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
                    assert np.array_equal(st[:, col], np.array([r[key] for r in want_hash])), (waves, lds_terms, lean, key)
                for e in range(B):
                    assert fnv64(_state_words(*env.state(e))) == want_hash[e]["state_hash"], (waves, lds_terms, lean, e)
            else:                                       # both environments run the same strategy on the same ideal
                o = bo.env("cyclic-6"); o.reset()
                adds = 0
                for t in range(T):
                    if o.nP == 0:
                        o.reset()
                    adds += int(-o.step(degree_action(o)))
                assert (st[:, 1] == adds).all() and (st[:, 7] == o.nG).all(), (waves, lds_terms, lean)
                assert np.array_equal(_state_words(*env.state(0)), _state_words(o.basis(), o.pairs(), o.reducer_order())), (waves, lds_terms, lean)
            del env


def test_step_with_observation_when_the_records_and_the_observation_block_grow_in_the_same_call():
    """Found by scripts/fuzz_gym.py (seed 20261, round 1518): a host step (bbx_step_obs) whose step enlarged the records — an
    intermediate polynomial longer than max_poly_terms — AND left a pair set taller than the observation block of the call.  The
    second attempt (block enlarged, observation rewritten) ran with the first attempt's parameters: the freed records, whose
    stale status then drove the growth loop to its 2^22-term limit.  The trajectory of the find, against the oracle: five
    5-3-3-0.5-uniform environments under LCM elimination, host-supplied actions, finished environments reset now and then."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    dist, elim, rew, seed0, B, T, k = "5-3-3-0.5-uniform", "lcm", "additions", 867467, 5, 27, 2
    arng = np.random.default_rng(seed0)
    env = VecLeadMonomialsEnv(dist, B, elim, rew, False, True, k, 0, None, "python")
    env.seed(np.arange(B) + seed0)
    os_ = []
    for e in range(B):
        o = bo.env(dist, elimination=elim, rewards=rew); o.seed(seed0 + e); o.reset(); os_.append(o)
    obs = env.reset()
    grown = 0
    for t in range(T):
        for e in range(B):
            assert np.array_equal(obs[e], os_[e].obs(k)), (t, e)
        acts = np.array([arng.integers(0, max(1, os_[e].nP)) for e in range(B)], dtype=np.int32)
        live = [os_[e].nP > 0 for e in range(B)]
        obs, r, d, _ = env.step(acts)
        mask = np.zeros(B, dtype=np.uint8)
        for e in range(B):
            if not live[e]:
                continue
            assert r[e] == os_[e].step(int(acts[e])) and bool(d[e]) == (os_[e].nP == 0), (t, e)
            if os_[e].nP == 0 and arng.random() < 0.7:
                mask[e] = 1
        if mask.any():
            obs = env.reset(mask)
            for e in np.flatnonzero(mask):
                os_[e].reset()
        grown = env.capacities()["grown"]
    assert grown >= 1 and max(o.nP for o in os_) > 256          # (both happened: the records grew, a pair set passed the first block)


def test_kernels_per_call():
    """What a call costs in launches (bbx_kernels_launched): one kernel for a reset (the Python reset() is two calls: the reset
    and the observation it returns), for a host-driven step of a small batch and for a rollout of a class without a
    continuation pass; two where the register/LDS class has the HBM-resident pass behind it, where
    the general class hands long polynomials to the wide class, and for a wide-class launch of more workgroups than CUs.  (A
    misplaced `else` once added the classes' kernels to every launch: nothing failed, every call paid for it.)"""
    from deepgroebner_amd import VecLeadMonomialsEnv
    def cost(env, fn):
        k0 = env.kernels_launched(); fn(); env.sync(); return env.kernels_launched() - k0
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=64, k=2)
    env.seed(np.arange(64)); env.accounting(False)
    assert cost(env, env.reset) == 2
    assert cost(env, lambda: env.rollout("random", 16, auto_reset=True)) == 2          # fast class + HBM-resident continuation
    small = VecLeadMonomialsEnv("3-20-10-weighted", batch=4, k=2)
    small.seed(np.arange(4)); small.accounting(False); small.reset()
    assert cost(small, lambda: small.step(np.zeros(4, dtype=np.int32))) == 1            # zero-copy host step: the hand-tuned kernel alone
    u5 = VecLeadMonomialsEnv("5-10-5-uniform", batch=64, k=2)
    u5.seed(np.arange(64)); u5.accounting(False)
    assert cost(u5, u5.reset) == 2
    assert cost(u5, lambda: u5.rollout("random", 16, auto_reset=True)) == 1             # HBM-resident binomial class
    cyc = VecLeadMonomialsEnv("cyclic-6", batch=8, k=1)
    assert cost(cyc, cyc.reset) == 2
    assert cost(cyc, lambda: cyc.rollout("random", 8, auto_reset=True)) == 1            # wide class, one workgroup per CU
    big = VecLeadMonomialsEnv("cyclic-6", batch=300, k=1)
    big.reset()
    assert cost(big, lambda: big.rollout("random", 4, auto_reset=True)) == 2            # ... more workgroups than CUs
    gen = VecLeadMonomialsEnv("3-5-4-0.5-uniform", batch=64, k=2)
    gen.seed(np.arange(64))
    assert cost(gen, gen.reset) == 2
    assert cost(gen, lambda: gen.rollout("random", 8, auto_reset=True)) == 2            # general class + wide class behind it


def test_wide_class_two_kernel_launch_for_more_workgroups_than_cus():
    """A wide-class launch of more workgroups than the device has CUs is two kernels (BbxParams::wide_tail): the first runs two
    workgroups per CU and stops — at step boundaries, BBX_ST_TIMESLICE — once the workgroups still at work would fit one per
    CU; the second takes the steps still owed.  Nothing of that may show: every environment of a batch of 300 against the
    oracle (counters of all, complete states of a sample), lean and accounting variants, and the same batch with the second
    kernel switched off; the general class's hand-over of long polynomials (5-4-4-1.0-uniform) likewise."""
    import os
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T, k = 300, 100, 1
    want = bo.run_random_many("cyclic-6", k, [0] * B, range(B), T, True, 0)
    for lean, tail in ((1, 1), (0, 1), (1, 0)):
        if not tail:
            os.environ["BBX_NO_WIDE_TAIL"] = "1"
        try:
            env = VecLeadMonomialsEnv("cyclic-6", batch=B, k=k)
            env.seed_agent(np.arange(B)); env.reset()
            if lean:
                env.accounting(False)
            env.timing(True)
            env.rollout("random", T, auto_reset=True)
            _, launches = env.timing(False)
        finally:
            os.environ.pop("BBX_NO_WIDE_TAIL", None)
        assert launches == (2 if tail else 1), (lean, tail, launches)
        st = env.stats()
        assert (st[:, 4] == 0).all()
        for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
            assert np.array_equal(st[:, col], np.array([r[key] for r in want])), (lean, tail, key)
        for e in list(range(0, B, 23)) + [int(st[:, 1].argmax())]:      # (the slowest environment certainly went through the second kernel)
            assert fnv64(_state_words(*env.state(e))) == want[e]["state_hash"], (lean, tail, e)
        del env
    B, T, k = 320, 48, 2
    seeds = list(range(1000, 1000 + B))
    want = bo.run_random_many("5-4-4-1.0-uniform", k, seeds, range(B), T, True, 0)
    env = VecLeadMonomialsEnv("5-4-4-1.0-uniform", batch=B, k=k)
    env.seed(np.array(seeds)); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
    env.rollout("random", T, auto_reset=True)
    st = env.stats()
    assert (st[:, 4] == 0).all()
    for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
        assert np.array_equal(st[:, col], np.array([r[key] for r in want])), key
    for e in list(range(0, B, 37)) + [int(st[:, 1].argmax())]:
        assert fnv64(_state_words(*env.state(e))) == want[e]["state_hash"], e



def test_wide_class_32_byte_monomials_cyclic8():
    """cyclic-8 lives in the reference's full N = 8 ring (polynomials.h:29): 32-byte monomials have no 8-byte sort key, so the
    wide class runs its unkeyed regime throughout (h as plain monomials in the record's scratch, merges on HBM-resident
    views).  60 steps of the random agent and of the Degree strategy against the oracle, counters and complete states; the
    wave-per-environment general class (wide_waves = -1) on the same ideal as the cross-check."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, T, k = 2, 60, 2
    want = bo.run_random_many("cyclic-8", k, [0] * B, range(B), T, True, 0)
    for caps in (None, {"wide_waves": 3, "wide_lds_terms": 64}, {"wide_waves": -1}):
        for lean in (0, 1):
            env = VecLeadMonomialsEnv("cyclic-8", batch=B, k=k, caps=caps)
            env.seed_agent(np.arange(B)); env.reset()
            if lean:
                env.accounting(False)
            env.rollout("random", T, auto_reset=True)
            st = env.stats()
            assert (st[:, 4] == 0).all()
            for key, col in (("steps", 0), ("additions", 1), ("zero_reductions", 3), ("nG", 7)):
                assert np.array_equal(st[:, col], np.array([r[key] for r in want])), (caps, lean, key)
            for e in range(B):
                assert fnv64(_state_words(*env.state(e))) == want[e]["state_hash"], (caps, lean, e)


def test_strategy_stats_like_make_strat():
    """scripts/make_strat.cpp of the reference: full Buchberger per ideal and strategy -> ZeroReductions,
    NonzeroReductions, PolynomialAdditions.  A mixed list of ideals (random binomial, random dense, cyclic) in one
    batch against the oracle's buchberger() statistics, plus the cyclic known answers of SURVEY 8c."""
    from deepgroebner_amd import strategy_stats
    bo = ffi.load("bo")
    ideals = []
    for dist, seed in (("3-20-10-weighted", 3), ("3-6-5-0.5-uniform", 4), ("4-5-4-weighted", 5), ("3-20-10-uniform", 6)):
        g = bo.generator(dist); g.seed(seed)
        ideals += [g.next() for _ in range(3)]
    ideals += [bo.cyclic(4), bo.cyclic(5)]
    trim = [[[(c, e[:5]) for c, e in f] for f in F] for F in ideals]
    for strategy in ("degree", "normal", "sugar", "first"):
        if strategy == "first":
            sub = trim[:-1]                                   # First explodes on cyclic-5
        else:
            sub = trim
        got = strategy_stats(sub, strategy)
        for n, F in enumerate(sub):
            _, st = bo.buchberger(F, selection=strategy, want_basis=False)
            assert got[n].tolist() == [st["zero_reductions"], st["nonzero_reductions"], st["polynomial_additions"]], (strategy, n)
    got = strategy_stats([trim[-1]], "degree")
    assert got[0].tolist() == [69, 41, 1442]                  # cyclic-5, Degree


def test_strategy_stats_of_ideals_without_pairs():
    """buchberger(F) of an ideal whose generators leave no pair (coprime lead monomials, a single generator) returns at once
    with zero statistics (buchberger.cpp:252-262); in a list such an ideal is its own finished run — BuchbergerEnv::reset's
    redraw (buchberger.cpp:313-314) has no say there — and the ideals next to it are not affected."""
    from deepgroebner_amd import strategy_stats
    bo = ffi.load("bo")
    x, y, z = (1, 0, 0), (0, 1, 0), (0, 0, 1)
    coprime = [[(1, x), (5, (0, 0, 0))], [(1, y), (7, (0, 0, 0))]]
    single = [[(1, (2, 1, 0)), (3, z)]]
    g = bo.generator("3-8-5-weighted"); g.seed(11)
    others = [[[(c, e[:3]) for c, e in f] for f in g.next()] for _ in range(2)]
    ideals = [coprime, others[0], single, others[1], coprime]
    for strategy in ("degree", "first", "spice"):
        got = strategy_stats(ideals, strategy)
        for n, F in enumerate(ideals):
            _, st = bo.buchberger(F, selection=strategy, want_basis=False)
            assert got[n].tolist() == [st["zero_reductions"], st["nonzero_reductions"], st["polynomial_additions"]], (strategy, n)
        assert got[0].tolist() == [0, 0, 0] and got[2].tolist() == [0, 0, 0] and got[4].tolist() == [0, 0, 0]


def test_strategy_stats_reversed_and_seeded_random():
    """The remaining SelectionType values of make_strat.cpp:49-59: Last / Codegree / Strange / Spice (maximum instead
    of minimum, buchberger.cpp:207-240) and Random drawn from std::default_random_engine seeded per run
    (buchberger.cpp:200-203, 244) — the same seed for every ideal, as make_strat.cpp:66 passes it."""
    from deepgroebner_amd import strategy_stats
    bo = ffi.load("bo")
    ideals = []
    for dist, seed in (("3-20-10-weighted", 3), ("3-6-5-0.5-uniform", 4), ("4-5-4-weighted", 5), ("3-20-10-uniform", 6)):
        g = bo.generator(dist); g.seed(seed)
        ideals += [g.next() for _ in range(3)]
    ideals.append(bo.cyclic(4))
    trim = [[[(c, e[:5]) for c, e in f] for f in F] for F in ideals]
    binom = trim[:3] + trim[9:12]                             # one batch that takes the binomial kernel class
    for strategy, seed in (("last", None), ("codegree", None), ("strange", None), ("spice", None), ("random", 5),
                           ("random", -7), ("random", 2147483647)):
        for sub in (trim, binom):
            got = strategy_stats(sub, strategy, seed=seed)
            for n, F in enumerate(sub):
                _, st = bo.buchberger(F, selection=strategy, want_basis=False, seed=seed)
                assert got[n].tolist() == [st["zero_reductions"], st["nonzero_reductions"], st["polynomial_additions"]], (strategy, seed, n)


def test_make_strat_pipeline(tmp_path):
    """scripts/make_dist.py -> scripts/make_strat.py: the CSV files of the reference's pipeline (make_strat.cpp:22-72),
    and its exit codes for a missing input (2) and an existing output (3)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = lambda *a: subprocess.run([sys.executable, *a], cwd=tmp_path, capture_output=True, text=True)   # noqa: E731
    assert run(os.path.join(root, "scripts/make_strat.py"), "3-20-10-weighted", "degree").returncode == 2
    r = run(os.path.join(root, "scripts/make_dist.py"), "3-20-10-weighted", "20", "9")
    assert r.returncode == 0, r.stderr
    r = run(os.path.join(root, "scripts/make_strat.py"), "3-20-10-weighted", "degree")
    assert r.returncode == 0, r.stderr
    assert run(os.path.join(root, "scripts/make_strat.py"), "3-20-10-weighted", "degree").returncode == 3
    r = run(os.path.join(root, "scripts/make_strat.py"), "3-20-10-weighted", "random", "11")
    assert r.returncode == 0, r.stderr
    bo = ffi.load("bo")
    from deepgroebner_amd import parse_ideal_string
    d = tmp_path / "data/stats/3-20-10-weighted"
    lines = (d / "3-20-10-weighted.csv").read_text().splitlines()
    assert lines[0] == "Ideal" and len(lines) == 21
    g = bo.generator("3-20-10-weighted"); g.seed(9)
    for name, sel, seed in (("3-20-10-weighted_degree.csv", "degree", None), ("3-20-10-weighted_random_11.csv", "random", 11)):
        out = (d / name).read_text().splitlines()
        assert out[0] == "ZeroReductions,NonzeroReductions,PolynomialAdditions" and len(out) == 21
        for n, line in enumerate(lines[1:]):
            F = parse_ideal_string(line)
            _, st = bo.buchberger(F, selection=sel, want_basis=False, seed=seed)
            assert out[1 + n] == "%d,%d,%d" % (st["zero_reductions"], st["nonzero_reductions"], st["polynomial_additions"])
    for line in lines[1:]:                                    # the file holds the generator's own stream
        assert parse_ideal_string(line) == g.next()


def test_step_obs_ragged_block_and_growth():
    """bbx_step_obs (what env.step()/reset() call): the ragged observation block against the oracle's matrices on a
    distribution whose pair sets outgrow the initial 128-row device block (5-10-5-uniform: |P| in the hundreds), so the
    block is enlarged and rewritten mid-run; step_ragged's flat/offsets view agrees with the list view."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, k = 3, 2
    env = VecLeadMonomialsEnv("5-10-5-uniform", batch=B, k=k)
    env.seed(np.arange(B) + 70)
    oracles = []
    for e in range(B):
        o = bo.env("5-10-5-uniform"); o.seed(70 + e); o.reset(); oracles.append(o)
    obs = env.reset()
    for e, o in enumerate(oracles):
        assert np.array_equal(obs[e], o.obs(k))
    grown = False
    for t in range(700):
        acts = np.array([(5 * t + e) % max(o.nP, 1) for e, o in enumerate(oracles)], dtype=np.int32)
        if t % 2:
            flat, off, r, d = env.step_ragged(acts, auto_reset=True)
            obs = [flat[off[e]:off[e + 1]] for e in range(B)]
            assert off[0] == 0 and off[-1] == flat.shape[0]
        else:
            obs, r, d, _ = env.step(acts, auto_reset=True)
        for e, o in enumerate(oracles):
            assert r[e] == o.step(int(acts[e]))
            if o.nP == 0:
                assert d[e]
                o.reset()
            assert np.array_equal(obs[e], o.obs(k)), (t, e)
            grown |= o.nP > 128
    assert grown, "the run never exceeded the initial observation block: pick a longer one"


def test_step_autoreset_vec_convention():
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, k = 4, 1
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=k)
    env.seed(np.arange(B) + 400)
    oracles = []
    for e in range(B):
        o = bo.env("3-20-10-weighted"); o.seed(400 + e); o.reset(); oracles.append(o)
    obs = env.reset()
    for t in range(160):
        acts = np.array([(7 * t + 3 + e) % o.nP for e, o in enumerate(oracles)], dtype=np.int32)
        obs, r, d, _ = env.step(acts, auto_reset=True)
        for e, o in enumerate(oracles):
            assert r[e] == o.step(int(acts[e]))
            fin = o.nP == 0
            assert bool(d[e]) == fin
            if fin:
                o.reset()
            assert np.array_equal(obs[e], o.obs(k)), (t, e)


def test_reduced_groebner_basis_known_answers():
    """buchberger() = interreduce(minimalize(G)) after the rollout (buchberger.cpp:265): cyclic-4/5/6 with Degree
    selection against the reduced bases recorded from the reference (sizes 7/20/45) and a random ideal
    against the oracle."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    from oracle.trace import flat_ideal
    bo = ffi.load("bo")
    want = meta()["buchberger"]
    for n in (4, 5, 6):
        env = VecLeadMonomialsEnv("cyclic-%d" % n, batch=1, k=1)
        env.reset()
        env.rollout("degree", 1 << 30, auto_reset=False)
        st = env.stats()[0]
        w = want["cyclic-%d|degree" % n]
        assert [st[3], st[0] - st[3], st[1]] == [w["zero_reductions"], w["nonzero_reductions"], w["polynomial_additions"]]
        G = env.reduced_basis(0)
        assert len(G) == w["basis_size"] and int(fnv64(flat_ideal(G))) == w["basis_hash"]
    g = bo.generator("3-6-5-0.5-uniform"); g.seed(21); F = g.next()
    env = VecLeadMonomialsEnv([[[(c, e[:3]) for c, e in f] for f in F]], batch=1, k=1)
    env.reset(); env.rollout("normal", 1 << 30, auto_reset=False)
    Gref, _ = bo.buchberger(F, selection="normal")
    assert env.reduced_basis(0) == Gref


def test_cyclic7_degree_agent_to_completion():
    """ONE cyclic-7 environment driven by the Degree agent until the pair set is empty — 5 882
    steps with polynomials of up to ~1 800 terms through the cooperative wide kernel — against the numbers recorded from
    the compiled reference: zero / non-zero reductions 4 592 / 1 290, 1 657 785 polynomial additions, a reduced
    Groebner basis of 209 elements (hash of every coefficient and exponent), and value('degree', 0.99) of the reset
    state = -13142.770134825347 (the same rollout in value mode on a device clone), all bit-exact."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    from oracle.trace import flat_ideal
    w = meta()["buchberger"]["cyclic-7|degree"]
    env = VecLeadMonomialsEnv("cyclic-7", batch=1, k=1)
    env.reset()
    assert env.value(0, "degree", 0.99) == w["discounted_return"]       # (value rollouts run the lean variant)
    env.rollout("degree", 1 << 30, auto_reset=False)
    st = env.stats()[0]
    assert [st[3], st[0] - st[3], st[1]] == [w["zero_reductions"], w["nonzero_reductions"], w["polynomial_additions"]]
    assert st[0] == 5882 and int(env.rows[0]) == 0
    G = env.reduced_basis(0)
    assert len(G) == w["basis_size"] == 209 and int(fnv64(flat_ideal(G))) == w["basis_hash"]


def test_interleaved_handles_and_copies():
    """Several handles alive at once (single environments on the zero-copy path, a batch on the copy path, copies of
    both made mid-episode), stepped in turn: each follows its own oracle; a copy continues like its source did."""
    from deepgroebner_amd import CLeadMonomialsEnv, VecLeadMonomialsEnv
    bo = ffi.load("bo")
    a = CLeadMonomialsEnv("3-20-10-weighted", k=2); a.seed(1)
    b = CLeadMonomialsEnv("5-10-5-uniform", k=1); b.seed(2)
    v = VecLeadMonomialsEnv("3-20-10-weighted", batch=12, k=2); v.seed(np.arange(12) + 30)
    oa = bo.env("3-20-10-weighted"); oa.seed(1); oa.reset()
    ob = bo.env("5-10-5-uniform"); ob.seed(2); ob.reset()
    ov = []
    for e in range(12):
        o = bo.env("3-20-10-weighted"); o.seed(30 + e); o.reset(); ov.append(o)
    sa, sb, sv = a.reset(), b.reset(), v.reset()
    copies = None
    for t in range(90):
        assert np.array_equal(sa, oa.obs(2)) and np.array_equal(sb, ob.obs(1))
        for e in range(12):
            assert np.array_equal(sv[e], ov[e].obs(2))
        if t == 40:
            copies = (a.copy(), v.copy(), [o.copy() for o in [oa] + ov])
        if len(sa) == 0:
            oa.reset(); sa = a.reset(); continue
        sa, ra, _, _ = a.step(t % len(sa)); assert ra == oa.step(t % oa.nP)
        if len(sb):
            sb, rb, _, _ = b.step(0); assert rb == ob.step(0)
        acts = np.array([t % max(o.nP, 1) for o in ov], dtype=np.int32)
        sv, rv, dv, _ = v.step(acts, auto_reset=True)
        for e, o in enumerate(ov):
            assert rv[e] == o.step(int(acts[e]))
            if o.nP == 0:
                o.reset()
    ca, cv, oc = copies
    s = ca._vec._step_obs(None, False)[0]
    assert np.array_equal(s, oc[0].obs(2))
    for t in range(30):
        if oc[0].nP == 0:
            break
        s, r, _, _ = ca.step(0); assert r == oc[0].step(0) and np.array_equal(s, oc[0].obs(2))
    acts = np.zeros(12, dtype=np.int32)
    for t in range(30):
        svv, rv, dv, _ = cv.step(acts, auto_reset=True)
        for e in range(12):
            o = oc[1 + e]
            assert rv[e] == o.step(0)
            if o.nP == 0:
                o.reset()
            assert np.array_equal(svv[e], o.obs(2))


def test_in_batch_clones_for_tree_search():
    """env.copy() per search node (mcts.py:89,96,147) as an in-batch clone: the clone continues exactly like its
    source, including the ideals it will draw after a reset, and then diverges under different actions."""
    from deepgroebner_amd import VecLeadMonomialsEnv
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=4, k=2)
    env.seed([11, 12, 13, 14]); env.reset()
    env.step([0, 1, 2, 3])
    env.clone_envs([0, 1], [2, 3])
    o = env.observations()
    assert np.array_equal(o[0], o[2]) and np.array_equal(o[1], o[3])
    for t in range(80):                                   # long enough to cross resets: the generator state travelled too
        acts = np.array([t % r if r else 0 for r in env.rows], dtype=np.int32)
        acts[2], acts[3] = acts[0], acts[1]
        obs, r, d, _ = env.step(acts, auto_reset=True)
        assert r[0] == r[2] and r[1] == r[3] and d[0] == d[2] and d[1] == d[3]
        assert np.array_equal(obs[0], obs[2]) and np.array_equal(obs[1], obs[3])
    env.step([0, 0, 1 % max(1, env.rows[2]), 0], auto_reset=True)
    with pytest.raises(Exception):
        env.clone_envs([0], [0])


# ---- full-size runs of every single-GPU BASELINE config on the kernels that are timed (bench.py / scripts/bench_configs.py):
# lean (no-accounting) variants through rollout_device with the observation written every step, and the accounting variants
# for the algorithmic-byte counters.  The oracle runs every environment on all host cores (oracle.ffi run_random_many).
_ORACLE_RUNS = {}


def _full_size(dist, B, T, k, nobs, lean, obs_rows, sample_every, caps=None):
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    key = (dist, B, T, k, nobs)
    if key not in _ORACLE_RUNS:                       # (shared by the lean / accounting variants of one config)
        _ORACLE_RUNS[key] = bo.run_random_many(dist, k, range(1000, 1000 + B), range(B), T, True, nobs)
    want = _ORACLE_RUNS[key]
    env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps)
    env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
    if lean:
        env.accounting(False)
    obs = torch.zeros((B, obs_rows, env.cols), dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    env.rollout_device("random", T, True, torch.cuda.current_stream().cuda_stream, rew, done, rows, obs, obs_rows, False, True)
    env.sync(); torch.cuda.synchronize()
    st = env.stats()
    assert (st[:, 4] == 0).all()
    for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
        w = np.array([r[key] for r in want])
        assert np.array_equal(st[:, col], w), (key, int(np.flatnonzero(st[:, col] != w)[0]))
    if not lean:
        w = np.array([r["bytes"] for r in want])
        assert np.array_equal(st[:, 6], w), ("bytes", int(np.flatnonzero(st[:, 6] != w)[0]))
    assert np.array_equal(rows.cpu().numpy(), np.array([r["nP"] for r in want]))
    for e in sorted(set(range(0, B, sample_every)) | {B - 1}):
        basis, pairs, order = env.state(e)
        assert fnv64(_state_words(basis, pairs, order)) == want[e]["state_hash"], e
    return env, obs, want


def test_full_size_headline_lean_kernel_vs_oracle():
    """BASELINE configs[1], B=4096 x T=256, on bbx_fast_headline_kernel exactly as bench.py launches it (lean, counter-hash
    agent, observation every step without fill, auto-reset): all counters of all environments, the final state of every
    64th, and the observation block left behind."""
    bo = ffi.load("bo")
    env, obs, want = _full_size("3-20-10-weighted", 4096, 256, 2, 3, True, 256, 64)
    for e in (0, 777, 4095):
        o = bo.env("3-20-10-weighted"); o.seed(1000 + e); o.reset()
        for t in range(256):
            o.step(ffi.agent_action(e, t, o.nP))
            if o.nP == 0:
                o.reset()
        assert np.array_equal(obs[e, :o.nP].cpu().numpy(), o.obs(2)), e


@pytest.mark.parametrize("lean", [1, 0])
def test_full_size_5_10_5_uniform_vs_oracle(lean):
    """BASELINE configs[2]: 5-10-5-uniform, B=4096 x T=2048 (HBM-resident binomial class, 16-byte monomials): step /
    addition / episode / zero-reduction / basis-size counters of ALL environments (+ the algorithmic bytes on the
    accounting run), the complete final state of every 64th environment."""
    _full_size("5-10-5-uniform", 4096, 2048, 2, 5, bool(lean), 2048, 64)


@pytest.mark.parametrize("lean", [1, 0])
def test_full_size_cyclic7_vs_oracle(lean):
    """BASELINE configs[4]: cyclic-7, B=512 x T=128 on the wide kernel (lean: the accumulator variant that
    scripts/bench_configs.py times; accounting: the eagerly merged one whose byte counter feeds the roofline): counters of
    all environments, the complete final state of every 64th."""
    _full_size("cyclic-7", 512, 128, 2, 6, bool(lean), 1024, 64)


def test_host_step_outputs_of_a_finished_environment_too_large_for_the_register_class():
    """Found by scripts/fuzz_gym.py: bbx_step on a batch in which one environment has finished its episode (no auto-reset) with
    a basis beyond the register/LDS class's capacity.  The class's kernel handed it to the HBM-resident pass, which had nothing
    to do for it either — nobody wrote its outputs, and the call returned the PREVIOUS call's row count, reward and done flag for
    it.  An environment with nothing to do is no longer handed over."""
    from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
    B, k = 16, 2
    env = VecLeadMonomialsEnv("3-8-6-uniform", batch=B, k=k, caps={"lds_max_basis": 16})
    env.seed(np.arange(B) + 50965); env.reset()
    L = _ffi.lib()
    acts = np.zeros(B, dtype=np.int32)
    seen = 0
    for t in range(400):
        rew = np.full(B, 7.0); done = np.full(B, 9, dtype=np.uint8); rows = np.full(B, -5, dtype=np.int32)
        finished_before = env.rows == 0
        _ffi.check(L.bbx_step(env._h, _ffi.ptr(acts), _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(rows)))
        st = env.stats()
        big = finished_before & (st[:, 7] > 16)
        assert (rows[finished_before] == 0).all() and (done[finished_before] == 1).all() and (rew[finished_before] == 0.0).all(), (t, rows, done, rew)
        seen += int(big.sum())
        env.rows[:] = rows
        if finished_before.all():
            break
    assert seen > 0                                            # (the case occurred: a finished environment with more than 16 basis elements)
