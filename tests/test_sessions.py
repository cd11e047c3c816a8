"""Persistent sessions (bbx_persistent): asynchronous rollouts queued behind each other feed ONE running kernel through a
device-visible step counter.  Results must be those of the same calls as separate launches, i.e. the oracle's, bit for bit."""
import time

import numpy as np
import pytest

from oracle import ffi
from oracle.trace import fnv64
from tests.test_gpu_parity import _state_words

pytestmark = pytest.mark.gpu
DIST = "3-20-10-weighted"


def _env(B, caps=None, k=2):
    from deepgroebner_amd import VecLeadMonomialsEnv
    env = VecLeadMonomialsEnv(DIST, batch=B, k=k, caps=caps)
    env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
    env.persistent(True)
    return env


def _check(env, want, rows=None, obs=None, sample_every=37):
    st = env.stats()
    assert (st[:, 4] == 0).all(), st[:, 4]
    for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
        w = np.array([r[key] for r in want])
        assert np.array_equal(st[:, col], w), (key, int(np.flatnonzero(st[:, col] != w)[0]), st[:, col][st[:, col] != w][:4], w[st[:, col] != w][:4])
    B = len(want)
    if rows is not None:
        assert np.array_equal(rows.cpu().numpy(), np.array([r["nP"] for r in want]))
    for e in sorted(set(range(0, B, sample_every)) | {B - 1}):
        basis, pairs, order = env.state(e)
        assert fnv64(_state_words(basis, pairs, order)) == want[e]["state_hash"], e


@pytest.mark.parametrize("B,K,calls,caps", [
    (4096, 20, 30, None),                                  # the bench's shape: 4096 environments, 20 steps per call
    (512, 1, 300, None), (512, 7, 60, None),
    (300, 13, 40, {"lds_max_basis": 16}),                  # a register/LDS class so small that environments keep leaving it
    (64, 50, 12, {"lds_max_basis": 32, "max_basis": 12, "max_pairs": 16}),   # ... and the HBM records grow on top of that
])
def test_session_equals_separate_launches(B, K, calls, caps):
    import torch
    bo = ffi.load("bo")
    want = bo.run_random_many(DIST, 2, range(1000, 1000 + B), range(B), K * calls, True, 0)
    env = _env(B, caps)
    R = 256
    obs = torch.zeros((B, R, env.cols), dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(calls):
        env.rollout_device("random", K, True, s, rew, done, rows, obs, R, False, True)
    env.sync()
    ss = env.session_stats()
    assert ss["sessions"] >= 1 and ss["sessions"] + ss["joined"] == calls, ss
    _check(env, want, rows)
    for e in (0, B // 2, B - 1):                           # the block holds the observation of the last state
        o = bo.env(DIST); o.seed(1000 + e); o.reset()
        for t in range(K * calls):
            o.step(ffi.agent_action(e, t, o.nP))
            if o.nP == 0:
                o.reset()
        assert np.array_equal(obs[e, :o.nP].cpu().numpy(), o.obs(2)), e


def test_session_survives_an_idle_host_and_other_calls():
    """Waves leave after 20 ms without news and a kernel's time slice is 10 ms: steps issued after that are taken by the
    session's next kernel; stats() / state() / copy() in the middle end the session and the next rollout begins a new one."""
    import torch
    bo = ffi.load("bo")
    B, K = 256, 16
    env = _env(B)
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    total = 0
    for rnd in range(3):
        for i in range(10):
            env.rollout_device("random", K, True, s, rows=rows)
            total += K
            if i in (3, 7):
                time.sleep(0.06)                           # the kernel has left by now
        want = bo.run_random_many(DIST, 2, range(1000, 1000 + B), range(B), total, True, 0)
        if rnd == 0:
            st = env.stats()                               # (ends the session without bbx_sync)
            assert np.array_equal(st[:, 0], np.full(B, total))
        elif rnd == 1:
            twin = env.copy()
            _check(twin, want)
        _check(env, want, sample_every=17)
    assert env.session_stats()["sessions"] == 3


def test_session_join_orders_the_callers_stream():
    """bbx_join: the caller's stream (also the null stream) waits on the device for the session; what it then reads is final."""
    import torch
    bo = ffi.load("bo")
    B, K, calls = 512, 20, 25
    want = bo.run_random_many(DIST, 2, range(1000, 1000 + B), range(B), K * calls, True, 0)
    env = _env(B)
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream()
    for _ in range(calls):
        env.rollout_device("random", K, True, s.cuda_stream, rewards=rew, rows=rows)
    env.join(s.cuda_stream)
    snapshot = rows.clone()                                # queued on the caller's stream behind the join
    torch.cuda.synchronize()
    assert np.array_equal(snapshot.cpu().numpy(), np.array([r["nP"] for r in want]))
    env.sync()
    _check(env, want, rows)


def test_session_only_where_admitted():
    """Other kernel classes, traced or accounting launches, host-side calls: plain launches, same results."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    bo = ffi.load("bo")
    B, K, calls = 33, 10, 6
    for dist, acct in (("5-10-5-uniform", False), (DIST, True)):
        want = bo.run_random_many(dist, 2, range(1000, 1000 + B), range(B), K * calls, True, 0)
        env = VecLeadMonomialsEnv(dist, batch=B, k=2)
        env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset(); env.accounting(acct)
        env.persistent(True)
        rows = torch.zeros(B, dtype=torch.int32, device="cuda")
        for _ in range(calls):
            env.rollout_device("random", K, True, torch.cuda.current_stream().cuda_stream, rows=rows)
        env.sync()
        assert env.session_stats()["sessions"] == 0
        _check(env, want, rows, sample_every=5)


@pytest.mark.parametrize("caps", [None, {"lds_max_basis": 16}])
def test_policy_step_calls_join_a_session(caps):
    """bbx_policy_step_device under bbx_persistent: consecutive calls whose uniforms are consecutive [B] slices of one array
    are served by one running kernel with the policy inside its step loop.  Final outputs (actions, log-probabilities,
    rewards, dones, rows, the observation block with its -1 padding) and the environments equal those of the same calls
    made one kernel each."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(1)
    B, T, R = 96, 37, 256
    envs = []
    for persistent in (False, True):
        env = VecLeadMonomialsEnv(DIST, batch=B, k=2, caps=caps)
        env.seed(np.arange(B) + 1000); env.reset(); env.accounting(False)
        env.persistent(persistent)
        envs.append(env)
    policy = PMLPPolicy(envs[0].cols, [64]).cuda()
    w = policy._fused_weights()
    u = torch.rand((T, B), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for env in envs:
        obs = torch.zeros((B, R, env.cols), dtype=torch.int32, device="cuda")
        rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
        rows = torch.zeros(B, dtype=torch.int32, device="cuda"); act = torch.zeros(B, dtype=torch.int32, device="cuda")
        logp = torch.zeros(B, dtype=torch.float32, device="cuda")
        env.rollout_device("first", 0, False, s, rew, done, rows, obs, R, True, False)   # the block the first call's policy reads
        env.sync()
        for t in range(T):
            env.policy_step_device(w["prepared"], w["hidden"], u[t], act, logp, rew, done, rows, obs, R, 2, s)
        env.sync()
        outs.append([x.cpu().numpy().copy() for x in (act, logp, rew, done, rows, obs)] + [np.delete(env.stats(), 6, axis=1)])   # (without the
                                                                  # algorithmic-byte column: only the accounting variants keep it)
    assert envs[1].session_stats()["sessions"] == 1 and envs[1].session_stats()["joined"] == T - 1
    for a, b, name in zip(outs[0], outs[1], ("actions", "logprobs", "rewards", "dones", "rows", "obs", "stats")):
        assert np.array_equal(a, b), name


@pytest.mark.parametrize("B,auto_reset", [(1, False), (3, True), (8, True)])
def test_host_mailbox_session_is_invisible_except_in_time(B, auto_reset):
    """Host-driven steps of a small batch (the reference's usage: one environment stepped from Python, wrapped.pyx:23-26) on the
    register/LDS class: from the fifth step in a row on, the calls feed ONE resident kernel through pinned host memory
    (bbx_api.cpp mbox_step) — control word and actions written by the host, rewards / dones / rows / observation published by
    the waves with every step.  Every step's reward, done flag and observation matrix against the oracle; in between, calls
    that need the records (state, copy, value, stats) and that close the session; an out-of-range action surfaces as an error
    and the batch goes on; the statistics say that sessions were in fact used."""
    from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
    bo = ffi.load("bo")
    k = 2
    env = VecLeadMonomialsEnv(DIST, batch=B, k=k)
    env.seed(np.arange(B) + 77); env.accounting(False)
    states = env.reset()
    oracles = []
    for e in range(B):
        o = bo.env(DIST); o.seed(77 + e); o.reset(); oracles.append(o)
    copies = []
    for t in range(400):
        for e, o in enumerate(oracles):
            assert np.array_equal(states[e], o.obs(k)), (t, e)
        if not auto_reset and oracles[0].nP == 0:
            for o in oracles:
                o.reset()
            states = env.reset()
            continue
        acts = np.array([ffi.agent_hash(e + 1, t) % max(o.nP, 1) for e, o in enumerate(oracles)], dtype=np.int32)
        states, rew, done, _ = env.step(acts, auto_reset=auto_reset)
        for e, o in enumerate(oracles):
            assert rew[e] == o.step(int(acts[e])), (t, e)
            assert bool(done[e]) == (o.nP == 0), (t, e)
            if o.nP == 0 and auto_reset:
                o.reset()
        if t == 60:                                        # the records, read in the middle of a session
            basis, pairs, order = env.state(B - 1)
            assert np.array_equal(_state_words(basis, pairs, order), _state_words(oracles[B - 1].basis(), oracles[B - 1].pairs(), oracles[B - 1].reducer_order()))
        if t == 120:
            copies.append((env.copy(), [o.copy() for o in oracles]))
        if t == 200 and oracles[0].nP > 0:
            assert env.value(0, "degree", 0.99) == oracles[0].value("degree", 0.99)
        if t == 300:                                       # an action beyond the pair set is an error (the reference indexes out of
            bad = acts.copy(); bad[0] = 100000                 # bounds, buchberger.cpp:399); a reset clears it, like everywhere else
            with pytest.raises(_ffi.BbxError) as ei:
                env.step(bad, auto_reset=auto_reset)
            assert ei.value.code == -6
            for o in oracles:
                o.reset()
            states = env.reset()
    ss = env.session_stats()
    assert ss["sessions"] >= 2 and ss["joined"] > 200, ss
    cenv, coracles = copies[0]                             # a copy made in the middle of a session continues like its source did
    for t in range(30):
        acts = np.array([0] * B, dtype=np.int32)
        if any(o.nP == 0 for o in coracles) and not auto_reset:
            break
        st, rew, done, _ = cenv.step(acts, auto_reset=auto_reset)
        for e, o in enumerate(coracles):
            assert rew[e] == o.step(0)
            if o.nP == 0 and auto_reset:
                o.reset()
            assert np.array_equal(st[e], o.obs(k))


def test_sessions_of_different_shapes_back_to_back():
    """A call of another shape ends the running session and begins the next one at once.  The control word belongs to the
    handle: the kernels that close the first session (queued, perhaps not yet run) still poll it, and the next session's step
    total, written too early, was theirs to take — 6 of the second call's 7 steps under the FIRST call's agent (found by
    scripts/fuzz_sessions.py).  One step under the random agent, then seven under First, on a batch most of which outgrows a
    register/LDS class capped at 24 elements: counters and states against a twin without sessions, and against the oracle for
    the first call alone."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    B, k, R = 1000, 2, 512
    envs = []
    for persistent in (True, False):
        e = VecLeadMonomialsEnv("3-10-10-uniform", batch=B, k=k, caps={"lds_max_basis": 24})
        e.seed(np.arange(B)); e.seed_agent(np.arange(B)); e.reset(); e.accounting(False)
        if persistent:
            e.persistent(True)
        envs.append(e)
    s = torch.cuda.current_stream().cuda_stream
    for rep in range(6):                                          # (alternating shapes: every call ends a session and begins one)
        for e in envs:
            rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
            rows = torch.zeros(B, dtype=torch.int32, device="cuda"); obs = torch.full((B, R, e.cols), -1, dtype=torch.int32, device="cuda")
            e.rollout_device("random", 1, True, s, rew, done, rows, obs, R, False, False)
            e.rollout_device("first", 7, True, s, rew, done, rows, None, 0, False, False)
        for e in envs:
            e.sync()
        a, b = envs[0].stats(), envs[1].stats()
        assert np.array_equal(a[:, 0], b[:, 0]) and (a[:, 0] == 8 * (rep + 1)).all(), (rep, a[:3], b[:3])
        assert np.array_equal(a[:, :5], b[:, :5]) and np.array_equal(a[:, 7], b[:, 7]), rep
    for e_ in range(0, B, 97):
        assert fnv64(_state_words(*envs[0].state(e_))) == fnv64(_state_words(*envs[1].state(e_))), e_
    assert envs[0].session_stats()["sessions"] >= 12


def test_policy_step_session_leaves_the_new_episodes_block():
    """bbx_policy_step_device with auto-reset: the observation block and row counts the call leaves describe the NEW episode of an
    environment whose episode the step ended.  Served by a session, the step wrote the block of the state it left (no rows) and
    the reset at the top of the next loop iteration wrote nothing — within the session nobody reads the block, but the caller
    does when the session ends (found by scripts/fuzz_sessions.py; both the register/LDS class and the HBM-resident continuation
    of a class capped at 16 basis elements).  After every call: the block against a fresh observation of the state."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy
    B, R, k = 64, 512, 2
    env = VecLeadMonomialsEnv("3-3-7-uniform", batch=B, k=k, caps={"lds_max_basis": 16})
    env.seed(np.arange(B) + 51); env.seed_agent(np.arange(B) + 3); env.reset(); env.accounting(False)
    env.persistent(True)
    torch.manual_seed(5003)
    policy = PMLPPolicy(env.cols, [64]).cuda()
    pw = policy._fused_weights()
    s = torch.cuda.current_stream().cuda_stream
    def bufs():
        return (torch.zeros(B, dtype=torch.float64, device="cuda"), torch.zeros(B, dtype=torch.uint8, device="cuda"),
                torch.zeros(B, dtype=torch.int32, device="cuda"), torch.full((B, R, env.cols), -1, dtype=torch.int32, device="cuda"))
    rew, done, rows, obs = bufs()
    rew2, done2, rows2, obs2 = bufs()
    act = torch.zeros(B, dtype=torch.int32, device="cuda"); logp = torch.zeros(B, dtype=torch.float32, device="cuda")
    U = torch.rand((120, B), device="cuda")
    env.rollout_device("first", 0, False, s, rew, done, rows, obs, R, True, False); env.sync()
    ended = 0
    for t in range(120):
        env.policy_step_device(pw["prepared"], pw["hidden"], U[t], act, logp, rew, done, rows, obs, R, 2, s)
        env.sync()
        ended += int(done.sum())
        env.rollout_device("first", 0, False, s, rew2, done2, rows2, obs2, R, True, False); env.sync()
        assert torch.equal(rows, rows2), t
        live = torch.arange(R, device="cuda")[None, :] < rows2[:, None]
        assert torch.equal(obs[live], obs2[live]), t
    assert ended > 50                                           # (episodes did end inside the calls)
