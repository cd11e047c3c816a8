"""The consumer side of the batched environment (deepgroebner_amd/rollout.py): trajectory-buffer
arithmetic against the reference's formulas and known answers (CPU), the fused policy kernel against the torch module,
and a device rollout with the policy in the loop replayed on the oracle (GPU)."""
import numpy as np
import pytest


# ---- CPU: pg.py:20-76 known answers (tests/test_pg.py:9-40 of the reference) and the batched buffer -----------------------
def test_discount_and_advantage_known_answers():
    from deepgroebner_amd.rollout import compute_advantages, discount_rewards
    assert np.array_equal(discount_rewards([], 0.9), [])
    assert np.array_equal(discount_rewards([1, 2, 3], 1), [6, 5, 3])
    assert np.array_equal(discount_rewards([1, 1, 1, 1], 0.9), [3.439, 2.71, 1.9, 1.])
    assert np.array_equal(discount_rewards(np.array([1., 1., 1., 1.]), 0.9), np.array([3.439, 2.71, 1.9, 1.]))
    L = [1, 2, 3, 4, 5]
    L[2:] = discount_rewards(L[2:], 0.5)
    assert L == [1, 2, 6.25, 6.5, 5]
    for gam, lam, want in ((1.0, 1.0, [5., 4., 3., 2., 1.]), (0.5, 1.0, [1.9375, 1.875, 1.75, 1.5, 1.]),
                           (1.0, 0.5, [1.9375, 1.875, 1.75, 1.5, 1.]), (0.5, 0.5, [1.33203125, 1.328125, 1.3125, 1.25, 1.])):
        assert np.array_equal(compute_advantages([1, 1, 1, 1, 1], [0, 0, 0, 0, 0], gam, lam), want)


def test_batched_buffer_equals_per_episode_reference_formulas():
    import torch
    from deepgroebner_amd.rollout import DeviceTrajectoryBuffer, compute_advantages, discount_rewards
    rng = np.random.default_rng(3)
    T, B, gam, lam = 60, 7, 0.97, 0.9
    buf = DeviceTrajectoryBuffer(T, B, gam, lam, obs_shape=None, device="cpu")
    rew = -rng.integers(1, 6, size=(T, B)).astype(np.float64)
    val = rng.normal(size=(T, B))
    done = rng.random((T, B)) < 0.12
    rows = rng.integers(1, 5, size=(T, B)).astype(np.int32)
    for t in range(T):
        buf.store(None, torch.tensor(rows[t]), torch.zeros(B, dtype=torch.int32), torch.tensor(rew[t]), torch.zeros(B), torch.tensor(val[t]),
                  torch.tensor(done[t]))
    ret, adv, comp = (x.numpy() for x in buf.finish())
    for b in range(B):
        start = 0
        for t in range(T):
            if done[t, b]:
                tau = slice(start, t + 1)
                assert np.allclose(ret[tau, b], discount_rewards(rew[tau, b], gam), rtol=0, atol=1e-12)
                assert np.allclose(adv[tau, b], compute_advantages(rew[tau, b], val[tau, b], gam, lam), rtol=0, atol=1e-12)
                assert comp[tau, b].all()
                start = t + 1
        assert not comp[start:, b].any()                        # the unfinished tail is left out
    _, a, lp, ad, vf = buf.get(normalize_advantages=True)
    keep = comp & (rows != 1)
    all_adv = adv.T[comp.T]                                        # trajectory after trajectory (environment-major)
    want = ((all_adv - all_adv.mean()) / all_adv.std())[(rows != 1).T[comp.T]]
    assert len(a) == keep.sum() and np.allclose(ad.numpy(), want.astype(np.float32), atol=1e-5)
    assert np.allclose(vf.numpy(), ret.T[keep.T].astype(np.float32))


def test_buffer_get_batches_like_the_reference_padded_batch():
    """TrajectoryBuffer.get(batch_size, sort, drop_remainder) (pg.py:162-240): batches of at most batch_size steps, the
    state block of a batch as tall as its tallest state, -1 below every state's own rows, optional sort by rows."""
    import torch
    from deepgroebner_amd.rollout import DeviceTrajectoryBuffer
    rng = np.random.default_rng(5)
    T, B, R, cols = 40, 5, 9, 4
    buf = DeviceTrajectoryBuffer(T, B, obs_shape=(R, cols), device="cpu")
    rows = rng.integers(1, R + 1, size=(T, B)).astype(np.int32)
    done = rng.random((T, B)) < 0.15
    for t in range(T):
        state = np.full((B, R, cols), -1, dtype=np.int32)
        for b in range(B):
            state[b, :rows[t, b]] = 100 * t + b
        buf.store(torch.tensor(state), torch.tensor(rows[t]), torch.tensor(np.arange(B, dtype=np.int32) + t), torch.tensor(-np.ones(B)),
                  torch.zeros(B), torch.zeros(B, dtype=torch.float64), torch.tensor(done[t]))
    whole = buf.get()
    n = len(whole[1])
    assert n > 20
    for sort in (False, True):
        for drop in (False, True):
            batches = buf.get(batch_size=8, sort=sort, drop_remainder=drop)
            assert len(batches) == (n // 8 if drop else -(-n // 8))
            seen = []
            for st, a, lp, ad, vf in batches:
                assert 1 <= len(a) <= 8 and (not drop or len(a) == 8)
                r = (st[:, :, -1] != -1).sum(dim=1)
                assert st.shape[1] == int(r.max()) and (r > 1).all()          # as tall as the tallest state, single rows filtered
                for i in range(len(a)):
                    assert (st[i, :r[i]] == st[i, 0, 0]).all() and (st[i, r[i]:] == -1).all()
                seen += r.tolist()
            if sort:
                assert seen == sorted(seen)
            elif not drop:
                assert torch.equal(torch.cat([b[1] for b in batches]), whole[1])


def test_pmlp_policy_masks_padding_like_the_reference():
    """networks.py:543-560 docstring example: padded rows get (numerically) zero probability, each row of the output a
    distribution over the valid rows."""
    import torch
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(0)
    pmlp = PMLPPolicy(2, [128])
    states = torch.tensor([[[0, 1], [3, 0], [-1, -1]], [[8, 5], [3, 3], [3, 5]], [[6, 7], [6, 8], [-1, -1]]], dtype=torch.int32)
    lp = pmlp(states)
    assert lp.shape == (3, 3)
    p = lp.exp()
    assert torch.allclose(p.sum(dim=1), torch.ones(3), atol=1e-5)
    assert p[0, 2] < 1e-30 and p[2, 2] < 1e-30 and (p[1] > 0).all()


# ---- GPU ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dist,k,hidden", [("3-20-10-weighted", 2, 128), ("5-10-5-uniform", 1, 64), ("3-20-10-uniform", 3, 200),
                                           ("4-5-4-uniform", 1, 32), ("3-20-10-uniform", 1, 7), ("5-10-5-uniform", 3, 256)])
def test_fused_policy_kernel_matches_torch_module(dist, k, hidden):
    """bbx_pmlp_act (exact-f32 MFMA tiles over prepared weights) against the torch module: odd column counts, hidden sizes
    that do not fill a tile, several unit-block groups and k-step counts."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(1)
    B, R = 300, 192
    env = VecLeadMonomialsEnv(dist, batch=B, k=k)
    env.seed(np.arange(B) + 5); env.seed_agent(np.arange(B)); env.reset()
    env.rollout("random", 25, auto_reset=True)
    obs = torch.full((B, R, env.cols), -7, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    env.rollout_device("first", 0, False, torch.cuda.current_stream().cuda_stream, rew, done, rows, obs, R, True, False)
    env.sync()
    assert int(rows.max()) <= R and int(rows.min()) >= 1
    policy = PMLPPolicy(env.cols, [hidden]).cuda()
    with torch.no_grad():
        for lin in list(policy.embedding) + [policy.deciding]:
            lin.weight.mul_(0.3)                                  # keep the logits in a range where several rows matter
    for trial in range(3):
        u = torch.rand(B, device="cuda")
        a_k, l_k = policy.act(obs, rows, u)
        a_t, l_t = policy.act_torch(obs, rows, u)
        torch.cuda.synchronize()
        assert (a_k >= 0).all() and (a_k < rows).all()
        lp = policy(obs)
        assert torch.allclose(l_k, lp.gather(1, a_k.long()[:, None]).squeeze(1), atol=2e-4, rtol=1e-4)
        same = (a_k == a_t)
        assert same.float().mean() > 0.99
        # a differing draw is a round-off tie: u sits on the boundary between the two rows
        if not same.all():
            p = lp.exp()
            cdf = torch.cumsum(p, dim=1)
            for e in torch.nonzero(~same).flatten().tolist():
                lo, hi = sorted((int(a_k[e]), int(a_t[e])))
                assert hi - lo == 1 and abs(float(cdf[e, lo]) - float(u[e])) < 1e-4, e


@pytest.mark.gpu
@pytest.mark.parametrize("dist,k,hidden", [("3-20-10-weighted", 2, (128, 128)), ("5-10-5-uniform", 1, (64, 128)), ("3-20-10-uniform", 3, (100, 40)),
                                           ("4-5-4-uniform", 1, (32, 7)), ("5-10-5-uniform", 3, (128, 64)), ("6-3-4-uniform", 3, (128, 128)),
                                           ("3-20-10-weighted", 2, (128, 128, 128)), ("5-10-5-uniform", 3, (40, 100, 17)),
                                           ("4-5-4-uniform", 1, (64, 33, 64)), ("6-3-4-uniform", 3, (128, 96, 128))])
def test_fused_two_layer_policy_kernel_matches_torch_module(dist, k, hidden):
    """bbx_pmlp2_act / bbx_pmlp3_act (two / three hidden layers on the matrix cores, each layer's k-steps in the order the
    previous layer's accumulators lie) against the torch module: every padded-size combination, all three k-step counts,
    layer sizes that do not fill a tile, environments with more than two tiles of rows (shared among the workgroup)."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(3)
    B, R = 300, 192
    env = VecLeadMonomialsEnv(dist, batch=B, k=k)
    env.seed(np.arange(B) + 5); env.seed_agent(np.arange(B)); env.reset()
    env.rollout("random", 25, auto_reset=True)
    obs = torch.full((B, R, env.cols), -7, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    env.rollout_device("first", 0, False, torch.cuda.current_stream().cuda_stream, rew, done, rows, obs, R, True, False)
    env.sync()
    assert int(rows.max()) <= R and int(rows.min()) >= 1
    policy = PMLPPolicy(env.cols, list(hidden)).cuda()
    assert policy.deep_ok(env.cols)
    with torch.no_grad():
        for lin in list(policy.embedding) + [policy.deciding]:
            lin.weight.mul_(0.3)
    for trial in range(3):
        u = torch.rand(B, device="cuda")
        a_k, l_k = policy.act(obs, rows, u)
        a_t, l_t = policy.act_torch(obs, rows, u)
        torch.cuda.synchronize()
        assert (a_k >= 0).all() and (a_k < rows).all()
        lp = policy(obs)
        assert torch.allclose(l_k, lp.gather(1, a_k.long()[:, None]).squeeze(1), atol=3e-4, rtol=1e-4)
        same = (a_k == a_t)
        assert same.float().mean() > 0.99
        if not same.all():
            cdf = torch.cumsum(lp.exp(), dim=1)
            for e in torch.nonzero(~same).flatten().tolist():
                lo, hi = sorted((int(a_k[e]), int(a_t[e])))
                assert hi - lo == 1 and abs(float(cdf[e, lo]) - float(u[e])) < 1e-4, e
        if trial == 0:                                            # an optimiser step: the prepared copy follows, in the same buffer
            before = policy._deep_weights()["prepared"].value
            with torch.no_grad():
                policy.embedding[1].weight.add_(0.01)
            assert policy._deep_weights()["prepared"].value == before


@pytest.mark.gpu
@pytest.mark.parametrize("R", [448, 512, 1024])
@pytest.mark.parametrize("hidden", [(128, 128), (128, 128, 128)])
def test_deep_policy_kernels_on_tall_observation_blocks(hidden, R):
    """The logits of a wave's environment live in LDS next to the staged weights: with 128 KB of weights (three layers of 128
    units) a tall block (up to 1024 rows = 4 KB of logits per wave) leaves room for fewer waves per workgroup — the launcher
    picks 16, 8 or 4 (found by scripts/fuzz_policy.py: the 16-wave launch was refused by the runtime)."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(6)
    B = 48
    env = VecLeadMonomialsEnv("5-10-5-uniform", batch=B, k=1)
    env.seed(np.arange(B) + 9); env.seed_agent(np.arange(B)); env.reset()
    env.rollout("random", 300, auto_reset=True)
    obs = torch.full((B, R, env.cols), -1, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda")
    env.rollout_device("first", 0, False, torch.cuda.current_stream().cuda_stream, rew, done, rows, obs, R, True, False)
    try:
        env.sync()
    except Exception:                                              # (rows beyond the block are an error of the block, not of this test)
        pass
    n = torch.clamp(rows, min=1, max=R)
    assert int(n.max()) > 64                                       # several tiles per environment: the workgroup's shared queue is in use
    policy = PMLPPolicy(env.cols, list(hidden)).cuda()
    with torch.no_grad():
        for lin in list(policy.embedding) + [policy.deciding]:
            lin.weight.mul_(0.2)
    u = torch.rand(B, device="cuda")
    a_k, l_k = policy.act(obs, n, u)
    torch.cuda.synchronize()
    assert (a_k >= 0).all() and (a_k < n).all()
    lp = policy(obs)
    valid = torch.arange(R, device="cuda")[None, :] < n[:, None]
    lpm = torch.log_softmax(torch.where(valid, lp, torch.full_like(lp, -1e30)), dim=1)
    assert torch.allclose(l_k, lpm.gather(1, a_k.long()[:, None]).squeeze(1), atol=5e-4, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [(128,), (64,), (128, 128), (64, 64, 64), (128, 128, 128)])
def test_policy_kernels_score_up_to_2048_rows(hidden):
    """5-10-5-uniform pair sets pass a thousand rows: the policy kernels keep an environment's logits in LDS, 2048 of them
    (1024 beside the 128 KB of weights of three wide layers: deep_max_rows, the torch path takes the rest).  Synthetic blocks
    with 1 .. R live rows per environment: draws and log-probabilities against the torch module."""
    import torch
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(11)
    B, cols = 40, 20
    policy = PMLPPolicy(cols, list(hidden)).cuda()
    with torch.no_grad():
        for lin in list(policy.embedding) + [policy.deciding]:
            lin.weight.mul_(0.3)
    for R in (1500, 2048):
        obs = torch.randint(0, 9, (B, R, cols), dtype=torch.int32, device="cuda")
        rows = torch.randint(1, R + 1, (B,), dtype=torch.int32, device="cuda")
        rows[0] = R; rows[1] = 1; rows[2] = 1025
        obs[torch.arange(R, device="cuda")[None, :] >= rows[:, None]] = -1
        u = torch.rand(B, device="cuda")
        a_k, l_k = policy.act(obs, rows, u)
        a_t, l_t = policy.act_torch(obs, rows, u)
        torch.cuda.synchronize()
        assert (a_k >= 0).all() and (a_k < rows).all()
        lp = policy(obs)
        assert torch.allclose(l_k, lp.gather(1, a_k.long()[:, None]).squeeze(1), atol=5e-4, rtol=1e-4)
        assert int((a_k != a_t).sum()) <= 2                            # (equal except on round-off ties of the cumulative sum)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [(64, 64), (32, 32, 32), (32, 32, 32, 32)])
def test_rollout_replayed_from_a_hip_graph_equals_the_eager_rollout(hidden):
    """run_rollout(graph=True): the vector step (policy ops + bbx_step_device_autoreset) recorded once and replayed gives the
    rollout of the same calls made one by one — two and three hidden layers (bbx_pmlp2_act / bbx_pmlp3_act in the graph)
    and four (torch ops in the graph); a second call reuses the recording; bbx_graph_replayed makes bbx_sync see the replayed work."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy, run_rollout
    torch.manual_seed(4)
    B, T = 96, 150
    policy = None
    res = {}
    for graph in (False, True):
        env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
        env.seed(np.arange(B) + 77); env.reset()
        if policy is None:
            policy = PMLPPolicy(env.cols, list(hidden)).cuda()
        g = torch.Generator(device="cuda"); g.manual_seed(11)
        tot1, ep1 = run_rollout(env, policy, T, obs_rows=128, generator=g, sync_every=40, graph=graph)
        tot2, ep2 = run_rollout(env, policy, 30, obs_rows=128, generator=g, sync_every=40, graph=graph)
        torch.cuda.synchronize()
        res[graph] = (tot1.cpu().numpy(), ep1.cpu().numpy(), tot2.cpu().numpy(), ep2.cpu().numpy(), env.stats().copy())
        if graph:
            assert len(env._step_graphs) == 1
    for a, b in zip(res[False], res[True]):
        assert np.array_equal(a, b)
    assert res[True][1].sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_records_outgrown_inside_a_chain_of_device_steps_are_reported(graph):
    """Asynchronous steps with caller-supplied actions queued behind each other (run_rollout between two waits, eager or
    replayed from a graph): an environment that outgrows its records stops at that step and sits out the rest of the chain;
    the action buffer then holds a later step's actions, so the library cannot take the missed step for it — bbx_sync
    enlarges the records and says so (BBX_E_CAPACITY, never a wrong or out-of-range action taken silently).  Under a
    recorded graph the enlarged records live at a new address, which the recording does not know: run_rollout drops it
    with the very call that reports the growth (one growth event costs ONE failed rollout: a recording kept beyond it would
    step the retired copy and fail again as 'stale' — 'record the step again' — at the next call).  Afterwards the batch is
    intact and steps normally."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd._ffi import BbxError
    from deepgroebner_amd.rollout import PMLPPolicy, run_rollout
    torch.manual_seed(5)
    B = 64
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2, caps={"max_basis": 16, "max_pairs": 32})
    env.seed(np.arange(B) + 300); env.reset()
    policy = PMLPPolicy(env.cols, [32, 32, 32, 32]).cuda()            # (torch ops: the path whose steps a graph is worth recording)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    kinds = []
    clean = 0
    for attempt in range(40):
        before = env.stats()[:, 0].copy()
        try:
            run_rollout(env, policy, 8, obs_rows=128, generator=g, sync_every=8, graph=graph)
        except BbxError as e:
            assert e.code == -3, str(e)                              # BBX_E_CAPACITY
            kinds.append("stale" if "record the step again" in str(e) else "chain" if "chain of asynchronous steps" in str(e) else str(e))
            assert kinds[-1] in ("stale", "chain"), kinds[-1]
            if graph:
                assert not env._step_graphs                          # the recording went with the error: the next call records again
            continue
        assert ((env.stats()[:, 0] - before) == 8).all()             # a call without an error: every environment took its 8 steps
        clean += 1
        if clean >= 3 and env.capacities()["grown"] >= 1:
            break
    assert "chain" in kinds and clean >= 3 and env.capacities()["grown"] >= 1
    assert "stale" not in kinds
    assert (env.stats()[:, 4] == 0).all()


@pytest.mark.gpu
def test_launches_that_need_the_host_refuse_graph_capture():
    """A launch that could not be replayed faithfully (here: a handle with persistent sessions enabled, whose calls talk to a
    running kernel through the host) is refused while the stream is capturing — BBX_E_UNSUPPORTED — and the handle works
    normally afterwards."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd._ffi import BbxError
    B = 8
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=1)
    env.seed(np.arange(B)); env.reset()
    env.persistent(True)
    act = torch.zeros(B, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    refused = False
    with torch.cuda.graph(g, stream=side):
        try:
            env.step_device(act, stream=torch.cuda.current_stream().cuda_stream)
        except BbxError as e:
            refused = e.code == -5 and "graph" in str(e)
        act.add_(0)                                                 # (something for the recording to hold)
    assert refused
    env.persistent(False)
    env.step_device(act, stream=torch.cuda.current_stream().cuda_stream)
    env.sync()
    assert int(env.stats()[:, 0].sum()) == B


@pytest.mark.gpu
def test_device_rollout_with_policy_in_the_loop_replays_on_the_oracle():
    """Actions sampled on the device index the rows the oracle steps: a rollout with the PMLP policy in the loop (no host
    round trip per step), its recorded states / actions / rewards / dones replayed environment by environment on the CPU
    oracle."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import DeviceTrajectoryBuffer, PMLPPolicy, run_rollout
    from oracle import ffi
    bo = ffi.load("bo")
    torch.manual_seed(2)
    B, T, k, R = 48, 150, 2, 128
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=k)
    env.seed(np.arange(B) + 700); env.reset()
    policy = PMLPPolicy(env.cols, [128]).cuda()
    buf = DeviceTrajectoryBuffer(T, B, 0.99, 0.97, obs_shape=(R, env.cols))
    total, episodes = run_rollout(env, policy, T, buffer=buf, obs_rows=R, sync_every=16)
    torch.cuda.synchronize()
    states = buf.states.cpu().numpy(); acts = buf.actions.cpu().numpy(); rews = buf.rewards.cpu().numpy()
    dones = buf.dones.cpu().numpy(); rows = buf.rows.cpu().numpy()
    for e in range(B):
        o = bo.env("3-20-10-weighted"); o.seed(700 + e); o.reset()
        tot, eps = 0.0, 0
        for t in range(T):
            want = o.obs(k)
            assert rows[t, e] == o.nP and np.array_equal(states[t, e, :o.nP], want) and (states[t, e, o.nP:] == -1).all(), (e, t)
            assert 0 <= acts[t, e] < o.nP
            r = o.step(int(acts[t, e]))
            tot += r
            assert rews[t, e] == r and bool(dones[t, e]) == (o.nP == 0), (e, t)
            if o.nP == 0:
                eps += 1
                o.reset()
        assert float(total[e]) == tot and int(episodes[e]) == eps
    ret, adv, comp = buf.finish()
    assert bool(comp.any()) and torch.isfinite(ret).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dist,k,hidden,caps", [("3-20-10-weighted", 2, 128, None), ("3-20-10-weighted", 2, 48, None),
                                                ("3-20-10-weighted", 2, 128, {"lds_max_basis": 16}), ("3-20-10-weighted", 2, 64, {"lds_max_basis": 16}),
                                                ("5-10-5-uniform", 2, 128, None), ("5-10-5-uniform", 1, 64, None), ("2-8-6-uniform", 2, 40, None),
                                                ("4-5-4-uniform", 2, 128, None)])
def test_policy_rollout_in_one_launch_equals_the_per_step_loop(dist, k, hidden, caps):
    """bbx_policy_rollout_device (policy inside the step kernel, T steps per launch) against T calls of
    bbx_policy_step_device on a copy of the same batch with the same uniform numbers: actions, log-probabilities, rewards,
    dones, row counts and the observation of every step are identical (the logits come from the same tile code in the
    same summation order), and so are the environments' counters afterwards.  With the register/LDS class capped at 16
    basis elements most environments outgrow it inside the rollout: the HBM-resident continuation pass (policy included)
    takes them over mid-launch.  Other rings and observation widths run in the HBM-resident binomial kernel from the start."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(3)
    B, T, R = (500, 70, 256) if dist.startswith("3-") else (200, 60, 1024)
    env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps)
    env.seed(np.arange(B) + 77); env.reset(); env.accounting(False)
    twin = env.copy(); twin.accounting(False)
    policy = PMLPPolicy(env.cols, [hidden]).cuda()
    with torch.no_grad():
        for lin in list(policy.embedding) + [policy.deciding]:
            lin.weight.mul_(0.3)
    w = policy._fused_weights()
    s = torch.cuda.current_stream().cuda_stream
    u = torch.rand((T, B), device="cuda")
    # reference: one call per step
    obs = torch.full((B, R, env.cols), -1, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda"); act = torch.zeros(B, dtype=torch.int32, device="cuda")
    logp = torch.zeros(B, dtype=torch.float32, device="cuda")
    env.rollout_device("first", 0, False, s, rew, done, rows, obs, R, True, False); env.sync()
    want = {k: [] for k in ("obs", "rows", "act", "logp", "rew", "done")}
    for t in range(T):
        want["obs"].append(obs.clone()); want["rows"].append(rows.clone())
        env.policy_step_device(w["prepared"], w["hidden"], u[t], act, logp, rew, done, rows, obs, R, 1, s)
        env.sync()
        for k, v in (("act", act), ("logp", logp), ("rew", rew), ("done", done)):
            want[k].append(v.clone())
    # one launch (two, to cross a launch boundary)
    A = torch.zeros((T, B), dtype=torch.int32, device="cuda"); L = torch.zeros((T, B), dtype=torch.float32, device="cuda")
    Rw = torch.zeros((T, B), dtype=torch.float64, device="cuda"); D = torch.zeros((T, B), dtype=torch.uint8, device="cuda")
    N = torch.zeros((T, B), dtype=torch.int32, device="cuda")
    O = torch.full((T, B, R, env.cols), -1, dtype=torch.int32, device="cuda")
    cut = 29
    twin.policy_rollout_device(w["prepared"], w["hidden"], cut, u[:cut], A[:cut], L[:cut], Rw[:cut], D[:cut], N[:cut], O[:cut], R, B * R * env.cols, s)
    twin.sync()
    twin.policy_rollout_device(w["prepared"], w["hidden"], T - cut, u[cut:], A[cut:], L[cut:], Rw[cut:], D[cut:], N[cut:], O[cut:], R, B * R * env.cols, s)
    twin.sync()
    for t in range(T):
        assert torch.equal(N[t], want["rows"][t]), t
        assert torch.equal(A[t], want["act"][t]), t
        assert torch.equal(L[t], want["logp"][t]), t
        assert torch.equal(Rw[t], want["rew"][t]) and torch.equal(D[t], want["done"][t]), t
        live = torch.arange(R, device="cuda")[None, :] < N[t][:, None]
        assert torch.equal(O[t][live], want["obs"][t][live]), t
        assert (O[t][~live] == -1).all()
    # steps, additions, episodes, zero reductions, status (the algorithmic-byte column is only kept by the HBM-resident
    # kernel when accounting is off, and the two drivers hand environments over to it at different moments)
    assert np.array_equal(env.stats()[:, :5], twin.stats()[:, :5])
    if dist.startswith("3-"):
        assert D.sum() > 0                                  # (episodes of these ideals end and restart inside the rollout)


@pytest.mark.gpu
def test_policy_in_the_loop_on_pair_sets_of_more_than_1024_rows():
    """Pair sets of five-variable binomial ideals pass a thousand rows (5-10-5-uniform: ~1000 at most in a batch of 4096;
    5-15-5-uniform: 1200 within 600 steps).  The policy kernels score up to 2048 (the logits of an environment live in LDS: the
    HBM-resident binomial class sizes its per-wave scratch for them): a batch pre-rolled until some pair set has more than
    1100 rows, then the policy rollout kernel against one bbx_policy_step_device call per step,
    as in test_policy_rollout_in_one_launch_equals_the_per_step_loop, and every draw against the torch module."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.rollout import PMLPPolicy
    torch.manual_seed(5)
    B, T, R, k = 256, 8, 2048, 1
    env = VecLeadMonomialsEnv("5-15-5-uniform", batch=B, k=k)
    env.seed(np.arange(B) + 300); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
    for _ in range(30):                                             # (a property of the seeds: checked, not assumed)
        env.rollout("random", 100, auto_reset=True)
        if int(env.rows.max()) > 1100:
            break
    assert 1100 < int(env.rows.max()) <= R
    twin = env.copy(); twin.accounting(False)
    policy = PMLPPolicy(env.cols, [64]).cuda()
    with torch.no_grad():
        for lin in list(policy.embedding) + [policy.deciding]:
            lin.weight.mul_(0.3)
    w = policy._fused_weights()
    s = torch.cuda.current_stream().cuda_stream
    u = torch.rand((T, B), device="cuda")
    obs = torch.full((B, R, env.cols), -1, dtype=torch.int32, device="cuda")
    rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rows = torch.zeros(B, dtype=torch.int32, device="cuda"); act = torch.zeros(B, dtype=torch.int32, device="cuda")
    logp = torch.zeros(B, dtype=torch.float32, device="cuda")
    env.rollout_device("first", 0, False, s, rew, done, rows, obs, R, True, False); env.sync()
    want = {k_: [] for k_ in ("rows", "act", "logp", "rew", "done")}
    big = 0
    for t in range(T):
        want["rows"].append(rows.clone())
        big = max(big, int(rows.max()))
        a_t, l_t = policy.act_torch(obs, rows, u[t])
        env.policy_step_device(w["prepared"], w["hidden"], u[t], act, logp, rew, done, rows, obs, R, 1, s)
        env.sync()
        assert int((act != a_t).sum()) <= 1 and torch.allclose(logp[act == a_t], l_t[act == a_t], atol=5e-4, rtol=1e-4), t
        for k_, v in (("act", act), ("logp", logp), ("rew", rew), ("done", done)):
            want[k_].append(v.clone())
    assert big > 1024
    A = torch.zeros((T, B), dtype=torch.int32, device="cuda"); L = torch.zeros((T, B), dtype=torch.float32, device="cuda")
    Rw = torch.zeros((T, B), dtype=torch.float64, device="cuda"); D = torch.zeros((T, B), dtype=torch.uint8, device="cuda")
    N = torch.zeros((T, B), dtype=torch.int32, device="cuda")
    twin.policy_rollout_device(w["prepared"], w["hidden"], T, u, A, L, Rw, D, N, None, R, 0, s)
    twin.sync()
    for t in range(T):
        assert torch.equal(N[t], want["rows"][t]), t
        assert torch.equal(A[t], want["act"][t]) and torch.equal(L[t], want["logp"][t]), t
        assert torch.equal(Rw[t], want["rew"][t]) and torch.equal(D[t], want["done"][t]), t
    assert np.array_equal(env.stats()[:, :5], twin.stats()[:, :5])


@pytest.mark.gpu
def test_policy_rollout_rejects_what_the_kernel_class_cannot_do():
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
    from deepgroebner_amd.rollout import PMLPPolicy
    env = VecLeadMonomialsEnv("cyclic-4", batch=8, k=2)        # fixed ideals: not a binomial kernel class
    env.reset(); env.accounting(False)
    policy = PMLPPolicy(env.cols, [64]).cuda()
    w = policy._fused_weights()
    z = torch.zeros((4, 8), device="cuda")
    with pytest.raises(_ffi.BbxError) as ei:
        env.policy_rollout_device(w["prepared"], w["hidden"], 4, z, z.int(), z.clone(), stream=torch.cuda.current_stream().cuda_stream)
    assert ei.value.code == -5                               # BBX_E_UNSUPPORTED (include/bbx.h)


@pytest.mark.gpu
def test_rollout_refuses_to_overrun_the_buffer_and_policy_row_limit():
    """run_rollout_fused hands raw pointers into the trajectory buffer to the kernel: a rollout that does not fit raises
    instead of writing past the arrays; observation blocks taller than the 2048 rows the policy kernels score are refused."""
    import torch
    from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
    from deepgroebner_amd.rollout import DeviceTrajectoryBuffer, PMLPPolicy, run_rollout_fused
    B = 32
    env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
    env.seed(np.arange(B) + 7); env.reset(); env.accounting(False)
    policy = PMLPPolicy(env.cols, [64]).cuda()
    buf = DeviceTrajectoryBuffer(40, B)
    run_rollout_fused(env, policy, 30, buffer=buf, chunk=16)
    assert buf.t == 30
    with pytest.raises(IndexError):
        run_rollout_fused(env, policy, 11, buffer=buf, chunk=16)
    run_rollout_fused(env, policy, 10, buffer=buf, chunk=16)
    assert buf.t == 40
    obs = torch.zeros((B, 2100, env.cols), dtype=torch.int32, device="cuda")
    w = policy._fused_weights()
    u = torch.rand((4, B), device="cuda"); act = torch.zeros((4, B), dtype=torch.int32, device="cuda"); lp = torch.zeros((4, B), device="cuda")
    with pytest.raises(_ffi.BbxError) as ei:
        env.policy_rollout_device(w["prepared"], w["hidden"], 4, u, act, lp, obs=obs, obs_rows=2100)
    assert ei.value.code == -5
    rows = torch.full((B,), 3, dtype=torch.int32, device="cuda")
    a2, l2 = policy.act(obs, rows, u[0])                            # (taller than the kernels score: the torch path serves it)
    a3, l3 = policy.act_torch(obs, rows, u[0])
    assert torch.equal(a2, a3) and torch.allclose(l2, l3)
