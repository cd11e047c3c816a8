"""Randomised self-consistency sweep of the policy paths (test infrastructure, GPU box): for random binomial distributions,
observation widths, hidden sizes, batch sizes and kernel capacities, bbx_policy_rollout_device (policy inside the step
kernels, T steps per launch, split at a random point) must reproduce T calls of bbx_policy_step_device on a copy of the batch
— actions, log-probabilities, rewards, dones, row counts, observations — and the torch module must agree with the sampled
log-probabilities; the two- / three-layer policy kernels (random layer sizes) must agree with the torch module on the
rollout's final block.     python scripts/fuzz_policy.py [ROUNDS] [SEED]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
from deepgroebner_amd.rollout import PMLPPolicy

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
torch.manual_seed(rng.randint(0, 10 ** 6))
t0 = time.time()
for it in range(rounds):
    n = rng.choice([2, 3, 3, 3, 4, 5])
    k = rng.choice([1, 2, 2])
    dist = "%d-%d-%d-%s" % (n, rng.randint(3, 10 if n <= 3 else 5), rng.randint(3, 8 if n <= 3 else 4), rng.choice(["uniform", "weighted"]))
    hidden = rng.randint(33, 128)
    B = rng.choice([3, 64, 300])
    T = rng.choice([12, 40])
    R = 256 if n <= 3 else 1024
    caps = rng.choice([None, None, {"lds_max_basis": 16}, {"lds_max_basis": -1}])
    tag = "%s k=%d hidden=%d B=%d T=%d caps=%s" % (dist, k, hidden, B, T, caps)
    env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps)
    env.seed(np.arange(B) + rng.randint(0, 10 ** 6)); env.reset(); env.accounting(False)
    twin = env.copy(); twin.accounting(False)
    policy = PMLPPolicy(env.cols, [hidden]).cuda()
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
    try:
        env.rollout_device("first", 0, False, s, rew, done, rows, obs, R, True, False); env.sync()
        want = {key: [] for key in ("obs", "rows", "act", "logp", "rew", "done")}
        for t in range(T):
            want["obs"].append(obs.clone()); want["rows"].append(rows.clone())
            if t == 0:                                       # the torch module on the same block
                lp = policy(obs)
            env.policy_step_device(w["prepared"], w["hidden"], u[t], act, logp, rew, done, rows, obs, R, 1, s); env.sync()
            if t == 0 and not torch.allclose(logp, lp.gather(1, act.long()[:, None]).squeeze(1), atol=3e-4, rtol=1e-4):
                print("MISMATCH %s: kernel log-probabilities vs the torch module" % tag); sys.exit(1)
            for key, v in (("act", act), ("logp", logp), ("rew", rew), ("done", done)):
                want[key].append(v.clone())
        A = torch.zeros((T, B), dtype=torch.int32, device="cuda"); L = torch.zeros((T, B), dtype=torch.float32, device="cuda")
        Rw = torch.zeros((T, B), dtype=torch.float64, device="cuda"); D = torch.zeros((T, B), dtype=torch.uint8, device="cuda")
        N = torch.zeros((T, B), dtype=torch.int32, device="cuda")
        O = torch.full((T, B, R, env.cols), -1, dtype=torch.int32, device="cuda")
        cut = rng.randint(1, T - 1)
        try:
            twin.policy_rollout_device(w["prepared"], w["hidden"], cut, u[:cut], A[:cut], L[:cut], Rw[:cut], D[:cut], N[:cut], O[:cut], R, B * R * env.cols, s)
            twin.sync()
            twin.policy_rollout_device(w["prepared"], w["hidden"], T - cut, u[cut:], A[cut:], L[cut:], Rw[cut:], D[cut:], N[cut:], O[cut:], R, B * R * env.cols, s)
            twin.sync()
        except _ffi.BbxError as ex:
            if ex.code == -5:
                print("unsupported (per-step path only) %s" % tag); continue
            raise
    except _ffi.BbxError as ex:
        if ex.code == -3:
            print("CAPACITY %s: %s" % (tag, str(ex)[:160])); sys.exit(1)   # (default capacities: a failure)
        print("ERROR %s: %s" % (tag, str(ex)[:200])); sys.exit(1)
    for t in range(T):
        live = torch.arange(R, device="cuda")[None, :] < N[t][:, None]
        okk = (torch.equal(N[t], want["rows"][t]) and torch.equal(A[t], want["act"][t]) and torch.equal(L[t], want["logp"][t]) and
               torch.equal(Rw[t], want["rew"][t]) and torch.equal(D[t], want["done"][t]) and torch.equal(O[t][live], want["obs"][t][live]))
        if not okk:
            print("MISMATCH %s: step %d (cut %d)" % (tag, t, cut)); sys.exit(1)
    if not np.array_equal(env.stats()[:, :5], twin.stats()[:, :5]):
        print("MISMATCH %s: counters" % tag); sys.exit(1)
    # the deeper kernels (bbx_pmlp2_act / bbx_pmlp3_act) on the final block of this rollout: random layer sizes against the
    # torch module — log-probability of the drawn row, and the draw itself up to round-off ties
    deep = [rng.randint(1, 128) for _ in range(rng.choice([2, 3]))]
    pol2 = PMLPPolicy(env.cols, deep).cuda()
    with torch.no_grad():
        for lin in list(pol2.embedding) + [pol2.deciding]:
            lin.weight.mul_(0.3)
    live_rows = torch.clamp(rows, min=1)
    uu = torch.rand(B, device="cuda")
    a_k, l_k = pol2.act(obs, live_rows, uu)
    a_t, _ = pol2.act_torch(obs, live_rows, uu)
    torch.cuda.synchronize()
    lp2 = pol2(obs)
    n_eff = torch.clamp(live_rows, max=R)
    if pol2.deep_ok(env.cols) and R <= 1024:
        # (the reference masks by the -1 padding, the kernel by the row count: compare on the rows both see)
        validm = torch.arange(R, device="cuda")[None, :] < n_eff[:, None]
        lpm = torch.log_softmax(torch.where(validm, lp2, torch.full_like(lp2, -1e30)), dim=1)
        if not ((a_k >= 0).all() and (a_k < n_eff).all()):
            print("MISMATCH %s: deep policy %s drew outside the rows" % (tag, deep)); sys.exit(1)
        if not torch.allclose(l_k, lpm.gather(1, a_k.long()[:, None]).squeeze(1), atol=5e-4, rtol=1e-4):
            print("MISMATCH %s: deep policy %s log-probabilities vs the torch module" % (tag, deep)); sys.exit(1)
        if (a_k == a_t).float().mean() < 0.98:
            print("MISMATCH %s: deep policy %s draws" % (tag, deep)); sys.exit(1)
        tag += " deep=%s" % deep
    print("ok " + tag)
    del env, twin, O
print("fuzz_policy: %d rounds, %.0f s, no mismatch" % (rounds, time.time() - t0))
