"""Randomised self-consistency sweep of persistent sessions (test infrastructure, GPU box): a batch with sessions enabled
(bbx_persistent: asynchronous rollout_device calls of one shape feed ONE resident kernel) against a twin that takes every
call as its own launch — random batch sizes, agents, steps per call, observation modes, and between the calls the things
that end a session (a call of another shape, host steps, masked resets, copies, state reads, joins on another stream).
After every synchronisation the two must agree on every counter, on the outputs in the callers' buffers and on the live
rows of the observation block.     python scripts/fuzz_sessions.py [ROUNDS] [SEED]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from deepgroebner_amd.rollout import PMLPPolicy

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
t_start = time.time()


oplog = []


def fail(msg):
    print("MISMATCH " + msg + "\n  ops: " + " | ".join(oplog)); sys.exit(1)


for it in range(rounds):
    dist = "3-%d-%d-%s" % (rng.randint(3, 20), rng.randint(2, 10), rng.choice(["uniform", "weighted", "maximum"]))
    k = rng.choice([1, 2])
    B = rng.choice([1, 4, 64, 300, 1000])
    caps = rng.choice([None, None, {"lds_max_basis": 16}, {"lds_max_basis": 24}])
    R = 512                                                 # (the register/LDS class holds at most 512 pairs: never a truncated block)
    tag = "%s k=%d B=%d caps=%s R=%d seed=%d/%d" % (dist, k, B, caps, R, seed, it)
    envs = []
    for persistent in (True, False):
        e = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps)
        e.seed(np.arange(B) + 17 * it); e.seed_agent(np.arange(B) + it); e.reset(); e.accounting(False)
        if persistent:
            e.persistent(True)
        envs.append(e)
    bufs = []
    for _ in envs:
        bufs.append({"rew": torch.zeros(B, dtype=torch.float64, device="cuda"), "done": torch.zeros(B, dtype=torch.uint8, device="cuda"),
                     "rows": torch.zeros(B, dtype=torch.int32, device="cuda"), "obs": torch.full((B, R, envs[0].cols), -1, dtype=torch.int32, device="cuda")})
    torch.manual_seed(seed * 1000 + it)
    policy = PMLPPolicy(envs[0].cols, [rng.choice([32, 64, 100, 128])]).cuda()
    with torch.no_grad():
        for lin in list(policy.embedding) + [policy.deciding]:
            lin.weight.mul_(0.3)
    pw = policy._fused_weights()
    U = torch.rand((64, B), device="cuda")
    for bf in bufs:
        bf["act"] = torch.zeros(B, dtype=torch.int32, device="cuda"); bf["logp"] = torch.zeros(B, dtype=torch.float32, device="cuda")
    u_at = 0
    side = torch.cuda.Stream()
    stream = torch.cuda.current_stream()
    shape = None                                            # (agent, K, obs mode) of the running sequence of calls

    def compare(where):
        errs = []
        for e in envs:
            try:
                e.sync()
                errs.append(None)
            except Exception as ex:                         # (rows beyond the block: an error of the block, raised by both or by neither)
                errs.append(str(ex)[:60])
        if (errs[0] is None) != (errs[1] is None):
            fail("%s: %s: one raised, the other did not: %s" % (tag, where, errs))
        torch.cuda.synchronize()
        a, b = envs[0].stats(), envs[1].stats()
        if not np.array_equal(a[:, :5], b[:, :5]) or not np.array_equal(a[:, 7], b[:, 7]):
            bad = int(np.flatnonzero((a[:, :5] != b[:, :5]).any(axis=1))[0])
            fail("%s: %s: counters of env %d: session %s twin %s" % (tag, where, bad, a[bad], b[bad]))
        if not np.array_equal(envs[0].rows, envs[1].rows):
            fail("%s: %s: row counts" % (tag, where))
        return errs[0] is None

    nops = rng.randint(4, 14)
    del oplog[:]
    for op_i in range(nops):
        op = rng.choice(["roll", "roll", "roll", "roll", "newshape", "hoststep", "reset", "copy", "read", "join", "sync", "pol", "pol", "stepdev", "value"])
        if op == "newshape" or (shape is None and op in ("roll", "sync")):
            shape = (rng.choice(["random", "degree", "first"]), rng.choice([1, 7, 64, 300]), rng.choice([0, 1, 2]))
            op = "roll"
        oplog.append(op + (" %s K=%d obs=%d" % shape if op == "roll" else ""))
        if op == "roll":
            agent, K, om = shape
            for e, bf in zip(envs, bufs):
                e.rollout_device(agent, K, True, stream.cuda_stream, bf["rew"], bf["done"], bf["rows"], bf["obs"] if om else None, R if om else 0, False, om == 2)
        elif op == "pol":                                      # per-step calls of the one-layer policy: they join sessions of their own
            n = rng.choice([1, 3, 20])
            if u_at + n > U.shape[0]:
                continue
            for e, bf in zip(envs, bufs):                       # (the policy reads the block the call before it left: write it first)
                e.rollout_device("first", 0, False, stream.cuda_stream, bf["rew"], bf["done"], bf["rows"], bf["obs"], R, True, False)
            if os.environ.get("FUZZ_DEBUG") and it == int(os.environ["FUZZ_DEBUG"]):
                for e in envs: e.sync()
                print("debug op %d: after the observation call: rew[0:4] %s / %s; stats %s" % (op_i, bufs[0]["rew"][:4].tolist(), bufs[1]["rew"][:4].tolist(), envs[0].session_stats()), flush=True)
            for t in range(n):
                for e, bf in zip(envs, bufs):
                    e.policy_step_device(pw["prepared"], pw["hidden"], U[u_at + t], bf["act"], bf["logp"], bf["rew"], bf["done"], bf["rows"], bf["obs"], R, 2, stream.cuda_stream)
            u_at += n
            ok = compare("policy steps, op %d" % op_i)
            if ok:
                for key in ("act", "logp", "rew", "done", "rows"):
                    if not torch.equal(bufs[0][key], bufs[1][key]):
                        bad = torch.nonzero(bufs[0][key] != bufs[1][key]).flatten()[:6].tolist()
                        fail("%s: policy output %s after op %d (n=%d): envs %s session %s twin %s; host rows %s / %s; done %s / %s" % (
                            tag, key, op_i, n, bad, bufs[0][key][bad].tolist(), bufs[1][key][bad].tolist(), envs[0].rows[bad].tolist(), envs[1].rows[bad].tolist(),
                            bufs[0]["done"][bad].tolist(), bufs[1]["done"][bad].tolist()) + " | session stats %s | steps taken %s / %s" % (
                            envs[0].session_stats(), envs[0].stats()[bad, 0].tolist(), envs[1].stats()[bad, 0].tolist()))
                live = torch.arange(R, device="cuda")[None, :] < bufs[0]["rows"][:, None]
                if not torch.equal(bufs[0]["obs"][live], bufs[1]["obs"][live]):
                    fail("%s: observation rows after policy steps, op %d" % (tag, op_i))
            shape = None
        elif op == "stepdev":                                  # caller-supplied actions from a device buffer: launches of their own, behind the session
            n = rng.choice([1, 4])
            zero = torch.zeros(B, dtype=torch.int32, device="cuda")
            for t in range(n):
                for e, bf in zip(envs, bufs):
                    e.step_device(zero, bf["rew"], bf["done"], bf["rows"], bf["obs"], R, 1, stream.cuda_stream, auto_reset=True)
            ok = compare("device steps, op %d" % op_i)
            if ok:
                for key in ("rew", "done", "rows"):
                    if not torch.equal(bufs[0][key], bufs[1][key]):
                        fail("%s: output %s after device steps, op %d" % (tag, key, op_i))
                live = torch.arange(R, device="cuda")[None, :] < bufs[0]["rows"][:, None]
                if not torch.equal(bufs[0]["obs"][live], bufs[1]["obs"][live]):
                    fail("%s: observation rows after device steps, op %d" % (tag, op_i))
            shape = None
        elif op == "value":
            try:
                v0, v1 = envs[0].values("degree", 0.99), envs[1].values("degree", 0.99)
            except Exception:
                compare("after a value() that raised, op %d" % op_i)
                continue
            if not np.array_equal(np.asarray(v0), np.asarray(v1)):
                fail("%s: value() at op %d" % (tag, op_i))
        elif op == "join":
            envs[0].join(side.cuda_stream)
            side.synchronize()
        elif op == "sync":
            ok = compare("sync after op %d" % op_i)
            if ok and shape is not None:
                for key in ("rew", "done", "rows"):
                    if not torch.equal(bufs[0][key], bufs[1][key]):
                        fail("%s: output %s after op %d" % (tag, key, op_i))
                if shape[2]:
                    live = torch.arange(R, device="cuda")[None, :] < bufs[0]["rows"][:, None]
                    if not torch.equal(bufs[0]["obs"][live], bufs[1]["obs"][live]):
                        fail("%s: observation rows after op %d" % (tag, op_i))
        elif op == "hoststep":
            compare("before host step %d" % op_i)
            acts = np.zeros(B, dtype=np.int32)
            outs = [e.step_ragged(acts, auto_reset=True) for e in envs]
            for x, y in zip(outs[0], outs[1]):
                if not np.array_equal(np.asarray(x), np.asarray(y)):
                    fail("%s: host step %d" % (tag, op_i))
        elif op == "reset":
            mask = (np.random.default_rng(it * 100 + op_i).random(B) < 0.3).astype(np.uint8)
            try:
                for e in envs:
                    e.reset(mask)
            except Exception:
                compare("after a reset that raised, op %d" % op_i)
        elif op == "copy":
            try:
                envs = [e.copy() for e in envs]
                envs[0].persistent(True)
            except Exception as ex:
                fail("%s: copy raised %s" % (tag, str(ex)[:80]))
        elif op == "read":
            try:
                s0, s1 = envs[0].state(0), envs[1].state(0)
            except Exception:
                compare("after a state read that raised, op %d" % op_i)
                continue
            for ci, (x, y) in enumerate(zip(s0, s1)):
                if repr(x) != repr(y):
                    a, b = envs[0].stats(), envs[1].stats()
                    fail("%s: state of env 0 at op %d, component %d: session %s twin %s; counters %s / %s" % (tag, op_i, ci, repr(x)[:300], repr(y)[:300], a[0], b[0]))
    compare("end")
    print("ok %s ops=%d sessions=%s" % (tag, nops, envs[0].session_stats()), flush=True)
    del envs, bufs
print("fuzz_sessions: %d rounds, %.0f s, no mismatch" % (rounds, time.time() - t_start))
