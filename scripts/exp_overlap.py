"""Experiment (GPU box): throughput of K-step launches of the headline workload (3-20-10-weighted, 4096 environments, counter-hash
agent, observation every step) queued on one stream against the same launches overlapped over n internal streams
(bbx_streams), for several K; every configuration is checked against the oracle's counters.
    python scripts/exp_overlap.py [K,K,...] [n,n,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi
B, R = 4096, 256
Ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 5, 20, 64, 1024]
Ns = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4]
bo = ffi.load("bo")
s = torch.cuda.current_stream()
for K in Ks:
    reps = max(3, min(int(os.environ.get("REPS_CAP", "2000")), int(int(os.environ.get("TOTAL", "40000")) / K)))
    total = 256 + reps * K
    want = None
    for n in Ns:
        env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
        env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
        obs = torch.empty((B, R, env.cols), dtype=torch.int32, device="cuda")
        rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
        rows = torch.zeros(B, dtype=torch.int32, device="cuda")
        env.rollout_device("random", 256, True, s.cuda_stream, rew, done, rows, obs, R, False, True); env.sync()
        env.persistent(n > 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            env.rollout_device("random", K, True, s.cuda_stream, rew, done, rows, obs, R, False, True)
        env.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = env.stats()
        ok = "unchecked"
        if total <= 6000:
            if want is None:
                want = bo.run_random_many("3-20-10-weighted", 2, range(1000, 1000 + B), range(B), total, True, 0)
            ok = "ok" if all(np.array_equal(st[:, c], np.array([r[k] for r in want])) for k, c in (("steps", 0), ("additions", 1), ("episodes", 2), ("nG", 7))) else "MISMATCH"
            ok += "" if np.array_equal(rows.cpu().numpy(), np.array([r["nP"] for r in want])) else " ROWS-MISMATCH"
        else:
            ok = "steps ok" if (st[:, 0] == total).all() and (st[:, 4] == 0).all() else "MISMATCH"
        print("K=%4d persistent=%d reps=%d: %7.1f us per launch, %6.3f us per step, %7.1f M env-steps/s  [%s] %s" % (K, n > 1, reps, dt / reps * 1e6, dt / reps / K * 1e6, B * K * reps / dt / 1e6, ok, env.session_stats() if n > 1 else ""), flush=True)
        del env
