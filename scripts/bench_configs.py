"""Throughput of the other BASELINE configs (not the contract bench line): steps/s on one GPU next to the
compiled reference on a sample of the same environments.  usage: bench_configs.py DIST BATCH STEPS [K_LEADS] [CPU_ENVS] [OBS_ROWS] [WIDE_WAVES]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi

dist, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cpu_envs = int(sys.argv[5]) if len(sys.argv) > 5 else 4
caps = {"queue_slots": max(8, T // 8 + 8)}
if len(sys.argv) > 7:
    caps["wide_waves"] = int(sys.argv[7])
torch.cuda.init()
env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
env.prefetch(); env.accounting(False)
# like bench.py: the observation matrix of every environment is materialised in HBM after every step
obs_rows = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
d_obs = torch.empty((B, obs_rows, env.cols), dtype=torch.int32, device="cuda")
d_rew = torch.empty(B, dtype=torch.float64, device="cuda"); d_done = torch.empty(B, dtype=torch.uint8, device="cuda")
d_rows = torch.empty(B, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
env.rollout_device("random", T, True, stream.cuda_stream, d_rew, d_done, d_rows, d_obs, obs_rows, False, True); env.sync()
t1 = time.perf_counter()
st = env.stats()
assert (st[:, 0] == T).all() and (st[:, 4] == 0).all(), st[:4]
lib = ffi.load("ref" if ffi.available("ref") else "bo")
res = lib.bench_random(dist, k, cpu_envs, T, 1000, 0)
ok = res["additions"] == int(st[:cpu_envs, 1].sum())
print(json.dumps({"dist": dist, "batch": B, "steps": T, "gpu_steps_per_s": B * T / (t1 - t0), "gpu_seconds": t1 - t0,
                  "additions_per_step": float(st[:, 1].sum()) / (B * T), "max_basis": int(st[:, 7].max()), "obs_rows_cap": obs_rows, "max_rows_last": int(d_rows.max().item()),
                  "cpu_steps_per_s": res["steps"] / res["seconds"], "cpu_kind": lib.kind, "cpu_envs": cpu_envs,
                  "additions_match_reference_on_sample": bool(ok)}))
