"""Throughput + roofline line of the BASELINE configs other than the contract bench line (bench.py): env-steps/s on one
GPU, the algorithmic bytes counted on the device (BbxHdr.alg_bytes, SURVEY 8d formula) over the HIP-event time of the
step kernel, next to the compiled reference on a sample of the same environments.

    python scripts/bench_configs.py cyclic-7 --batch 512 --steps 512
    python scripts/bench_configs.py 5-10-5-uniform --batch 4096 --steps 2048 --obs-rows 1024
    python scripts/bench_configs.py cyclic-7 --batch 1 --agent degree --to-completion
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi

ap = argparse.ArgumentParser()
ap.add_argument("dist")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=512)
ap.add_argument("--k", type=int, default=2)
ap.add_argument("--cpu-envs", type=int, default=4, help="first chunk of the CPU sample (0: no CPU leg)")
ap.add_argument("--cpu-seconds", type=float, default=2.0, help="the CPU sample grows until it has run this long")
ap.add_argument("--obs-rows", type=int, default=1024)
ap.add_argument("--wide-waves", type=int, default=0)
ap.add_argument("--wide-lds-terms", type=int, default=0)
ap.add_argument("--agent", default="random")
ap.add_argument("--to-completion", action="store_true", help="one rollout until every pair set is empty (no auto-reset)")
ap.add_argument("--no-obs", action="store_true", help="diagnostic: observation only at the end of the rollout")
ap.add_argument("--accounting", action="store_true", help="time the accounting variant itself")
ap.add_argument("--no-twin", action="store_true", help="profiling runs: no accounting replay (no algorithmic bytes, no roofline object)")
ap.add_argument("--kernel", default=None, help="name of the step kernel (for the roofline object)")
ap.add_argument("--profile", default=None, help="profiles/*.json with hbm_traffic_bytes_per_launch for this exact workload")
a = ap.parse_args()

B, T, k = a.batch, a.steps, a.k
caps = {"queue_slots": max(8, T // 8 + 8)}
if a.wide_waves:
    caps["wide_waves"] = a.wide_waves
if a.wide_lds_terms:
    caps["wide_lds_terms"] = a.wide_lds_terms
torch.cuda.init()
env = VecLeadMonomialsEnv(a.dist, batch=B, k=k, caps=caps)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
env.prefetch()
binomial3 = a.dist.startswith("3-") and "." not in a.dist
fixed = a.dist.startswith("cyclic")
twin = None
lean = False
if (binomial3 or fixed) and not a.accounting:
    lean = True
if lean and not a.no_twin:
    # classes with a lean variant (no algorithmic-byte counting): time the lean one, take the bytes — a property of
    # the workload — from a replay of the same steps on a copy with accounting on (like bench.py)
    twin = env.copy()
if lean:
    env.accounting(False)
d_obs = torch.empty((B, a.obs_rows, env.cols), dtype=torch.int32, device="cuda")
d_rew = torch.empty(B, dtype=torch.float64, device="cuda"); d_done = torch.empty(B, dtype=torch.uint8, device="cuda")
d_rows = torch.empty(B, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream()
st0 = env.stats()
env.timing(True)
torch.cuda.synchronize()
nsteps = (1 << 30) if a.to_completion else T
t0 = time.perf_counter()
env.rollout_device(a.agent, nsteps, not a.to_completion, stream.cuda_stream, d_rew, d_done, d_rows, d_obs, a.obs_rows, False, not a.no_obs)
env.sync()
t1 = time.perf_counter()
kernel_ms, launches = env.timing(False)
st = env.stats()
d = st - st0
assert (st[:, 4] == 0).all(), st[:4]
if not a.to_completion:
    assert (d[:, 0] == T).all(), d[:4]
steps = int(d[:, 0].sum()); adds = int(d[:, 1].sum()); alg = int(d[:, 6].sum())
if twin is not None:
    twin.accounting(True)
    twin.rollout(a.agent, nsteps, auto_reset=not a.to_completion)
    dtw = twin.stats() - st0
    assert np.array_equal(dtw[:, :2], d[:, :2]), "accounting replay diverged"
    alg = int(dtw[:, 6].sum())
    del twin
out = {"dist": a.dist, "batch": B, "steps_per_env": T if not a.to_completion else int(d[:, 0].max()), "agent": a.agent,
       "gpu_steps_per_s": steps / (t1 - t0), "gpu_additions_per_s": adds / (t1 - t0), "gpu_seconds": t1 - t0,
       "additions_per_step": adds / max(1, steps), "max_basis": int(st[:, 7].max()), "obs_rows_cap": a.obs_rows,
       "max_rows_last": int(d_rows.max().item())}
if alg > 0 and kernel_ms > 0:
    ach = alg / (kernel_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
            "alg_bytes_per_env_step": alg / max(1, steps), "alg_bytes_total": alg, "kernel": a.kernel,
            "kernel_ms_total": kernel_ms, "launches": launches}
    if a.profile and os.path.exists(a.profile):
        pj = json.load(open(a.profile))
        roof["traffic"] = pj.get("hbm_traffic_bytes_per_launch"); roof["traffic_source"] = a.profile
    out["roofline"] = roof
if a.cpu_envs > 0 and a.agent == "random" and not a.to_completion:
    # the compiled reference on the first environments of the same batch, in chunks that double until at least 2 s of CPU
    # time have been measured (a sample of a few milliseconds is noise) or the whole batch has been run
    lib = ffi.load("ref" if ffi.available("ref") else "bo")
    done, chunk, secs, csteps, cadds = 0, a.cpu_envs, 0.0, 0, 0
    while done < B and secs < a.cpu_seconds:
        n = min(chunk, B - done)
        res = lib.bench_random(a.dist, k, n, T, 1000 + done, done)
        secs += res["seconds"]; csteps += res["steps"]; cadds += res["additions"]
        done += n; chunk *= 2
    out["cpu_baseline"] = {"value": csteps / secs, "unit": "env-steps/s", "cores": 1,
                           "kind": "reference" if lib.kind == "ref" else "port",
                           "sample": "envs 0..%d x %d steps, %d steps, %.1f s" % (done - 1, T, csteps, secs),
                           "additions_match_device": bool(cadds == int(st[:done, 1].sum()))}
print(json.dumps(out))
