#!/usr/bin/env python3
"""bench.py — env-steps/s of the BuchbergerEnv step path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over the whole batch: every environment of the batch performs
one BuchbergerEnv step (pair selection by the built-in counter-hash random agent, S-polynomial,
full reduction over GF(32003), Gebauer-Moeller update, lead-monomial observation of the new state;
finished episodes draw their next pre-generated ideal on the device).  Workload = BASELINE.json
configs[1]: 3-20-10-weighted, batch 4096 per GPU, k=2, random-selection agent.  Inputs (seeded
ideals) are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and
`cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIST = "3-20-10-weighted"
BATCH = 4096          # environments per GPU
K_LEADS = 2
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--dist", default=DIST)
    ap.add_argument("--chunk", type=int, default=0, help="steps per kernel launch (0 = all K in one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ablate-obs", action="store_true", help="diagnostic only: do not write per-step observations")
    ap.add_argument("--cpu-sample-envs", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as graft
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if rank == 0:
        graft.build()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()
    else:
        torch.cuda.set_device(local_rank)
    from deepgroebner_amd import VecLeadMonomialsEnv

    K, Wm, B = args.steps, args.warmup, args.batch
    chunk = args.chunk if args.chunk > 0 else max(K, 1)
    # enough pre-generated ideals per environment for warmup + timed steps (mean episode ~60 steps;
    # the library refills and resumes if an environment still runs dry, so this only affects speed)
    slots = (K + Wm) // 4 + 16
    env = VecLeadMonomialsEnv(args.dist, batch=B, k=K_LEADS, device=local_rank, caps={"queue_slots": slots})
    from deepgroebner_amd.shard import plan
    pl = plan(rank, world, B)                         # contiguous block of global environment ids
    env.seed(pl["ideal_seeds"])
    env.seed_agent(pl["agent_seeds"])
    env.reset()

    stream = torch.cuda.current_stream()
    cols = env.cols
    obs_rows = 128
    d_obs = torch.empty((B, obs_rows, cols), dtype=torch.int32, device="cuda")
    d_rew = torch.empty(B, dtype=torch.float64, device="cuda")
    d_done = torch.empty(B, dtype=torch.uint8, device="cuda")
    d_rows = torch.empty(B, dtype=torch.int32, device="cuda")

    def run(nsteps):
        done = 0
        while done < nsteps:
            n = min(chunk, nsteps - done)
            env.rollout_device("random", n, True, stream.cuda_stream, d_rew, d_done, d_rows, d_obs, obs_rows, False, not args.ablate_obs)
            env.sync()
            done += n

    if Wm > 0:
        run(Wm)
    env.prefetch()                                  # inputs for the timed region resident in HBM
    st0 = env.stats()
    twin = env.copy()                               # same state, same queued ideals: used after the timed region
    env.accounting(False)                           # timed run: lean kernel (no per-step byte counting)
    env.timing(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    run(K)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    region_ms = ev0.elapsed_time(ev1)
    kernel_ms, nlaunch = env.timing(False)       # HIP events around each step-kernel launch, on its stream
    st1 = env.stats()
    d = st1 - st0
    steps_done = int(d[:, 0].sum())
    assert steps_done == K * B, "every environment must have executed exactly K steps (%d != %d)" % (steps_done, K * B)
    assert (st1[:, 4] == 0).all(), "an environment reported an error status"
    additions = int(d[:, 1].sum())
    # algorithmic bytes of exactly these K steps: replay them on the twin with the accounting kernel (untimed)
    twin.accounting(True)
    twin.rollout("random", K, auto_reset=True)
    dt = twin.stats() - st0
    assert np.array_equal(dt[:, :2], d[:, :2]) and np.array_equal(twin.stats()[:, 7], st1[:, 7]), "accounting replay diverged"
    alg_bytes = int(dt[:, 6].sum())
    del twin

    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
        s = torch.tensor([steps_done, additions, alg_bytes], dtype=torch.int64, device="cuda")
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        steps_done, additions, alg_bytes = int(s[0]), int(s[1]), int(s[2])

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, env, st1, K + Wm, B)

    traffic, traffic_src = None, None
    if rank == 0:
        # HBM bytes of one launch from rocprofv3 PMC passes (scripts/profile_bench.sh; separate --pmc runs, FETCH_SIZE
        # doubled per MI355X_MICROARCH.md): only valid for the workload it was collected on
        pf = os.path.join(ROOT, "profiles", "r01_pmc_fast_kernel.json")
        if os.path.exists(pf) and args.dist == DIST and B == BATCH and K == 1024 and chunk == 1024 and not args.ablate_obs:
            traffic = json.load(open(pf)).get("hbm_traffic_bytes_per_launch")
            traffic_src = "profiles/r01_pmc_fast_kernel.json"
    if rank == 0:
        value = steps_done / elapsed
        # the dominant (only) kernel: per launch, algorithmic bytes of ONE GPU / its HIP-event duration
        per_launch_bytes = alg_bytes / world / nlaunch
        per_launch_s = kernel_ms * 1e-3 / nlaunch
        achieved = per_launch_bytes / per_launch_s / 1e9
        out = {
            "metric": "env steps/sec (polynomial additions) on 3-20-10-weighted, batch=4096, 1/2/4/8 GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16 exponents / u32 GF(32003)", "data": "synthetic (random binomial ideals drawn on the device from per-environment seeds, inside the timed region)",
            "config": {"workload": "%s k=%d batch=%d/GPU random-hash agent auto-reset" % (args.dist, K_LEADS, B),
                       "global_batch": B * world, "steps_per_launch": chunk, "parallelism": "env-sharded x%d, no collectives" % world},
            "additions_per_s": additions / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "alg_bytes_per_launch": per_launch_bytes,
                         "kernel": "bbx_fast_headline_kernel" if not args.ablate_obs else "bbx_fast_kernel<false,false>", "alg_bytes_per_env_step": alg_bytes / steps_done,
                         "kernel_ms_per_launch": kernel_ms / nlaunch, "launches": nlaunch, "timed_region_ms": region_ms},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, env, st_final, steps_per_env, B):
    """Time the reference C++ itself (oracle/_ref, kind 'reference') — or our C restatement when the
    prebuilt reference library did not travel (kind 'port') — single-threaded on the host, on a
    bounded sample of the SAME workload: the first n environments (same ideal seeds, same agent
    seeds) for the same number of steps.  Their addition totals must equal the device's."""
    from oracle import ffi
    kind = "reference" if ffi.available("ref") else "port"
    lib = ffi.load("ref" if kind == "reference" else "bo")
    n = args.cpu_sample_envs or max(1, min(B, int(4.5e6 // max(1, steps_per_env))))   # ~10-12 s of one host core
    res = lib.bench_random(args.dist, K_LEADS, n, steps_per_env, 1000, 0)
    dev_adds = int(st_final[:n, 1].sum())
    return {"value": res["steps"] / res["seconds"], "unit": "env-steps/s", "cores": 1, "kind": kind,
            "sample": "envs 0..%d of the same batch x %d steps (warmup+timed), %d steps, %.1f s" % (n - 1, steps_per_env, res["steps"], res["seconds"]),
            "additions_match_device": bool(res["additions"] == dev_adds)}


if __name__ == "__main__":
    main()
