#!/usr/bin/env python3
"""bench.py — env-steps/s of the BuchbergerEnv step path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over the whole batch: every environment of the batch performs
one BuchbergerEnv step (pair selection by the built-in counter-hash random agent, S-polynomial,
full reduction over GF(32003), Gebauer-Moeller update, lead-monomial observation of the new state;
finished episodes draw their next ideal on the device).  Workload = BASELINE.json configs[1]:
3-20-10-weighted, batch 4096 per GPU, k=2, random-selection agent.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Measurement protocol (so that the line means the same at --steps 20 and at --steps 1024):
  * pre-roll: every environment runs `preroll_steps` (>= 256) untimed steps first, so that the batch is in its steady
    state (a mix of episode phases) rather than all environments in their first episode;
  * W warm-up steps (one launch), then eight chained calibration launches of K steps (untimed);
  * timed region: R back-to-back launches of K steps each (`repeats`; R is chosen so that the region lasts >= 0.5 s),
    enqueued asynchronously, bracketed by barrier + synchronize on both sides; `ms_per_step` is the region's wall time /
    (R*K), `value` = batch * R * K / wall time (max over ranks).  The launches go through a persistent session
    (bbx_persistent, include/bbx.h): the first starts the step kernel, the others raise a device-visible step counter
    its waves look at, so environments never wait for each other between launches (`config.launch_path`;
    --no-persistent issues one kernel per launch, the round-2 path);
  * `roofline.kernel_ms_per_launch` = HIP-event time of the region / R: the events are recorded on the caller's stream,
    the first before the first launch, the second behind bbx_join (which makes that stream wait, on the device, for the
    session's kernels).

Prints ONE JSON line on rank 0 with `roofline` and `cpu_baseline` objects (the only thing that reaches stdout: everything
else, native libraries' chatter included, is sent to stderr).  With --gpus N > 1 and no launcher
environment, the ranks are started here as fresh child processes (before anything touches the GPU).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIST = "3-20-10-weighted"
BATCH = 4096          # environments per GPU
K_LEADS = 2
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
PREROLL = 256
MIN_REGION_MS = 500.0  # a sustained figure: long enough to contain the rare environments whose bases grow large
LONG_REGION_MS = 250.0
PMC_PROFILES = [os.path.join("profiles", "r04_pmc_fast_kernel.json"), os.path.join("profiles", "r03_pmc_fast_kernel.json")]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--dist", default=DIST)
    ap.add_argument("--repeats", type=int, default=0, help="launches of K steps in the timed region (0 = enough for >= 0.5 s)")
    ap.add_argument("--preroll", type=int, default=PREROLL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic-kernel", action="store_true",
                    help="diagnostic: pad the observation with -1 (obs_fill), which selects bbx_fast_kernel<false,false> "
                         "instead of the compile-time specialised headline variant")
    ap.add_argument("--cpu-sample-envs", type=int, default=0)
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the one-environment-per-core leg of the CPU baseline")
    ap.add_argument("--no-long-launch", action="store_true",
                    help="skip the context figure after the timed region (the same workload in 8 launches of 1024 steps; skip it under "
                         "rocprofv3: it would mix two launch lengths into the kernel's average)")
    ap.add_argument("--no-persistent", action="store_true", help="one kernel per launch instead of a persistent session")
    ap.add_argument("--allow-oversubscribe", action="store_true",
                    help="rehearsal only: accept more ranks than visible GPUs (the line then says \"oversubscribed\": true)")
    ap.add_argument("--rehearse-plumbing", action="store_true",
                    help="no GPU is touched and nothing is measured: a stand-in that only counts the steps issued to it takes the device's "
                         "place, so that the rank plumbing of an N-GPU run (rendezvous, build-once barrier, device-identity check, agreement "
                         "on the repeat count, max-over-ranks time, ONE line from rank 0) can be rehearsed on a box without N GPUs; the "
                         "line says \"rehearsal\": true and carries value null")
    return ap.parse_args()


class _CountingStandIn:
    """--rehearse-plumbing: the calls bench.py makes on a batch, answered by step counters alone (no device, no oracle, no
    arithmetic: there is nothing to measure and nothing is reported as measured)."""

    def __init__(self, batch, counters=None):
        import numpy as np
        self.B, self.cols = batch, 12
        self.c = np.zeros((batch, 8), dtype=np.int64) if counters is None else counters.copy()

    def seed(self, seeds): pass
    def seed_agent(self, seeds): pass
    def reset(self): pass
    def accounting(self, on): pass
    def persistent(self, on): pass
    def sync(self): pass
    def join(self, stream=0): pass
    def stats(self): return self.c.copy()
    def session_stats(self): return {"sessions": 0, "joined": 0, "later_kernel_steps": 0, "kernels": 0, "spills": 0}
    def copy(self): return _CountingStandIn(self.B, self.c)

    def rollout_device(self, agent, nsteps, *rest):
        self.c[:, 0] += nsteps; self.c[:, 1] += 3 * nsteps; self.c[:, 6] += 1578 * nsteps

    def rollout(self, agent, nsteps, auto_reset=True):
        self.rollout_device(agent, nsteps)


class _HostEvent:
    def __init__(self, enable_timing=True): self.t = 0.0
    def record(self, stream=None): self.t = time.perf_counter()
    def elapsed_time(self, other): return (other.t - self.t) * 1e3


def spawn_ranks(args):
    """--gpus N without a launcher: start N fresh ranks (torch.distributed.run) and relay rank 0's line."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("--gpus and --steps must be positive, --warmup non-negative")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))               # nothing has touched the GPU in this process
    # Exactly ONE line goes to stdout: whatever libraries print there from native code (gloo announces its rendezvous on
    # stdout) is sent to stderr for the rest of the run, and the result line is written to the saved descriptor.
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))

    import numpy as np
    import torch
    import __graft_entry__ as graft

    dist = None
    if world > 1:
        # environments never exchange data: the process group only carries the barrier and a few host scalars,
        # so it is a CPU (gloo) group — no collective on the data path, no RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    rehearsal = args.rehearse_plumbing
    if local_rank == 0:
        graft.build()                                 # (one rank builds, the others wait at the barrier: the library is shared)
    if world > 1:
        dist.barrier()
    ndev = int(os.environ.get("BBX_REHEARSE_DEVICES", world)) if rehearsal else torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("no GPU visible: libbbx has no CPU fallback")
    # one rank per distinct GPU, or the line must say otherwise: a scaling figure from ranks that share a device is not one
    oversubscribed = world > ndev
    if oversubscribed and not args.allow_oversubscribe:
        raise SystemExit("--gpus %d but only %d GPU(s) visible: refusing to oversubscribe (rehearsals: --allow-oversubscribe)" % (world, ndev))
    device = local_rank % ndev
    if rehearsal:
        uuid = "rehearsal-device-%d" % device
    else:
        torch.cuda.set_device(device)
        uuid = str(torch.cuda.get_device_properties(device).uuid) if hasattr(torch.cuda.get_device_properties(device), "uuid") else "dev%d" % device
    if world > 1:
        uuids = [None] * world
        dist.all_gather_object(uuids, uuid)
        if not oversubscribed and len(set(uuids)) != world:
            raise SystemExit("ranks share a GPU (device ids %s) although %d are visible" % (uuids, ndev))
    from deepgroebner_amd import VecLeadMonomialsEnv
    from deepgroebner_amd.shard import plan

    K, Wm, B = args.steps, args.warmup, args.batch
    if rehearsal:                                     # counters in place of the device; the product path below is otherwise unchanged
        import types
        env = _CountingStandIn(B)
        tdev, gpu_sync, Event = "cpu", (lambda: None), _HostEvent
        stream = types.SimpleNamespace(cuda_stream=0)
    else:
        env = VecLeadMonomialsEnv(args.dist, batch=B, k=K_LEADS, device=device)
        tdev, gpu_sync, Event = "cuda", torch.cuda.synchronize, torch.cuda.Event
        stream = torch.cuda.current_stream()
    pl = plan(rank, world, B)                         # contiguous block of global environment ids
    env.seed(pl["ideal_seeds"])
    env.seed_agent(pl["agent_seeds"])
    env.reset()

    cols = env.cols
    obs_rows = 512                                    # the fast class holds |P| <= 512: no observation row is ever cut
    d_obs = torch.empty((B, obs_rows, cols), dtype=torch.int32, device=tdev)
    d_rew = torch.empty(B, dtype=torch.float64, device=tdev)
    d_done = torch.empty(B, dtype=torch.uint8, device=tdev)
    d_rows = torch.empty(B, dtype=torch.int32, device=tdev)
    chain = not os.environ.get("BBX_HOST_GEN")        # device-drawn ideals: launches need no host service in between

    def launch(nsteps):
        env.rollout_device("random", nsteps, True, stream.cuda_stream, d_rew, d_done, d_rows, d_obs, obs_rows,
                           args.generic_kernel, True)
        if not chain:
            env.sync()

    env.accounting(False)                             # lean kernel (no per-step byte counting) from here on
    persistent = chain and not args.no_persistent and not args.generic_kernel
    env.persistent(persistent)
    launch(max(args.preroll, 0) or 1); env.sync()     # steady state
    if Wm > 0:
        launch(Wm); env.sync()
    gpu_sync()
    t0 = time.perf_counter()
    ncal = (64 if persistent else 8) if chain else 1 # calibration (also the untimed warm-up of this launch shape): chained
    for _ in range(ncal):                             # launches, so that the synchronisation is not mistaken for launch time
        launch(K)
    env.sync(); gpu_sync()
    t_launch = (time.perf_counter() - t0) / ncal
    R = args.repeats if args.repeats > 0 else int(min(1 << 16, max(3, MIN_REGION_MS * 1e-3 / max(t_launch, 1e-6) + 1)))
    if world > 1:
        r_all = [None] * world
        dist.all_gather_object(r_all, R)
        R = max(r_all)
    st0 = env.stats()
    sess0 = env.session_stats()
    twin = env.copy()                               # same state, same generator state: replayed after the timed region

    gpu_sync()
    if world > 1:
        dist.barrier()
    ev0, ev1 = Event(enable_timing=True), Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(R):
        launch(K)
    env.join(stream.cuda_stream)                    # (persistent session: the caller's stream waits for its kernels, on the device)
    ev1.record(stream)
    env.sync()
    gpu_sync()
    t1 = time.perf_counter()
    if world > 1:
        dist.barrier()
    elapsed = t1 - t0
    region_ms = ev0.elapsed_time(ev1)               # HIP events on the stream the kernels were launched on
    st1 = env.stats()
    # context, outside the timed region: the same kernel in launches of 1024 steps (how a rollout would normally be
    # issued; launches of few steps end with the waves that met an episode reset)
    long_launch = None
    sess = env.session_stats()
    if not args.no_long_launch and world == 1:
        launch(64)                                  # (the statistics call above closed the session: a new one gets going)
        env.join(stream.cuda_stream)
        n_long = int(max(8, LONG_REGION_MS / max(1024 * region_ms / (R * K), 1e-6) + 1))
        ev2, ev3 = Event(enable_timing=True), Event(enable_timing=True)
        ev2.record(stream)
        for _ in range(n_long):
            launch(1024)
        env.join(stream.cuda_stream)
        ev3.record(stream)
        env.sync(); gpu_sync()
        long_ms = ev2.elapsed_time(ev3)
        long_launch = {"steps_per_launch": 1024, "launches": n_long, "value": n_long * 1024 * B / (long_ms * 1e-3), "unit": "env-steps/s",
                       "region_ms": long_ms,
                       "timing": "HIP events on the launch stream around the launches (and the join behind them)",
                       "note": "context only: the same trajectories later on, issued 1024 steps per launch, >= 0.25 s"}
    sess_end = env.session_stats()
    d = st1 - st0
    steps_done = int(d[:, 0].sum())
    assert steps_done == R * K * B, "every environment must have executed exactly R*K steps (%d != %d)" % (steps_done, R * K * B)
    assert (st1[:, 4] == 0).all(), "an environment reported an error status"
    assert rehearsal or int(d_rows.max().item()) <= obs_rows
    additions = int(d[:, 1].sum())
    # algorithmic bytes of exactly these steps: replay them on the twin with the accounting kernel (untimed)
    twin.accounting(True)
    for _ in range(R):
        twin.rollout("random", K, auto_reset=True)
    st_twin = twin.stats()
    dt = st_twin - st0
    assert np.array_equal(dt[:, :2], d[:, :2]) and np.array_equal(st_twin[:, 7], st1[:, 7]), "accounting replay diverged"
    alg_bytes = int(dt[:, 6].sum())
    del twin

    per_rank = [steps_done / elapsed]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, steps_done / elapsed)
        t = torch.tensor([elapsed, region_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, region_ms = float(t[0]), float(t[1])
        s = torch.tensor([steps_done, additions, alg_bytes], dtype=torch.int64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        steps_done, additions, alg_bytes = int(s[0]), int(s[1]), int(s[2])

    cpu = cpu_all = None
    if rank == 0 and not args.no_cpu_baseline and not rehearsal:
        cpu = cpu_baseline(args, st1, int(st1[0, 0]), B)
        if not args.no_cpu_all_cores:
            cpu_all = cpu_baseline_all_cores(args, st1, int(st1[0, 0]), B, cpu)

    if rank == 0:
        value = None if rehearsal else steps_done / elapsed
        kernel = ("bbx_fast_headline_persistent_kernel" if persistent else "bbx_fast_headline_kernel") if not args.generic_kernel else "bbx_fast_kernel<false,false>"
        # the dominant (only) kernel: per launch, algorithmic bytes of ONE GPU / its HIP-event duration
        per_launch_bytes = alg_bytes / world / R
        per_launch_s = region_ms * 1e-3 / R
        achieved = per_launch_bytes / per_launch_s / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "traffic_unit": "bytes/launch", "traffic_source": None,
                "note": "achieved = ALGORITHMIC bytes (SURVEY 8d formula) / launch time: a nominal figure — the state is "
                        "register/LDS resident, the measured HBM traffic (`traffic`) is a fraction of a percent of it, and "
                        "the kernel is bound by per-wave instruction issue (see `issue_bound`)",
                "alg_bytes_per_launch": per_launch_bytes, "alg_bytes_per_env_step": alg_bytes / steps_done,
                "kernel": kernel, "kernel_ms_per_launch": region_ms / R, "launches": R, "timed_region_ms": region_ms,
                "kernels_in_timed_region": (sess["kernels"] - sess0["kernels"]) if persistent else R,
                # every batch step this process pushed through that kernel (pre-roll, warm-up, calibration, timed region): a
                # rocprofv3 --stats run of the same command shows the kernel's total time, total / this = time per batch step
                "batch_steps_through_kernel": (max(args.preroll, 0) or 1) + Wm + ncal * K + R * K + ((64 + long_launch["launches"] * 1024) if long_launch else 0),
                "launch_note": ("the R launches of the timed region are served by the kernels of one persistent session (time slices of 10 ms); "
                                "kernel_ms_per_launch = HIP-event time of the region / R") if persistent else "one kernel per launch"}
        PMC_PROFILE = next((q for q in PMC_PROFILES if os.path.exists(os.path.join(ROOT, q))), None)
        if PMC_PROFILE and args.dist == DIST and B == BATCH and not args.generic_kernel:
            prof = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
            tr = prof.get("hbm_traffic")
            if tr:   # PMC passes at two launch lengths: traffic = fixed part (records in/out, observation block) + per-step part
                roof["traffic"] = tr["fixed_bytes_per_launch"] + tr["bytes_per_batch_step"] * K
                roof["traffic_source"] = PMC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes, FETCH doubled)"
                # counter collection serialises kernels, so the counters are those of the one-kernel-per-launch form of the same
                # step body (fast_body): say which kernel was profiled when it is not the one that is timed
                roof["traffic_kernel"] = prof.get("kernel")
                roof["traffic_kernel_is_timed_kernel"] = prof.get("kernel") == kernel
            ib = prof.get("issue_bound")
            if ib:   # the binding resource: scalar-pipe instructions per cycle per CU against the one scalar unit of a CU
                roof["issue_bound"] = dict(ib, source=PMC_PROFILE, kernel=prof.get("kernel"))
                roof["binding"] = "issue"
                roof["binding_note"] = ("`bound`/`frac` are the HBM figure SURVEY 8d prescribes (algorithmic bytes against 8 TB/s); what limits "
                                        "this kernel is instruction issue (`issue_bound`): the state never leaves registers / LDS / L2")
        out = {
            "metric": "env steps/sec (polynomial additions) on 3-20-10-weighted, batch=4096, 1/2/4/8 GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": None if rehearsal else elapsed * 1e3 / (R * K), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16 exponents / u32 GF(32003)",
            "data": ("NONE: plumbing rehearsal, no device, nothing measured" if rehearsal else
                     "synthetic (random binomial ideals drawn on the device from per-environment seeds, inside the timed region)"),
            "repeats": R, "preroll_steps": max(args.preroll, 0) or 1, "timed_steps": R * K, "elapsed_s": elapsed,
            "config": {"workload": "%s k=%d batch=%d/GPU random-hash agent auto-reset, observation written every step" % (args.dist, K_LEADS, B),
                       "global_batch": B * world, "steps_per_launch": K, "parallelism": "env-sharded x%d, no collectives" % world,
                       "devices_visible": ndev, "launch_path": "persistent session (bbx_persistent)" if persistent else "one kernel per launch"},
            "oversubscribed": bool(oversubscribed), "per_rank_value": [None] * world if rehearsal else per_rank,
            "rehearsal": bool(rehearsal),
            "additions": None if rehearsal else additions,
            "additions_per_s": None if rehearsal else additions / elapsed,
            "long_launch": long_launch,
            "session_stats": ({"timed_region": {k_: sess[k_] - sess0[k_] for k_ in sess}, "whole_run": sess_end,
                               "note": "spills = environments that left the register/LDS class (basis beyond 256 elements or 512 pairs) and "
                                       "were continued by the HBM-resident pass; later_kernel_steps = env-steps taken by closing kernels"}
                              if persistent else None),
            "roofline": None if rehearsal else roof,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
        }
        sys.stdout.flush()
        os.write(out_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, st_final, steps_per_env, B):
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
            "sample": "envs 0..%d of the same batch x %d steps each (pre-roll + warm-up + timed), %d steps, %.1f s" % (n - 1, steps_per_env, res["steps"], res["seconds"]),
            "additions_match_device": bool(res["additions"] == dev_adds)}


def _cpu_share():
    """Host cores this process may use: the affinity mask, cut down by a cgroup CPU quota where one is set."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = min(cores, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline_all_cores(args, st_final, steps_per_env, B, single, seconds=8.0):
    """SURVEY 8d-ii / BASELINE.md 3.3: the same CPU code with one environment stream per host core — one thread per core
    this process may use (`cores`; `nproc` = what the machine has), each running environments of its own slice of the same
    batch (same seeds as the device's), chunk after chunk until `seconds` have passed: a bounded sample whatever share of
    the machine the cores really are.  The library call releases the GIL; environments share nothing."""
    import threading
    from oracle import ffi
    kind = "reference" if ffi.available("ref") else "port"
    lib = ffi.load("ref" if kind == "reference" else "bo")
    cores = _cpu_share()
    per = max(1, B // cores)                                        # thread i owns environments [i*per, (i+1)*per)
    T = max(1, min(steps_per_env, int(single["value"] * 0.05)))     # one chunk = one environment x T steps, ~50 ms of a free core
    deadline = time.perf_counter() + seconds
    steps = [0] * cores

    def work(i):
        e = 0
        while time.perf_counter() < deadline:
            g = i * per + e % per
            steps[i] += lib.bench_random(args.dist, K_LEADS, 1, T, 1000 + g, g)["steps"]
            e += 1
    ths = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t0
    total = sum(steps)
    return {"value": total / wall, "unit": "env-steps/s", "cores": cores, "nproc": os.cpu_count(), "kind": kind, "threads": cores,
            "sample": "%d threads, each environments of its slice of the same batch x %d steps, for %.0f s: %d steps, %.1f s wall" % (cores, T, seconds, total, wall),
            "additions_match_device": None, "speedup_over_one_core": total / wall / single["value"]}


if __name__ == "__main__":
    main()
