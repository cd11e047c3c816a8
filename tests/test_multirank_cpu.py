"""N>1 path on CPU: two gloo ranks each run their shard of the global batch (the oracle stands in for
the device here — this test is about the placement logic: ids, seeds, aggregation); the gathered
result must equal a single-process run over the whole batch, i.e. results do not depend on placement."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_shard(rank, world, per_rank, T):
    sys.path.insert(0, ROOT)
    from deepgroebner_amd.shard import plan
    from oracle import ffi
    bo = ffi.load("bo")
    pl = plan(rank, world, per_rank)
    out = np.zeros((per_rank, 3), dtype=np.int64)
    for i in range(per_rank):
        env = bo.env("3-20-10-weighted")
        env.seed(int(pl["ideal_seeds"][i]))
        env.reset()
        adds = 0
        for t in range(T):
            adds += int(-env.step(ffi.agent_action(int(pl["agent_seeds"][i]), t, env.nP)))
            if env.nP == 0:
                env.reset()
        out[i] = (pl["ids"][i], adds, env.nG)
    return out


def _worker(rank, world, port, per_rank, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.from_numpy(_run_shard(rank, world, per_rank, T))
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.barrier()
    dist.all_gather(gathered, mine)                       # test-side aggregation only
    total = mine[:, 1].sum().clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM)          # what bench.py does with its counters
    if rank == 0:
        q.put((torch.cat(gathered).numpy(), int(total)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    world, per_rank, T = 2, 6, 80
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _run_shard(0, 1, world * per_rank, T)          # the same global batch in one process
    assert np.array_equal(got, want)
    assert total == int(want[:, 1].sum())


def test_plan_is_a_partition():
    sys.path.insert(0, ROOT)
    from deepgroebner_amd.shard import plan
    ids = np.concatenate([plan(r, 8, 4096)["ids"] for r in range(8)])
    assert np.array_equal(ids, np.arange(8 * 4096))
    assert plan(3, 8, 4096)["ideal_seeds"][0] == 1000 + 3 * 4096


def _bench_rehearsal(extra, env_extra=None):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--repeats", "7", "--batch", "64",
                           "--rehearse-plumbing"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT), json


def test_eight_rank_launch_plumbing_without_gpus():
    """The 8-GPU launch of the driver, rehearsed where there is no GPU at all: `bench.py --gpus 8 --rehearse-plumbing` starts
    its eight ranks (torch.distributed.run on 127.0.0.1), one of them builds, all meet at the barrier, check that no two
    share a device, agree on the repeat count, reduce their times and counters over gloo, and rank 0 prints exactly ONE line
    — with a stand-in that only counts the steps issued in place of the device, so the line carries no measurement
    (value null, "rehearsal": true).  Nothing here can tell how fast anything is; it can tell that the first real 8-GPU run
    does not die of plumbing."""
    p, json = _bench_rehearsal(["--gpus", "8"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["value"] is None and d["roofline"] is None and d["cpu_baseline"] is None
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 * 64 and d["config"]["devices_visible"] == 8
    assert d["oversubscribed"] is False and len(d["per_rank_value"]) == 8 and d["repeats"] == 7 and d["timed_steps"] == 140
    assert d["scaling"] == "weak" and "no collectives" in d["config"]["parallelism"]


def test_eight_ranks_on_four_devices_are_refused():
    p, _ = _bench_rehearsal(["--gpus", "8"], {"BBX_REHEARSE_DEVICES": "4"})
    assert p.returncode != 0 and "refusing to oversubscribe" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
