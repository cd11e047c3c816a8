"""The N > 1 launch path on whatever GPUs this box has (north_star: independent environments sharded over ranks, no
collective on the data path): fresh child processes — started before this process touches the GPU — each run their
shard of the global batch; the union must equal one process running the whole batch.  Also bench.py's own rank
plumbing (spawn of torch.distributed.run, gloo scalars, max-over-ranks time) end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(rank, world, B, T):
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_child.py"), str(rank), str(world), str(B), str(T), "0"],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_two_ranks_equal_one_process_running_the_union():
    B, T, world = 96, 150, 2
    procs = [_child(r, world, B, T) for r in range(world)]      # both at once, sharing GPU 0
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    whole = _child(0, 1, B * world, T)
    so, se = whole.communicate(timeout=600)
    assert whole.returncode == 0, se[-2000:]
    whole = json.loads(so.strip().splitlines()[-1])
    ids = sum((o["ids"] for o in sorted(outs, key=lambda o: o["rank"])), [])
    stats = sum((o["stats"] for o in sorted(outs, key=lambda o: o["rank"])), [])
    assert ids == whole["ids"] == list(range(B * world))
    assert np.array_equal(np.array(stats), np.array(whole["stats"]))
    assert (np.array(stats)[:, 0] == T).all()


def _bench(*extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--repeats", "3",
                        "--preroll", "40", "--no-cpu-baseline"] + list(extra), capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_bench_two_ranks_via_its_own_launcher():
    """`python bench.py --gpus 2` with no launcher environment starts the ranks itself; the two ranks' environments are
    the first 2 x B global ids, so their additions over the timed region equal a 1-rank run of batch 2 x B."""
    two = _bench("--gpus", "2", "--batch", "128", "--allow-oversubscribe")
    one = _bench("--gpus", "1", "--batch", "256")
    assert two["oversubscribed"] is True and one["oversubscribed"] is False and len(two["per_rank_value"]) == 2
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["global_batch"] == one["config"]["global_batch"] == 256
    assert two["repeats"] == one["repeats"] == 3 and two["steps"] == 8 and two["warmup"] == 2
    assert two["additions"] == one["additions"] > 0
    assert two["roofline"]["alg_bytes_per_env_step"] == one["roofline"]["alg_bytes_per_env_step"]
    for r in (one, two):
        assert r["value"] > 0 and r["scaling"] == "weak" and r["roofline"]["launches"] == 3


def test_four_ranks_sharing_the_device_equal_one_process_and_bench_runs_four():
    """Four ranks at once — a GPU box of this pool lets one user have six processes on a card, and the test runner and the
    launcher count (the 8-rank launch is rehearsed without GPUs in tests/test_multirank_cpu.py) — each with its shard: their
    union equals one process running all of it, and `bench.py --gpus 4` (oversubscribed, said so in the line) goes through
    rendezvous, build barrier and reduction."""
    B, T, world = 32, 100, 4
    procs = [_child(r, world, B, T) for r in range(world)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    whole = _child(0, 1, B * world, T)
    so, se = whole.communicate(timeout=600)
    assert whole.returncode == 0, se[-2000:]
    whole = json.loads(so.strip().splitlines()[-1])
    outs.sort(key=lambda o: o["rank"])
    assert sum((o["ids"] for o in outs), []) == whole["ids"] == list(range(B * world))
    assert np.array_equal(np.array(sum((o["stats"] for o in outs), [])), np.array(whole["stats"]))
    four = _bench("--gpus", "4", "--batch", "64", "--allow-oversubscribe")
    assert four["n_gpus"] == 4 and four["oversubscribed"] is True and len(four["per_rank_value"]) == 4 and four["config"]["global_batch"] == 256
    assert four["value"] > 0 and four["rehearsal"] is False


def test_bench_refuses_more_ranks_than_gpus():
    """A scaling line can only come from one rank per distinct GPU: without --allow-oversubscribe a launch with more ranks
    than visible devices stops before it measures anything."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with a single GPU")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--repeats", "2",
                        "--batch", "64", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode != 0 and "refusing to oversubscribe" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
