"""The contract of bench.py's output line (what the driver parses): exactly one line on stdout, the fields of the task's
schema, the roofline and cpu_baseline objects, and the consistency the numbers must have among themselves."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_fields_and_consistency():
    d = _run(["--steps", "16", "--warmup", "4", "--repeats", "6", "--cpu-sample-envs", "16"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "repeats", "preroll_steps"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["warmup"] == 4 and d["repeats"] == 6 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    B = d["config"]["global_batch"]
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6          # value = batch / time per batch step
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is not None and r["traffic"] > 0 and r["issue_bound"]["peak"] == 1.0
    assert 0 < r["kernel_ms_per_launch"] <= d["ms_per_step"] * d["steps"] * 1.05          # the kernel is inside the timed launch
    # the HBM figure is the one SURVEY 8d prescribes; the line also says what really binds, and which kernel the counters are of
    assert r["binding"] == "issue" and r["traffic_kernel"] and isinstance(r["traffic_kernel_is_timed_kernel"], bool)
    assert r["issue_bound"]["kernel"] == r["traffic_kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["additions_match_device"] is True
    ca = d["cpu_baseline_all_cores"]                                                       # one environment stream per host core
    assert ca["kind"] == c["kind"] and ca["cores"] >= 1 and ca["threads"] == ca["cores"] and ca["nproc"] >= ca["cores"] and ca["value"] > 0
    assert ca["additions_match_device"] in (True, None)
    ss = d["session_stats"]                                                                # the sustained figure's evidence: nothing left the class
    assert set(ss["timed_region"]) == {"sessions", "joined", "later_kernel_steps", "kernels", "spills"} and ss["timed_region"]["spills"] >= 0
    assert d["long_launch"]["launches"] >= 8 and d["long_launch"]["value"] > 0 and d["rehearsal"] is False


@pytest.mark.gpu
def test_bench_spawns_its_ranks_and_prints_one_line():
    d = _run(["--gpus", "2", "--steps", "8", "--warmup", "2", "--repeats", "4", "--no-cpu-baseline", "--allow-oversubscribe"])   # two ranks on the one device
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * 4096 and d["value"] > 0
    assert d["oversubscribed"] is True and len(d["per_rank_value"]) == 2 and all(v > 0 for v in d["per_rank_value"])
