"""Fixed-seed slices of the randomised parity drivers (scripts/fuzz_*.py: random distribution strings, batch sizes, horizons,
k, eliminations, reward modes, kernel capacities, agents, policies — every case against the oracle) as part of the GPU suite.
A line with CAPACITY, MISMATCH or ERROR fails the test: capacities grow on demand, so an environment the oracle could
finish must finish on the device too."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,rounds,seed,env", [
    ("fuzz_parity.py", 600, 301, {}),
    ("fuzz_parity.py", 100, 302, {"FUZZ_LONG": "1"}),          # ten times the horizons: long polynomials, grown records
    ("fuzz_parity.py", 20, 303, {"FUZZ_LARGE": "1"}),          # batches of 1024 / 4096 environments
    ("fuzz_wide.py", 100, 304, {}),
    ("fuzz_gym.py", 400, 305, {}),
    ("fuzz_value.py", 150, 306, {}),
    ("fuzz_policy.py", 60, 307, {}),
    ("fuzz_strategies.py", 150, 308, {}),
    ("fuzz_sessions.py", 300, 309, {}),                        # persistent sessions against a twin without them
])
def test_fuzz_slice(script, rounds, seed, env):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(rounds), str(seed)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, **env))
    out = p.stdout.decode(errors="replace")
    bad = [ln for ln in out.splitlines() if ln.startswith(("CAPACITY", "MISMATCH", "ERROR")) or ln.lower().startswith("capacity")]
    assert p.returncode == 0 and not bad and "no mismatch" in out, out[-3000:]
