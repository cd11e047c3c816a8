"""A short run of the host-generator fuzz (scripts/fuzz_generators.py: random distribution strings and seeds, the library's
ideal generators and text format against the oracle) as part of the CPU suite."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_generators_against_oracle_on_random_distribution_strings():
    for seed in ("7", "8"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_generators.py"), "120", seed], cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0 and b"no mismatch" in p.stdout, p.stdout.decode()[-1500:]
