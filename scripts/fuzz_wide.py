"""Randomised parity sweep of the wide class (test infrastructure, GPU box): cyclic-n under the counter-hash agent with random
workgroup widths, LDS capacities (which decide the merge tiers), batch sizes and horizons, lean and accounting variants,
against the CPU restatement.     python scripts/fuzz_wide.py [ROUNDS] [SEED]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi
from test_gpu_parity import fnv64, _state_words

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bo = ffi.load("bo")
torch.cuda.init()
t0 = time.time()
for it in range(rounds):
    n = rng.choice([4, 5, 5, 6, 6, 7])
    dist = "cyclic-%d" % n
    T = {4: 40, 5: 120, 6: rng.choice([60, 150]), 7: rng.choice([30, 70])}[n]
    B = rng.choice([1, 2, 5, 16])
    if n <= 5 and rng.random() < 0.25:                     # more workgroups than CUs: the launch is two kernels (BbxParams::wide_tail)
        B, T = rng.choice([260, 300, 520]), rng.choice([15, 40])
    k = rng.choice([1, 2])
    caps = {}
    if rng.random() < 0.6:
        caps["wide_waves"] = rng.choice([1, 2, 3, 4, 5, 8])
    if rng.random() < 0.6:
        caps["wide_lds_terms"] = rng.choice([40, 64, 128, 256, 1000, 3000])
    lean = rng.random() < 0.6
    want = bo.run_random_many(dist, k, [0] * B, range(B), T, True, 0)
    env = VecLeadMonomialsEnv(dist, batch=B, k=k, caps=caps or None)
    env.seed_agent(np.arange(B)); env.reset()
    if lean:
        env.accounting(False)
    try:
        env.rollout("random", T, auto_reset=True)
    except Exception as ex:
        print("ERROR %s k=%d B=%d T=%d caps=%s lean=%s: %s" % (dist, k, B, T, caps, lean, str(ex)[:200])); sys.exit(1)
    st = env.stats()
    for key, col in (("steps", 0), ("additions", 1), ("episodes", 2), ("zero_reductions", 3), ("nG", 7)):
        w = np.array([r[key] for r in want])
        if not np.array_equal(st[:, col], w):
            e = int(np.flatnonzero(st[:, col] != w)[0])
            print("MISMATCH %s k=%d B=%d T=%d caps=%s lean=%s: %s of env %d: device %d oracle %d" % (dist, k, B, T, caps, lean, key, e, st[e, col], w[e])); sys.exit(1)
    if not lean:
        w = np.array([r["bytes"] for r in want])
        if not np.array_equal(st[:, 6], w):
            print("MISMATCH %s caps=%s: algorithmic bytes" % (dist, caps)); sys.exit(1)
    for e in (range(B) if B <= 16 else list(range(0, B, 41)) + [int(st[:, 1].argmax())]):
        basis, pairs, order = env.state(e)
        if fnv64(_state_words(basis, pairs, order)) != want[e]["state_hash"]:
            print("MISMATCH %s k=%d B=%d T=%d caps=%s lean=%s: final state of env %d" % (dist, k, B, T, caps, lean, e)); sys.exit(1)
    print("ok %-9s k=%d B=%-3d T=%-3d caps=%-48s lean=%d additions %d" % (dist, k, B, T, caps, lean, st[:, 1].sum()))
    del env
print("fuzz_wide: %d rounds, %.0f s, no mismatch" % (rounds, time.time() - t0))
