"""Randomised parity sweep of the strategy runs (test infrastructure, GPU box): buchberger(F, strategy) statistics — zero
reductions, non-zero reductions, polynomial additions (scripts/make_strat.cpp of the reference) — for random lists of ideals
(random binomial, random dense, cyclic) under all nine selection strategies, against the CPU restatement.
    python scripts/fuzz_strategies.py [ROUNDS] [SEED]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgroebner_amd import strategy_stats
from oracle import ffi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bo = ffi.load("bo")
STRATS = ["first", "degree", "normal", "sugar", "last", "codegree", "strange", "spice", "random"]
t0 = time.time(); cases = 0
for it in range(rounds):
    kind = rng.choice(["binom3", "binom3", "binomN", "dense", "mixed"])
    dists = {"binom3": ["3-%d-%d-%s" % (rng.randint(3, 12), rng.randint(2, 8), rng.choice(["uniform", "weighted", "maximum"]))],
             "binomN": ["%d-%d-%d-%s" % (rng.randint(4, 6), rng.randint(2, 4), rng.randint(2, 4), rng.choice(["uniform", "weighted"]))],
             "dense": ["%d-%d-%d-%s-uniform" % (rng.randint(2, 4), rng.randint(2, 3), rng.randint(2, 3), rng.choice(["0.3", "0.5"]))],
             "mixed": ["3-8-5-weighted", "3-3-3-0.5-uniform", "4-3-3-uniform"]}[kind]
    ideals = []
    for dist in dists:
        g = bo.generator(dist); g.seed(rng.randint(0, 10 ** 6))
        ideals += [g.next() for _ in range(rng.randint(1, 4))]
    if kind == "mixed" and rng.random() < 0.5:
        ideals.append(bo.cyclic(4))
    trim = [[[(c, e[:7]) for c, e in f] for f in F] for F in ideals]
    strategy = rng.choice(STRATS)
    seed = rng.choice([5, -7, 123456]) if strategy == "random" else None
    # the oracle first, with a cap on the work: strategies that explode on an ideal are skipped for that round
    want = []
    for F in trim:
        _, st = bo.buchberger(F, selection=strategy, want_basis=False, seed=seed)
        want.append([st["zero_reductions"], st["nonzero_reductions"], st["polynomial_additions"]])
    if max(w[2] for w in want) > 200000:
        print("skip (%s explodes on %s)" % (strategy, dists)); continue
    try:
        got = strategy_stats(trim, strategy, seed=seed)
    except Exception as ex:
        print("ERROR %s %s: %s" % (dists, strategy, str(ex)[:200])); sys.exit(1)
    for n in range(len(trim)):
        if got[n].tolist() != want[n]:
            print("MISMATCH %s strategy=%s seed=%s ideal %d: device %s oracle %s" % (dists, strategy, seed, n, got[n].tolist(), want[n])); sys.exit(1)
    cases += len(trim)
    print("ok %-40s %-9s %d ideals, additions %s" % (",".join(dists), strategy, len(trim), [w[2] for w in want]))
print("fuzz_strategies: %d rounds, %d ideal runs, %.0f s, no mismatch" % (rounds, cases, time.time() - t0))
