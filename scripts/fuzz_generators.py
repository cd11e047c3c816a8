"""Randomised parity sweep of the HOST ideal generators (no GPU needed): random distribution strings and seeds, the first
ideals of bbx_gen_* (the libstdc++ <random> restatement in bbx_ideals.cpp) against the oracle's generator, term by term;
and parse_ideal_string(format_ideal(F)) == F.      python scripts/fuzz_generators.py [ROUNDS] [SEED]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgroebner_amd.ideals import parse_ideal_dist, format_ideal, parse_ideal_string
from oracle import ffi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bo = ffi.load("bo")
ok = 0
for it in range(rounds):
    n = rng.randint(1, 8)
    d = rng.randint(1, 12 if n <= 3 else (6 if n <= 5 else 4)); s = rng.randint(1, 12)
    if rng.random() < 0.6:
        dist = "%d-%d-%d-%s" % (n, d, s, rng.choice(["uniform", "weighted", "maximum"]))
    else:
        lam = rng.choice(["0.1", "0.5", "1.0", "2.5", "7.0", "11.9", "12.0", "16", "30.5"])
        dist = "%d-%d-%d-%s-%s" % (n, d, s, lam, rng.choice(["uniform", "weighted", "maximum"]))
    for flag in ("consts", "homog", "pure"):
        if rng.random() < 0.2:
            dist += "-" + flag
    seed = rng.choice([0, 1, 123, 2147483646, 2147483647, rng.randint(0, 2 ** 31)])
    try:
        og = ffi.Generator(bo, dist)
    except ValueError:
        try:
            parse_ideal_dist(dist)
            print("MISMATCH %s: the oracle rejects the string, the library accepts it" % dist); sys.exit(1)
        except Exception:
            continue
    g = parse_ideal_dist(dist); g.seed(seed); og.seed(seed)
    for i in range(6):
        try:
            F = next(g)
        except Exception as ex:
            Fo = og.next()
            if Fo is not None and len(Fo) == s:
                print("MISMATCH %s seed %d ideal %d: library fails (%s), oracle delivers" % (dist, seed, i, str(ex)[:60])); sys.exit(1)
            break
        Fo = og.next()
        if Fo is None or len(Fo) != len(F):
            print("MISMATCH %s seed %d ideal %d: oracle fails / short, library delivers" % (dist, seed, i)); sys.exit(1)
        nv = len(F[0][0][1])
        norm = lambda I: [[(int(c), tuple(int(x) for x in e[:nv])) for c, e in f] for f in I]
        if norm(F) != norm(Fo):
            print("MISMATCH %s seed %d ideal %d" % (dist, seed, i)); sys.exit(1)
        if norm(parse_ideal_string(format_ideal(F))) != norm(F):
            print("MISMATCH %s seed %d ideal %d: text round trip" % (dist, seed, i)); sys.exit(1)
    ok += 1
print("fuzz_generators: %d rounds (%d distributions exercised), no mismatch" % (rounds, ok))
