#!/usr/bin/env python3
"""Generate the CSV file of strategy performance for sampled ideals, every ideal of the file at once on the GPU.

Same command line, files, messages and exit codes as the reference's scripts/make_strat.cpp:22-72:

    python scripts/make_strat.py <distribution> <strategy> [<seed>]

reads  data/stats/<distribution>/<distribution>.csv  (header line, then one ideal per line), writes
data/stats/<distribution>/<distribution>_<strategy>.csv (…_random_<seed>.csv for seeded random runs) with the columns
ZeroReductions,NonzeroReductions,PolynomialAdditions.  Strategy names unknown to the reference's map select First
there (std::map::operator[]); here as well.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

STRATEGIES = ("first", "degree", "normal", "sugar", "random", "last", "codegree", "strange", "spice")


def main(argv):
    if len(argv) < 3:
        print("Usage: make_strat <distribution> <strategy> <seed>")
        return 1
    dist, strat = argv[1], argv[2]
    seed = int(argv[3]) if len(argv) > 3 else None
    in_name = "data/stats/%s/%s.csv" % (dist, dist)
    if not os.path.isfile(in_name):
        print("No distribution file found. Run scripts/make_dist.py first.")
        return 2
    out_name = "data/stats/%s/%s_%s.csv" % (dist, dist, strat)
    if seed is not None and strat == "random":
        out_name = "data/stats/%s/%s_%s_%s.csv" % (dist, dist, strat, argv[3])
    if os.path.exists(out_name):
        print("Output file %s already exists. Delete or move it first." % out_name)
        return 3

    from deepgroebner_amd import parse_ideal_string, strategy_stats
    with open(in_name) as f:
        f.readline()                                   # column name
        ideals = [parse_ideal_string(line) for line in f if line.strip()]
    chunk = int(os.environ.get("BBX_MAKE_STRAT_BATCH", "16384"))
    with open(out_name, "w") as out:
        out.write("ZeroReductions,NonzeroReductions,PolynomialAdditions\n")
        for lo in range(0, len(ideals), chunk):
            st = strategy_stats(ideals[lo:lo + chunk], strat if strat in STRATEGIES else "first", seed=seed)
            for z, nz, a in st:
                out.write("%d,%d,%d\n" % (z, nz, a))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
