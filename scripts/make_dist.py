#!/usr/bin/env python3
"""Generate the CSV file of ideals sampled from a distribution: the input of make_strat.

    python scripts/make_dist.py <distribution> <samples> [<seed>]

writes data/stats/<distribution>/<distribution>.csv in the format of the reference's scripts/make_dist.m2:50-77 (header
"Ideal", one ideal per line, polynomials joined by "|").  The reference draws its samples with Macaulay2's generator;
here they come from the C++-compatible generators of libbbx (the streams of deepgroebner/ideals.cpp), so the ideals
differ from a Macaulay2 run with the same seed while the distribution and the file format are the same.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv):
    if len(argv) < 3:
        print("Usage: make_dist <distribution> <samples> <seed>")
        return 1
    dist, samples = argv[1], int(argv[2])
    directory = "data/stats/%s/" % dist
    out_name = directory + dist + ".csv"
    if os.path.exists(out_name):
        print("Output file %s already exists. Delete or move it first." % out_name)
        return 3
    from deepgroebner_amd import format_ideal, parse_ideal_dist
    gen = parse_ideal_dist(dist)
    if len(argv) > 3:
        gen.seed(int(argv[3]))
    os.makedirs(directory, exist_ok=True)
    with open(out_name, "w") as out:
        out.write("Ideal\n")
        for _ in range(samples):
            out.write(format_ideal(next(gen)) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
