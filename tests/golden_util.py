"""Helpers shared by the golden-vector tests (CPU oracle and HIP path)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def meta():
    with open(os.path.join(GOLD, "values.json")) as f:
        return json.load(f)


def load_trace(name):
    return np.load(os.path.join(GOLD, "trace_%s.npz" % name))


def trace_names():
    return sorted(meta()["traces"].keys())


def assert_trace_equal(got, gold, e, label=""):
    for key in ("action", "reward", "nP", "nG", "done", "obs_hash", "pairs_hash", "newpoly_hash",
                "init", "init_hash", "final_pairs", "final_order", "final_basis", "final_obs"):
        g = gold["e%d_%s" % (e, key)]
        h = got[key]
        assert h.shape == g.shape, "%s env %d: %s shape %s != %s" % (label, e, key, h.shape, g.shape)
        if not np.array_equal(h, g):
            bad = int(np.flatnonzero(np.asarray(h).ravel() != np.asarray(g).ravel())[0])
            raise AssertionError("%s env %d: %s first differs at %d: %r != %r" % (label, e, key, bad, h.ravel()[bad], g.ravel()[bad]))
