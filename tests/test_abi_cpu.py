"""CPU-side checks of the product library: it loads without a GPU, exports every symbol that
include/bbx.h declares, its host-side ideal generators reproduce the reference's seeded streams
(goldens), and it fails loudly instead of falling back when no device is present."""
import os
import re

import numpy as np
import pytest

import __graft_entry__ as graft
from tests.golden_util import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ffi_():
    graft.build()
    from deepgroebner_amd import _ffi
    return _ffi


def test_exports_every_declared_symbol(ffi_):
    hdr = open(os.path.join(ROOT, "include", "bbx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bbx_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(ffi_.SIGNATURES), (declared ^ set(ffi_.SIGNATURES))
    dll = ffi_.lib()
    for name in declared:
        assert hasattr(dll, name), name


def test_generators_match_reference_streams(ffi_):
    from deepgroebner_amd import ideals
    from oracle.trace import flat_ideal
    gold = np.load(os.path.join(GOLD, "generators.npz"))
    keys = [k for k in gold.files if not k.endswith("|nvars")]
    seen = {}
    for key in sorted(keys, key=lambda s: (s.split("|")[0], int(s.split("|")[1]), int(s.split("|")[2]))):
        dist, seed, draw = key.split("|")
        if (dist, seed) not in seen:
            g = ideals.parse_ideal_dist(dist)
            g.seed(int(seed))
            seen[(dist, seed)] = g
            assert g.nvars == int(gold["%s|nvars" % dist][0])
        assert np.array_equal(flat_ideal(next(seen[(dist, seed)])), gold[key]), key


def test_python_mirrors(ffi_):
    from deepgroebner_amd import ideals
    assert ideals.basis(3, 2) == [(2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2)]
    assert ideals.degree_distribution(3, 3, "uniform", False) == [0.0, 3.0 / 19, 6.0 / 19, 10.0 / 19]
    assert ideals.cyclic(3)[2] == [(1, (1, 1, 1)), (32002, (0, 0, 0))]
    g = ideals.RandomBinomialIdealGenerator(3, 5, 5)
    g.seed(123)
    assert next(g)[0] == [(1, (0, 1, 4, 0, 0, 0, 0, 0)), (31, (0, 3, 1, 0, 0, 0, 0, 0))]


def test_bad_distribution_string(ffi_):
    from deepgroebner_amd import ideals
    with pytest.raises(ffi_.BbxError):
        ideals.parse_ideal_dist("nonsense")


def test_agent_hash_matches_oracle_definition(ffi_):
    from oracle import ffi as offi
    for seed, t in ((0, 0), (1, 2), (4095, 255), (0xFFFFFFFF, 0xFFFFFFFF)):
        assert ffi_.lib().bbx_agent_hash(seed, t) == offi.agent_hash(seed, t)
        for rows in (1, 2, 19, 255):
            assert ffi_.lib().bbx_agent_action(seed, t, rows) == offi.agent_action(seed, t, rows) < rows


def test_no_silent_cpu_fallback(ffi_):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from deepgroebner_amd import VecLeadMonomialsEnv
    with pytest.raises(ffi_.BbxError) as ei:
        VecLeadMonomialsEnv("3-20-10-weighted", batch=2)
    assert ei.value.code == -2


# ---- the text format of data/stats/*.csv (host code, no device involved) -------------------------------------
@pytest.mark.parametrize("s,want", [                       # tests/test_polynomials.cpp:219-246 of the reference
    ("a^2*b+c*d", [(1, (2, 1, 0, 0)), (1, (0, 0, 1, 1))]),
    ("413*a^2*b^5*c+32*d^2-5", [(413, (2, 5, 1, 0)), (32, (0, 0, 0, 2)), (32003 - 5, (0, 0, 0, 0))]),
    ("3", [(3, ())]),
    ("12*a^2-b*c+13*d", [(12, (2, 0, 0, 0)), (32003 - 1, (0, 1, 1, 0)), (13, (0, 0, 0, 1))]),
])
def test_parse_polynomial_known_answers(ffi_, s, want):
    from deepgroebner_amd import parse_polynomial
    pad = lambda e: tuple(e) + (0,) * (8 - len(e))          # noqa: E731
    assert parse_polynomial(s) == [(c, pad(e)) for c, e in want]


def test_parse_matches_oracle_and_round_trips(ffi_):
    from oracle import ffi as offi
    from deepgroebner_amd import format_ideal, parse_ideal_dist, parse_ideal_string, parse_polynomial
    bo = offi.load("bo")
    for dist, seed in (("3-20-10-weighted", 1), ("5-10-5-uniform", 2), ("3-6-5-0.5-uniform-consts", 3), ("4-3-6-1.5-maximum", 4)):
        g = parse_ideal_dist(dist); g.seed(seed)
        for _ in range(5):
            F = next(g)
            line = format_ideal(F)
            assert " " not in line and line.count("|") == len(F) - 1
            assert parse_ideal_string(line) == F
            for text, f in zip(line.split("|"), F):
                got = bo.parse_polynomial(text)             # the restated reference parser reads what we write
                assert [(c, tuple(e)) for c, e in got] == f
    # the grammar's corners, settled by the reference's recursion (polynomials.cpp:226-294)
    for s in ("a-a+b", "3a", "--3*a*", "+a", "a*b^2*a", "2*a+3*a", "h^2-1"):
        got = bo.parse_polynomial(s)
        assert parse_polynomial(s) == [(c, tuple(e)) for c, e in got], s


@pytest.mark.parametrize("bad", ["a+i", "a^-2", "32003*a", "a||b", "a*3", "a^b", "", "a-a", "99999999999*a"])
def test_parse_rejects_what_the_reference_leaves_undefined(ffi_, bad):
    from deepgroebner_amd import parse_ideal_string
    with pytest.raises(ffi_.BbxError):
        parse_ideal_string(bad)


def test_policy_shapes_are_checked_on_the_host(ffi_):
    """What the policy kernels are built for is answered without a device: prepared-buffer sizes for the supported shapes,
    BBX_E_UNSUPPORTED (-5) with a message for the others (the Python side then takes its torch path)."""
    dll = ffi_.lib()
    assert dll.bbx_pmlp_prepared_floats(12, 128) == (2 * 6 + 2) * 128 + 4
    assert dll.bbx_pmlp_prepared_floats(65, 128) == -5 and dll.bbx_pmlp_prepared_floats(12, 257) == -5
    n = dll.bbx_pmlp2_prepared_floats(12, 128, 128)          # W1p [12][128] | b1p | A2 [128 x 128, permuted] | b2p | w3p | b3 + pad
    assert n == (2 * 6 + 1) * 128 + 128 * 128 + 2 * 128 + 4
    assert dll.bbx_pmlp2_prepared_floats(12, 64, 7) == (2 * 6 + 1) * 64 + 64 * 64 + 2 * 64 + 4      # (layers are padded to 64 units at least)
    assert dll.bbx_pmlp2_prepared_floats(40, 128, 64) == (2 * 32 + 1) * 128 + 128 * 64 + 2 * 64 + 4
    for bad in ((65, 128, 128), (12, 129, 128), (12, 128, 0)):
        assert dll.bbx_pmlp2_prepared_floats(*bad) == -5
        assert b"two-layer" in dll.bbx_last_error()
    assert dll.bbx_pmlp3_prepared_floats(12, 128, 128, 128) == (4 * 3 + 1) * 128 + 2 * 128 * 128 + 3 * 128 + 4
    assert dll.bbx_pmlp3_prepared_floats(12, 40, 64, 17) == (4 * 3 + 1) * 64 + 2 * 64 * 64 + 3 * 64 + 4
    assert dll.bbx_pmlp3_prepared_floats(12, 40, 100, 17) == (4 * 3 + 1) * 128 + 2 * 128 * 128 + 3 * 128 + 4   # (all three padded to the widest)
    assert dll.bbx_pmlp3_prepared_floats(12, 40, 129, 17) == -5 and b"three-layer" in dll.bbx_last_error()
    assert dll.bbx_graph_replayed(None, None) == -1                                                   # BBX_E_ARG
