"""Known answers of the reference's own test-suite, transliterated as data
(tests/test_polynomials.cpp, tests/test_buchberger.cpp, tests/test_ideals.cpp and the
GF(32003)/grevlex cases of tests/test_buchberger.py) and checked on the C restatement —
and on the compiled reference too where oracle/_ref is available."""
import numpy as np
import pytest

from oracle import ffi


@pytest.fixture(params=["bo", "ref"])
def lib(request):
    import os
    if request.param == "ref" and not ffi.available("ref") and not os.path.isdir("/root/reference"):
        pytest.skip("oracle/_ref unavailable")
    return ffi.load(request.param)


def P(*terms):
    """terms (c, exps) -> normalised python polynomial as the checkers return it."""
    return [(c % 32003, tuple(list(e) + [0] * (8 - len(e)))) for c, e in terms]


def canon(poly):
    return [(c % 32003, tuple(e)) for c, e in poly]


# ---- tests/test_polynomials.cpp:5-49
def test_coefficient(lib):
    assert lib.coef("norm", 2045) == 2045 and lib.coef("norm", -2) == 32001 and lib.coef("norm", 32008) == 5
    assert lib.coef("add", 3, 10) == 13 and lib.coef("sub", 10, 3) == 7
    assert lib.coef("mul", 3, 10) == 30 and lib.coef("mul", 3, -2) == 31997
    assert lib.coef("div", 3, 10) == 28803 and lib.coef("div", 23002, 32001) == 20502
    assert lib.coef("div", 12000, 4) == 3000 and lib.coef("div", 12345, 1) == 12345


# ---- tests/test_polynomials.cpp:76-86
def test_monomial_order(lib):
    m1, m2, m3 = [1] * 8, [0] * 7 + [9], [0, 0, 0, 0, 2, 2, 2, 2]
    assert lib.mono_gt(m1, m3) and lib.mono_gt(m2, m1) and lib.mono_gt(m2, m3)
    assert not lib.mono_gt(m3, m1) and not lib.mono_gt(m1, m2) and not lib.mono_gt(m3, m2)


# ---- tests/test_polynomials.cpp:146-217
def test_polynomial_ctor_sorts(lib):
    p = lib.polylist([[(1, [1] * 8), (3, [0, 0, 0, 0, 1, 1, 1, 1]), (9, [1, 1, 2, 2, 3, 4, 1, 1]), (1, [0] * 8)]]).get(0)
    assert p[0] == (9, (1, 1, 2, 2, 3, 4, 1, 1))


def test_polynomial_add_sub(lib):
    p1 = [(1, (1, 2, 1)), (3, (1, 0, 1)), (7, (0, 0, 0))]
    p2 = [(9, (7, 0, 0)), (-3, (1, 0, 1)), (1, (1, 0, 0))]
    p3 = [(9, (7, 0, 0)), (1, (1, 2, 1)), (1, (1, 0, 0)), (7, (0, 0, 0))]
    assert lib.binop("poly_add", p1, p2) == P(*p3)
    assert lib.binop("poly_sub", p3, p2) == P(*p1)
    assert lib.binop("poly_sub", p3, p1) == P(*p2)
    assert lib.binop("poly_sub", p1, p1) == []


def test_polynomial_multiply(lib):
    p1 = [(1, (1, 2, 0)), (1, (0, 1, 1)), (1, (0, 0, 0))]
    p2 = [(1, (1, 1, 1)), (1, (1, 0, 0))]
    p3 = [(1, (2, 3, 1)), (1, (1, 2, 2)), (1, (2, 2, 0)), (2, (1, 1, 1)), (1, (1, 0, 0))]
    assert lib.binop("poly_mul", p1, p2) == P(*p3)


# ---- tests/test_polynomials.cpp:219-246
@pytest.mark.parametrize("s,want", [
    ("a^2*b+c*d", [(1, (2, 1, 0, 0)), (1, (0, 0, 1, 1))]),
    ("413*a^2*b^5*c+32*d^2-5", [(413, (2, 5, 1, 0)), (32, (0, 0, 0, 2)), (-5, (0, 0, 0, 0))]),
    ("3", [(3, ())]),
    ("12*a^2-b*c+13*d", [(12, (2, 0, 0, 0)), (-1, (0, 1, 1, 0)), (13, (0, 0, 0, 1))]),
])
def test_parse_polynomial(lib, s, want):
    assert lib.parse_polynomial(s) == lib.polylist([want]).get(0)


# ---- tests/test_buchberger.cpp:9-55 and tests/test_buchberger.py:15-27 (ring R1)
@pytest.mark.parametrize("f,g,s", [
    ([(1, (1, 2, 1)), (3, (1, 0, 1)), (7, (0, 0, 0))], [(9, (7, 0, 0)), (-3, (1, 0, 1)), (1, (1, 0, 0))],
     [(3, (7, 0, 1)), (7, (6, 0, 0)), (10668, (1, 2, 2)), (28447, (1, 2, 1))]),
    ([(1, (2, 0)), (1, (1, 1))], [(1, (0, 2)), (1, (1, 1))], []),
    ([(1, (3, 2)), (-1, (2, 3))], [(1, (4, 1)), (1, (0, 2))], [(-1, (3, 3)), (-1, (0, 3))]),
    ([(1, (2, 0)), (1, (0, 3))], [(1, (1, 2)), (1, (1, 0)), (1, (0, 0))], [(1, (3, 0)), (-1, (1, 1)), (-1, (0, 1))]),
])
def test_spoly(lib, f, g, s):
    assert lib.spoly(f, g) == P(*s)


# ---- tests/test_buchberger.cpp:58-77, tests/test_buchberger.py:30-45 (first case)
def test_reduce(lib):
    g = [(1, (3, 1, 2)), (1, (2, 0, 1))]
    F = [[(1, (2, 0, 0)), (1, (0, 1, 0))], [(1, (1, 1, 1)), (1, (0, 0, 1))], [(1, (1, 0, 2)), (1, (0, 2, 0))]]
    assert lib.reduce(g, F)[0] == P((1, (0, 1, 2)), (-1, (0, 1, 1)))
    g = [(1, (5, 10, 4)), (22982, (3, 1, 2))]
    F = [[(1, (5, 12, 0)), (25797, (1, 5, 2))], [(1, (1, 3, 1)), (27630, (2, 1, 0))], [(1, (1, 9, 1)), (8749, (2, 0, 0))]]
    r, steps = lib.reduce(g, F)
    assert r == P((2065, (9, 2, 0)), (22982, (3, 1, 2))) and steps == 4


# ---- tests/test_buchberger.cpp:80-102, tests/test_buchberger.py:110-153 (ring R1 cases)
@pytest.mark.parametrize("elim", ["none", "lcm", "gebauermoeller"])
def test_update_empty(lib, elim):
    f = [(1, (2, 0)), (1, (1, 1)), (2, (0, 0))]
    G, Pn = lib.update([], [], f, elim)
    assert G == [P(*f)] and Pn == []


@pytest.mark.parametrize("elim,want", [("none", [(0, 1)]), ("lcm", []), ("gebauermoeller", [])])
def test_update_1(lib, elim, want):
    G = [[(1, (1, 2, 0)), (2, (1, 0, 1)), (-1, (1, 0, 0))]]            # x*y^2 + 2xz - x
    f = [(1, (0, 0, 5)), (2, (2, 1, 1)), (1, (1, 0, 1))]                # z^5 + 2x^2yz + xz
    assert lib.update(G, [], f, elim)[1] == want


@pytest.mark.parametrize("elim,want", [
    ("none", [(0, 2), (0, 3), (1, 3), (2, 3)]), ("lcm", [(0, 2), (0, 3), (1, 3)]), ("gebauermoeller", [(0, 2)])])
def test_update_5(lib, elim, want):
    G = [[(1, (1, 2, 0)), (2, (0, 0, 1))], [(1, (1, 0, 2)), (-1, (0, 2, 0)), (-1, (0, 0, 1))], [(1, (1, 0, 0)), (3, (0, 0, 0))]]
    f = [(1, (0, 2, 3)), (-1, (0, 2, 0)), (4, (0, 0, 4)), (1, (0, 0, 2))]
    assert lib.update(G, [(0, 2)], f, elim)[1] == want


# ---- tests/test_buchberger.cpp:105-133
def test_minimalize_interreduce(lib):
    G = [[(1, (1, 2, 0)), (1, (0, 0, 1))], [(1, (1, 0, 1)), (3, (0, 1, 0))], [(1, (2, 0, 0)), (1, (0, 1, 1))],
         [(-3, (0, 3, 0)), (1, (0, 2, 0))], [(-9, (0, 1, 0)), (-1, (0, 0, 3))], [(1, (0, 0, 8)), (243, (0, 0, 1))]]
    Gmin = [[(1, (1, 0, 1)), (3, (0, 1, 0))], [(1, (2, 0, 0)), (1, (0, 1, 1))], [(-1, (0, 0, 3)), (-9, (0, 1, 0))],
            [(-3, (0, 3, 0)), (1, (0, 2, 0))], [(1, (1, 2, 0)), (1, (0, 0, 1))]]
    assert lib.minimalize(G) == [P(*p) for p in Gmin]
    Gred = [[(1, (1, 0, 1)), (3, (0, 1, 0))], [(1, (2, 0, 0)), (1, (0, 1, 1))], [(1, (0, 0, 3)), (9, (0, 1, 0))],
            [(1, (0, 3, 0)), (21335, (0, 2, 0))], [(1, (1, 2, 0)), (1, (0, 0, 1))]]
    assert lib.interreduce(Gmin) == [P(*p) for p in Gred]


# ---- tests/test_buchberger.py:226-243 (ring R1 rows): full Buchberger -> reduced GB
@pytest.mark.parametrize("elim", ["none", "lcm", "gebauermoeller"])
def test_buchberger_small(lib, elim):
    F = [[(1, (0, 1, 0)), (-1, (2, 0, 0))], [(1, (0, 0, 1)), (-1, (3, 0, 0))]]           # y - x^2, z - x^3
    G, _ = lib.buchberger(F, elimination=elim)
    want = [[(1, (0, 2, 0)), (-1, (1, 0, 1))], [(1, (1, 1, 0)), (-1, (0, 0, 1))], [(1, (2, 0, 0)), (-1, (0, 1, 0))]]
    assert G == [P(*p) for p in want]
    G, _ = lib.buchberger(lib.cyclic(3), elimination=elim)
    want = [[(1, (1, 0, 0)), (1, (0, 1, 0)), (1, (0, 0, 1))], [(1, (0, 2, 0)), (1, (0, 1, 1)), (1, (0, 0, 2))], [(1, (0, 0, 3)), (-1, (0, 0, 0))]]
    assert G == [P(*p) for p in want]


# ---- tests/test_ideals.cpp:9-144
def test_cyclic3(lib):
    want = [[(1, (1, 0, 0)), (1, (0, 1, 0)), (1, (0, 0, 1))], [(1, (1, 1, 0)), (1, (0, 1, 1)), (1, (1, 0, 1))], [(1, (1, 1, 1)), (-1, (0, 0, 0))]]
    assert lib.cyclic(3) == [lib.polylist([p]).get(0) for p in want]


def test_basis_order(lib):
    assert lib.basis(3, 0)[:, :3].tolist() == [[0, 0, 0]]
    assert lib.basis(4, 1)[:, :4].tolist() == [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    assert lib.basis(3, 2)[:, :3].tolist() == [[2, 0, 0], [1, 1, 0], [1, 0, 1], [0, 2, 0], [0, 1, 1], [0, 0, 2]]
    assert lib.basis(3, 3)[:, :3].tolist() == [[3, 0, 0], [2, 1, 0], [2, 0, 1], [1, 2, 0], [1, 1, 1], [1, 0, 2], [0, 3, 0], [0, 2, 1], [0, 1, 2], [0, 0, 3]]


@pytest.mark.parametrize("n,d,dist,consts,want", [
    (3, 1, "weighted", False, [0.0, 1.0]), (3, 1, "weighted", True, [0.5, 0.5]), (3, 1, "uniform", True, [0.25, 0.75]),
    (3, 5, "weighted", False, [0.0, 0.2, 0.2, 0.2, 0.2, 0.2]), (3, 5, "weighted", True, [1.0 / 6] * 6),
    (3, 5, "uniform", True, [1.0 / 56, 3.0 / 56, 6.0 / 56, 10.0 / 56, 15.0 / 56, 21.0 / 56]),
    (3, 3, "maximum", True, [0.5, 0.0, 0.0, 0.5]), (3, 3, "maximum", False, [0.0, 0.0, 0.0, 1.0]),
    (3, 3, "uniform", False, [0.0, 3.0 / 19, 6.0 / 19, 10.0 / 19]), (3, 3, "weighted", False, [0.0, 1.0 / 3, 1.0 / 3, 1.0 / 3]),
])
def test_degree_distribution(lib, n, d, dist, consts, want):
    assert lib.degree_distribution(n, d, dist, consts) == want


def test_seed123_generators(lib):
    g = lib.generator("3-5-5-uniform"); g.seed(123)
    want = [[(1, (0, 1, 4)), (31, (0, 3, 1))], [(1, (3, 1, 1)), (16013, (3, 0, 2))], [(1, (2, 2, 0)), (18427, (1, 0, 1))],
            [(1, (2, 0, 3)), (15139, (2, 1, 1))], [(1, (0, 3, 2)), (5374, (1, 0, 2))]]
    assert g.next() == [P(*p) for p in want]
    g = lib.generator("3-5-5-0.5-uniform"); g.seed(123)
    want = [[(1, (0, 1, 3)), (22264, (0, 0, 4))], [(1, (1, 1, 1)), (1541, (0, 0, 2))],
            [(1, (2, 2, 1)), (15981, (0, 2, 1)), (7023, (0, 0, 1))],
            [(1, (1, 4, 0)), (10365, (0, 5, 0)), (5289, (1, 3, 0)), (13942, (1, 1, 0))], [(1, (3, 1, 0)), (11636, (1, 1, 0))]]
    assert g.next() == [P(*p) for p in want]


# ---- episode-level known answers, tests/test_buchberger.py:270-296 (rewards='reductions')
def _grevlex_key(e):
    return (int(sum(e)), tuple(-int(x) for x in reversed(e)))


def _select(env, strategy):
    pairs = env.pairs()
    lm = [env.poly(i)[1][0] for i in range(env.nG)]
    def key(p):
        i, j = int(p[0]), int(p[1])
        l = np.maximum(lm[i], lm[j])
        k = []
        for s in strategy:
            if s == "first":
                k += [j, i]
            elif s == "degree":
                k.append(int(l.sum()))
            elif s == "normal":
                k.append(_grevlex_key(l))
        return tuple(k)
    return min(range(len(pairs)), key=lambda r: key(pairs[r]))


def _episode(env, strategy):
    env.reset()
    total = 0.0
    while env.nP:
        total += env.step(_select(env, strategy))
    return total


@pytest.mark.parametrize("strategy", [["first"], ["degree", "first"], ["normal", "first"]])
def test_episode_katsura(lib, strategy):
    def var(i):
        e = [0] * 5; e[i] = 1; return tuple(e)
    def sq(i):
        e = [0] * 5; e[i] = 2; return tuple(e)
    def mul(i, j):
        e = [0] * 5; e[i] += 1; e[j] += 1; return tuple(e)
    one = (0,) * 5
    a, b, c, d, e_ = range(5)
    F = [[(1, var(a)), (2, var(b)), (2, var(c)), (2, var(d)), (2, var(e_)), (-1, one)],
         [(1, sq(a)), (2, sq(b)), (2, sq(c)), (2, sq(d)), (2, sq(e_)), (-1, var(a))],
         [(2, mul(a, b)), (2, mul(b, c)), (2, mul(c, d)), (2, mul(d, e_)), (-1, var(b))],
         [(1, sq(b)), (2, mul(a, c)), (2, mul(b, d)), (2, mul(c, e_)), (-1, var(c))],
         [(2, mul(b, c)), (2, mul(a, d)), (2, mul(b, e_)), (-1, var(d))]]
    env = lib.env(fixed=F, rewards="reductions")
    assert _episode(env, strategy) == -28


@pytest.mark.parametrize("elim,reward", [("none", -45), ("lcm", -35), ("gebauermoeller", -11)])
def test_episode_cyclic4(lib, elim, reward):
    env = lib.env(fixed=lib.cyclic(4), elimination=elim, rewards="reductions")
    assert _episode(env, ["normal", "first"]) == reward


# ---- docstring transcripts of buchberger.py:261-318 / 470-506 are for the numpy RNG; the C++ stream's
# first seed-123 reset is pinned by SURVEY 8c(4):
def test_seed123_reset_obs(lib):
    env = lib.env("3-20-10-weighted"); env.seed(123); env.reset()
    obs = env.obs(2)
    assert obs.shape == (19, 12) and obs[0].tolist() == [11, 6, 3, 9, 7, 2, 7, 0, 5, 2, 0, 2]
    assert env.value("degree", 0.99) == -122.89915880550187
    assert env.value("env", 0.99) == env.value("first", 0.99) == -216.24918322453357


# ---- FixedIdealGenerator nvars quirk (ideals.cpp:146-154): max variable INDEX, so cyclic-4 -> 3
def test_fixed_nvars_quirk(lib):
    assert lib.generator("cyclic-4").nvars() == 3 and lib.generator("cyclic-7").nvars() == 6
