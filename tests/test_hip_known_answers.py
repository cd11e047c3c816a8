"""The reference's own unit vectors for its free functions — tests/test_polynomials.cpp:146-217 (Polynomial +, -, *),
tests/test_buchberger.cpp:9-133 and the GF(32003)/grevlex cases of tests/test_buchberger.py:15-243 (spoly, reduce, update
under all three eliminations, minimalize, interreduce, buchberger) — replayed on the HIP path: the very test functions of
tests/test_oracle_known_answers.py (where the vectors live as data) run here against an adapter over
deepgroebner_amd.PolyLists / buchberger, i.e. the kernels of csrc/bbx_algebra.hip through the C ABI (bbx_alg_*)."""
import numpy as np
import pytest

from tests.test_oracle_known_answers import (P, test_buchberger_small, test_minimalize_interreduce, test_polynomial_add_sub,  # noqa: F401
                                             test_polynomial_multiply, test_reduce, test_spoly, test_update_1, test_update_5,
                                             test_update_empty)

pytestmark = pytest.mark.gpu


class HipLib:
    """The surface of oracle.ffi.Lib the imported tests use, on the device."""

    def binop(self, name, f, g):
        from deepgroebner_amd import PolyLists
        L = PolyLists([[f, g]])
        L.binop({"poly_add": "add", "poly_sub": "sub", "poly_mul": "mul", "spoly": "spoly"}[name], (0, 1))
        return L.get(0)[2]

    def spoly(self, f, g):
        return self.binop("spoly", f, g)

    def reduce(self, g, F):
        from deepgroebner_amd import PolyLists
        L = PolyLists([list(F) + [g]])
        steps = L.reduce((len(F), len(F)))
        return L.get(0)[-1], int(steps[0])

    def update(self, G, P_, f, elimination="gebauermoeller"):
        from deepgroebner_amd import PolyLists
        L = PolyLists([list(G) + [f]])
        pairs = L.update([list(P_)], elimination)[0]
        return L.get(0), pairs

    def minimalize(self, G):
        from deepgroebner_amd import PolyLists
        L = PolyLists([list(G)]); L.minimalize()
        return L.get(0)

    def interreduce(self, G):
        from deepgroebner_amd import PolyLists
        L = PolyLists([list(G)]); L.interreduce()
        return L.get(0)

    def buchberger(self, F, elimination="gebauermoeller"):
        from deepgroebner_amd import buchberger
        G, stats = buchberger(F, elimination=elimination)
        return [[(c, tuple(list(e) + [0] * (8 - len(e)))) for c, e in f] for f in G], stats

    def cyclic(self, n):
        from deepgroebner_amd.ideals import cyclic
        return cyclic(n)


@pytest.fixture
def lib():
    return HipLib()


def test_free_functions_in_the_shape_of_the_python_reference():
    """spoly / reduce / update / minimalize / interreduce / buchberger of deepgroebner_amd (buchberger.py:11-240): term lists in,
    term lists of the ring's width out; reduce returns (remainder, {'steps': n}); update modifies G and P."""
    from deepgroebner_amd import buchberger, interreduce, minimalize, reduce, reduce_many, spoly, update
    g = [(1, (5, 10, 4)), (22982, (3, 1, 2))]
    F = [[(1, (5, 12, 0)), (25797, (1, 5, 2))], [(1, (1, 3, 1)), (27630, (2, 1, 0))], [(1, (1, 9, 1)), (8749, (2, 0, 0))]]
    r, stats = reduce(g, F)
    assert r == [(2065, (9, 2, 0)), (22982, (3, 1, 2))] and stats == {"steps": 4}
    assert reduce_many([(g, F), (F[0], F)]) == [(r, {"steps": 4}), ([], {"steps": 1})]
    assert spoly([(1, (2, 0)), (1, (1, 1))], [(1, (0, 2)), (1, (1, 1))]) == []
    G = [[(1, (1, 2, 0)), (2, (0, 0, 1))], [(1, (1, 0, 2)), (32002, (0, 2, 0)), (32002, (0, 0, 1))], [(1, (1, 0, 0)), (3, (0, 0, 0))]]
    Pl = [(0, 2)]
    f = [(1, (0, 2, 3)), (32002, (0, 2, 0)), (4, (0, 0, 4)), (1, (0, 0, 2))]
    G2, P2 = update(G, Pl, f)
    assert G2 is G and P2 is Pl and len(G) == 4 and Pl == [(0, 2)]            # the docstring example of buchberger.py:97-108
    with pytest.raises(ValueError):
        update(G, Pl, f, strategy="bogus")
    basis, stats = buchberger([[(1, (0, 1, 0)), (32002, (2, 0, 0))], [(1, (0, 0, 1)), (32002, (3, 0, 0))]])
    assert basis == [[(1, (0, 2, 0)), (32002, (1, 0, 1))], [(1, (1, 1, 0)), (32002, (0, 0, 1))], [(1, (2, 0, 0)), (32002, (0, 1, 0))]]
    assert set(stats) == {"zero_reductions", "nonzero_reductions", "polynomial_additions", "total_reward", "discounted_return"}
    assert interreduce(minimalize(basis)) == basis


def test_batched_algebra_against_the_oracle_on_random_polynomials():
    """Many independent problems per launch, polynomials long enough for the merge-path tiles and records that have to grow:
    +, -, *, spoly, reduce, minimalize and interreduce of random dense polynomials in 3, 5 and 8 variables, every result
    against the oracle's."""
    from deepgroebner_amd import PolyLists
    from oracle import ffi
    bo = ffi.load("bo")
    rng = np.random.default_rng(4)
    for nv, deg, nterms, nlists in ((3, 9, 40, 24), (5, 5, 90, 12), (8, 3, 60, 8)):
        def rand_poly(nt):
            seen, out = set(), []
            while len(out) < nt:
                e = tuple(int(x) for x in rng.multinomial(int(rng.integers(0, deg + 1)), np.ones(nv) / nv))
                if e not in seen:
                    seen.add(e); out.append((int(rng.integers(1, 32003)), e))
            return bo.polylist([out]).get(0)                    # (sorted like the Polynomial constructor does)
        lists = [[rand_poly(int(rng.integers(2, nterms))) for _ in range(4)] for _ in range(nlists)]
        L = PolyLists(lists)
        for op, name in (("add", "poly_add"), ("sub", "poly_sub"), ("mul", "poly_mul"), ("spoly", "spoly")):
            L.binop(op, (0, 1))
            for k in range(nlists):
                assert L.get(k)[-1] == bo.binop(name, lists[k][0], lists[k][1]), (nv, op, k)
        L = PolyLists(lists)
        steps = L.reduce((3, 3))                               # element 3 by elements 0..2
        for k in range(nlists):
            r, s = bo.reduce(lists[k][3], lists[k][:3])
            assert L.get(k)[-1] == r and steps[k] == s, (nv, "reduce", k)
        L = PolyLists(lists); L.minimalize()
        mins = [bo.minimalize(lists[k]) for k in range(nlists)]
        for k in range(nlists):
            assert L.get(k) == mins[k], (nv, "minimalize", k)
        L.interreduce()
        for k in range(nlists):
            assert L.get(k) == bo.interreduce(mins[k]), (nv, "interreduce", k)


def test_reduced_bases_of_a_batch_device_to_device():
    """bbx_alg_from_envs + minimalize + interreduce: the reduced Groebner bases of a batch of finished runs (binomial and
    general layouts) equal interreduce(minimalize(G)) of the oracle on the same final bases."""
    from deepgroebner_amd import PolyLists, VecLeadMonomialsEnv
    from oracle import ffi
    bo = ffi.load("bo")
    for dist, B in (("3-20-10-weighted", 16), ("3-5-4-0.5-uniform", 8), ("5-10-5-uniform", 4)):
        env = VecLeadMonomialsEnv(dist, batch=B, k=1)
        env.seed(np.arange(B) + 300); env.reset()
        env.rollout("degree", 1 << 30, auto_reset=False)
        L = PolyLists.from_envs(env)
        L.minimalize(); L.interreduce()
        for e in range(B):
            basis, _, _ = env.state(e)
            G = [[(int(c), tuple(int(x) for x in ex)) for c, ex in zip(cs, es)] for cs, es in basis]
            assert L.get(e) == bo.interreduce(bo.minimalize(G)), (dist, e)
            assert env.reduced_basis(e) == L.get(e)
