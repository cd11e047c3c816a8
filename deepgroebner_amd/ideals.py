"""Ideal generators with the surface of the reference's deepgroebner/ideals.py, producing the
seeded streams of the reference's C++ generators (deepgroebner/ideals.cpp) through libbbx.

A polynomial is a list of (coefficient, exponent-tuple) terms in descending grevlex order,
coefficients in GF(32003); an ideal is a list of polynomials.
"""
import ctypes as C

import numpy as np

from . import _ffi

NV = 8


def _terms(nterms, coefs, exps, n):
    out, at = [], 0
    for m in nterms:
        out.append([(int(coefs[at + t]), tuple(int(x) for x in exps[at + t, :n])) for t in range(m)])
        at += m
    return out


class IdealGenerator:
    """Iterator over ideals (reference ideals.py:142-170 / ideals.h:86-104)."""

    def __init__(self, dist):
        self.dist = dist
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().bbx_gen_create(dist.encode(), C.byref(self._h)))
        self.nvars = _ffi.lib().bbx_gen_nvars(self._h)

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().bbx_gen_destroy(self._h)
        except Exception:
            pass

    def __iter__(self):
        return self

    def seed(self, seed=None):
        if seed is not None:
            _ffi.check(_ffi.lib().bbx_gen_seed(self._h, int(seed)))

    def __next__(self):
        np_, nt = C.c_int32(), C.c_int32()
        _ffi.check(_ffi.lib().bbx_gen_next(self._h, C.byref(np_), C.byref(nt)))
        nterms = np.zeros(max(np_.value, 1), dtype=np.int32)
        coefs = np.zeros(max(nt.value, 1), dtype=np.int32)
        exps = np.zeros((max(nt.value, 1), NV), dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_gen_get(self._h, _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), None))
        return _terms(nterms[:np_.value], coefs, exps, NV)


class FixedIdealGenerator:
    """Repeats one ideal (reference ideals.py:173-189).  F: list of term lists."""

    def __init__(self, F):
        self.F = [[(int(c), tuple(int(x) for x in e)) for c, e in f] for f in F]
        self.nvars = max(len(e) for f in self.F for _, e in f)

    def __iter__(self):
        return self

    def __next__(self):
        return self.F

    def seed(self, seed=None):
        pass


def _dist_name(n, d, s, dist, constants, homogeneous, pure=False, lam=None):
    parts = [str(n), str(d), str(s)] + ([repr(float(lam))] if lam is not None else []) + [dist]
    if constants:
        parts.append("consts")
    if homogeneous:
        parts.append("homog")
    if pure:
        parts.append("pure")
    return "-".join(parts)


P = 32003


def _grevlex_key(e):
    """Sort key of an exponent tuple: larger key = larger monomial in grevlex (what the reference's Python path compares,
    ideals.py:236-238)."""
    return (sum(e), tuple(-x for x in reversed(e)))


class _NumpyStream:
    """The seeded stream of the reference's PYTHON generators (ideals.py:214, 250, 302: numpy's default_rng, PCG64) instead of
    the C++ generators' minstd_rand0: the same draws in the same order — integers(1, P) for a coefficient, choice(len, p=...)
    for a degree, choice(len) for a monomial of that degree in the enumeration order of ideals.py:18-43, poisson(lam) for a
    length — so that generator.seed(123) yields the ideals the reference's tests/test_ideals.py:48-69 expect.  Host-side only
    (the batch environments draw on the device, from the C++ streams)."""

    def _setup(self, n, d, degrees, constants):
        import itertools as it
        from math import comb
        self.nvars = n
        self.bases = [[tuple([0] * n)]]
        for deg in range(1, d + 1):
            row = []
            for combo in it.combinations_with_replacement(range(n), deg):
                e = [0] * n
                for v in combo:
                    e[v] += 1
                row.append(tuple(e))
            self.bases.append(row)
        head = [1 if constants else 0]
        if degrees == "uniform":
            tail = [comb(n + i - 1, n - 1) for i in range(1, d + 1)]
        elif degrees == "weighted":
            tail = d * [1]
        elif degrees == "maximum":
            tail = (d - 1) * [0] + [1]
        else:
            raise ValueError("unrecognized dist option")
        count = np.array(head + tail)
        self.degree_dist = count / np.sum(count)
        self.rng = np.random.default_rng()

    def __iter__(self):
        return self

    def seed(self, seed=None):
        self.rng = np.random.default_rng(seed)

    def _monomial(self, deg):
        row = self.bases[deg]
        return row[self.rng.choice(len(row))]


class _NumpyBinomials(_NumpyStream):
    def __init__(self, n, d, s, degrees, constants, homogeneous, pure):
        self._setup(n, d, degrees, constants)
        self.s, self.homogeneous, self.pure = s, homogeneous, pure

    def __next__(self):
        F = []
        for _ in range(self.s):
            c = P - 1 if self.pure else int(self.rng.integers(1, P))
            if self.homogeneous:
                d1 = d2 = self.rng.choice(len(self.degree_dist), p=self.degree_dist)
            else:
                d1, d2 = self.rng.choice(len(self.degree_dist), size=2, p=self.degree_dist)
            for _ in range(1000):
                m1, m2 = self._monomial(d1), self._monomial(d2)
                k1, k2 = _grevlex_key(m1), _grevlex_key(m2)
                if k1 != k2:
                    lead, tail = (m1, m2) if k1 > k2 else (m2, m1)
                    F.append([(1, lead), (c, tail)])
                    break
            else:
                raise RuntimeError("failed to generate two distinct random monomials after 1000 trials")
        return F


class _NumpyPolynomials(_NumpyStream):
    def __init__(self, n, d, s, lam, degrees, constants, homogeneous):
        self._setup(n, d, degrees, constants)
        self.s, self.lam, self.homogeneous = s, lam, homogeneous

    def __next__(self):
        F = []
        for _ in range(self.s):
            f = {}
            terms = 2 + self.rng.poisson(self.lam)
            deg = self.rng.choice(len(self.degree_dist), p=self.degree_dist)
            for _ in range(terms):
                c = int(self.rng.integers(1, P))
                m = self._monomial(deg)
                f[m] = (f.get(m, 0) + c) % P                     # (terms are not checked to be distinct: they add up or cancel)
                if not self.homogeneous:
                    deg = self.rng.choice(len(self.degree_dist), p=self.degree_dist)
            f = {m: c for m, c in f.items() if c}
            if not f:                                            # (everything cancelled: the zero polynomial, as the reference's f.monic())
                F.append([])
                continue
            inv = pow(f[max(f, key=_grevlex_key)], P - 2, P)
            F.append(sorted(((c * inv % P, m) for m, c in f.items()), key=lambda t: _grevlex_key(t[1]), reverse=True))
        return F


class RandomBinomialIdealGenerator(IdealGenerator):
    """Reference ideals.py:192-250 / ideals.cpp:156-201.  stream="cpp" (default): the C++ generator's seeded stream — what the
    environments draw; stream="numpy": the Python generator's (see _NumpyStream), exponent tuples of n entries."""

    def __new__(cls, n=3, d=20, s=10, degrees="uniform", constants=False, homogeneous=False, pure=False, stream="cpp"):
        if stream == "numpy":
            return _NumpyBinomials(n, d, s, degrees, constants, homogeneous, pure)
        return super().__new__(cls)

    def __init__(self, n=3, d=20, s=10, degrees="uniform", constants=False, homogeneous=False, pure=False, stream="cpp"):
        if stream != "cpp":
            raise ValueError("stream must be 'cpp' or 'numpy'")
        super().__init__(_dist_name(n, d, s, degrees, constants, homogeneous, pure))


class RandomIdealGenerator(IdealGenerator):
    """Reference ideals.py:253-323 / ideals.cpp:203-231.  stream: as for RandomBinomialIdealGenerator."""

    def __new__(cls, n=3, d=20, s=10, lam=0.5, degrees="uniform", constants=False, homogeneous=False, stream="cpp"):
        if stream == "numpy":
            return _NumpyPolynomials(n, d, s, lam, degrees, constants, homogeneous)
        return super().__new__(cls)

    def __init__(self, n=3, d=20, s=10, lam=0.5, degrees="uniform", constants=False, homogeneous=False, stream="cpp"):
        if stream != "cpp":
            raise ValueError("stream must be 'cpp' or 'numpy'")
        super().__init__(_dist_name(n, d, s, degrees, constants, homogeneous, lam=lam))


def parse_ideal_dist(ideal_dist):
    """String -> generator, same grammar as the reference (ideals.py:112-139, ideals.cpp:103-143)."""
    return IdealGenerator(ideal_dist)


def cyclic(n):
    g = IdealGenerator("cyclic-%d" % n)
    return [[(c, e[:n]) for c, e in f] for f in next(g)]


def basis(n, d):
    """Monomials of degree d in n variables in the reference's enumeration order (ideals.cpp:39-64)."""
    out = []

    def rec(prefix, left, slots):
        if slots == 1:
            out.append(tuple(prefix + [left]))
            return
        for first in range(left, -1, -1):
            rec(prefix + [first], left - first, slots - 1)
    rec([], d, n)
    return out


def degree_distribution(n, d, dist="uniform", constants=False):
    """Probabilities over degrees 0..d (ideals.cpp:75-100)."""
    from math import comb
    w = [1 if constants else 0]
    if dist == "uniform":
        w += [comb(n + i - 1, n - 1) for i in range(1, d + 1)]
    elif dist == "weighted":
        w += [1] * d
    elif dist == "maximum":
        w += [0] * (d - 1) + [1]
    else:
        raise ValueError("unrecognized distribution type")
    tot = float(sum(w))
    return [x / tot for x in w]


def parse_ideal_string(text):
    """One line of data/stats/<dist>/<dist>.csv -> ideal: polynomials like "413*a^2*b^5*c+32*d^2-5" joined by "|"
    (reference parse_polynomial, polynomials.cpp:226-300; parse_ideal_string, scripts/make_strat.cpp:12-19).
    Exponent tuples keep all 8 slots, as the reference's Monomial does."""
    raw = text.strip().encode()
    cap_p, cap_t = raw.count(b"|") + 1, max(len(raw), 1)
    np_, nt = C.c_int32(), C.c_int32()
    nterms = np.zeros(cap_p, dtype=np.int32)
    coefs = np.zeros(cap_t, dtype=np.int32)
    exps = np.zeros((cap_t, NV), dtype=np.int32)
    _ffi.check(_ffi.lib().bbx_parse_ideal(raw, cap_p, cap_t, C.byref(np_), C.byref(nt), _ffi.ptr(nterms), _ffi.ptr(coefs),
                                          _ffi.ptr(exps)))
    return _terms(nterms[:np_.value], coefs, exps, NV)


def parse_polynomial(text):
    """reference parse_polynomial (polynomials.cpp:297-300) for one non-zero polynomial."""
    if "|" in text:
        raise ValueError("one polynomial expected")
    return parse_ideal_string(text)[0]


def format_ideal(F):
    """The inverse of parse_ideal_string: the line scripts/make_dist.m2:69-77 of the reference writes for an ideal."""
    nterms = np.array([len(f) for f in F], dtype=np.int32)
    coefs = np.array([c for f in F for c, _ in f], dtype=np.int32)
    exps = np.zeros((max(len(coefs), 1), NV), dtype=np.int32)
    r = 0
    for f in F:
        for _, e in f:
            exps[r, :len(e)] = e
            r += 1
    L = _ffi.lib()
    need = L.bbx_format_ideal(len(F), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), None, 0)
    if need < 0:
        _ffi.check(need)
    buf = C.create_string_buffer(need + 1)
    _ffi.check(min(0, L.bbx_format_ideal(len(F), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), buf, need + 1)))
    return buf.value.decode()
