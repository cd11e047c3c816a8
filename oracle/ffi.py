"""TEST INFRASTRUCTURE — ctypes front-end shared by the two CPU checkers.

`load("bo")`  -> oracle/_build/liboracle.so  (our C restatement, oracle/bbx_oracle.c)
`load("ref")` -> oracle/_ref/libref.so       (the reference C++ itself + oracle/ref_driver.cpp)

Both export the same function set with a different prefix, so every helper
below works on either.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NV = 8  # exponent slots per monomial (polynomials.h:29)

_PATHS = {"bo": os.path.join(HERE, "_build", "liboracle.so"),
          "ref": os.path.join(HERE, "_ref", "libref.so")}

ELIM = {"gebauermoeller": 0, "lcm": 1, "none": 2}
REWARDS = {"additions": 0, "reductions": 1}
SELECTION = {"first": 0, "degree": 1, "normal": 2, "sugar": 3, "random": 4,
             "last": 5, "codegree": 6, "strange": 7, "spice": 8}
DIST = {"uniform": 0, "weighted": 1, "maximum": 2}


def build():
    """(Re)build both checkers; the reference one only where /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"], stdout=sys.stderr)


def available(kind):
    return os.path.exists(_PATHS[kind])


_ip = C.POINTER(C.c_int)
_vp = C.c_void_p


def _sig(lib, pre):
    def f(name, res, *args):
        fn = getattr(lib, pre + "_" + name)
        fn.restype = res
        fn.argtypes = list(args)
    f("pl_new", _vp); f("pl_free", None, _vp); f("pl_clear", None, _vp); f("pl_len", C.c_int, _vp)
    f("pl_add", None, _vp, C.c_int, _ip, _ip); f("pl_nterms", C.c_int, _vp, C.c_int)
    f("pl_sugar", C.c_int, _vp, C.c_int); f("pl_get", None, _vp, C.c_int, _ip, _ip)
    for n in ("coef_add", "coef_sub", "coef_mul", "coef_div"):
        f(n, C.c_int, C.c_int, C.c_int)
    f("coef_norm", C.c_int, C.c_int); f("mono_gt", C.c_int, _ip, _ip)
    for n in ("poly_add", "poly_sub", "poly_mul", "spoly"):
        f(n, None, _vp, C.c_int, C.c_int, _vp)
    f("parse_polynomial", None, C.c_char_p, _vp)
    f("reduce", C.c_int, _vp, C.c_int, _vp, _vp)
    f("update", C.c_int, _vp, _ip, C.c_int, _vp, C.c_int, C.c_int)
    f("minimalize", None, _vp, _vp); f("interreduce", None, _vp, _vp)
    f("buchberger", None, _vp, _ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
      C.c_int, C.c_int, _vp, C.POINTER(C.c_double))
    f("cyclic", None, C.c_int, _vp); f("basis", C.c_int, C.c_int, C.c_int, _ip, C.c_int)
    f("degree_distribution", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double))
    f("gen_new", _vp, C.c_char_p); f("gen_free", None, _vp); f("gen_seed", None, _vp, C.c_int)
    f("gen_nvars", C.c_int, _vp); f("gen_next", None, _vp, _vp); f("gen_copy", _vp, _vp)
    f("env_new", _vp, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int)
    f("env_new_fixed", _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int)
    f("env_free", None, _vp); f("env_copy", _vp, _vp); f("env_seed", None, _vp, C.c_int)
    f("env_nvars", C.c_int, _vp); f("env_reset", None, _vp)
    f("env_step_pair", C.c_double, _vp, C.c_int, C.c_int); f("env_step", C.c_double, _vp, C.c_int)
    f("env_value", C.c_double, _vp, C.c_char_p, C.c_double)
    f("env_nG", C.c_int, _vp); f("env_nP", C.c_int, _vp); f("env_pairs", None, _vp, _ip)
    f("env_poly_nterms", C.c_int, _vp, C.c_int); f("env_poly_sugar", C.c_int, _vp, C.c_int)
    f("env_poly_get", None, _vp, C.c_int, _ip, _ip); f("env_reducer_order", None, _vp, _ip)
    f("env_obs", None, _vp, C.c_int, C.c_int, _ip)
    f("bench_random", C.c_double, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
      C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_ulonglong))
    if pre == "bo":
        f("env_last_step_bytes", C.c_longlong, _vp)
        f("stat_max_terms", C.c_int, C.c_int)
        f("run_random", C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong))
    else:
        f("lme_new", _vp, C.c_char_p, C.c_int, C.c_int, C.c_int); f("lme_free", None, _vp)
        f("lme_copy", _vp, _vp); f("lme_seed", None, _vp, C.c_int); f("lme_reset", None, _vp)
        f("lme_step", C.c_double, _vp, C.c_int); f("lme_value", C.c_double, _vp, C.c_char_p, C.c_double)
        f("lme_cols", C.c_int, _vp); f("lme_state_size", C.c_int, _vp); f("lme_state", None, _vp, _ip)


def _ia(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def pad_exps(exps):
    """[[e0,e1,..],..] with <=8 entries per monomial -> int32 [n, 8]."""
    out = np.zeros((len(exps), NV), dtype=np.int32)
    for r, e in enumerate(exps):
        out[r, :len(e)] = e
    return out


class PolyList:
    """A std::vector<Polynomial>.  Polynomials go in/out as [(coef, (e0,e1,...)), ...]."""

    def __init__(self, lib, polys=()):
        self.lib = lib
        self.h = lib.fn("pl_new")()
        for p in polys:
            self.add(p)

    def __del__(self):
        try:
            self.lib.fn("pl_free")(self.h)
        except Exception:
            pass

    def add(self, terms):
        coef, cp = _ia([t[0] for t in terms])
        exps, ep = _ia(pad_exps([t[1] for t in terms]) if terms else np.zeros((0, NV)))
        self.lib.fn("pl_add")(self.h, len(terms), cp, ep)

    def __len__(self):
        return self.lib.fn("pl_len")(self.h)

    def sugar(self, i):
        return self.lib.fn("pl_sugar")(self.h, i)

    def get(self, i):
        n = self.lib.fn("pl_nterms")(self.h, i)
        coef = np.zeros(max(n, 1), dtype=np.int32)
        exps = np.zeros((max(n, 1), NV), dtype=np.int32)
        self.lib.fn("pl_get")(self.h, i, coef.ctypes.data_as(_ip), exps.ctypes.data_as(_ip))
        return [(int(coef[k]), tuple(int(x) for x in exps[k])) for k in range(n)]

    def all(self):
        return [self.get(i) for i in range(len(self))]


class Env:
    """BuchbergerEnv (buchberger.h:161-208) on either checker."""

    def __init__(self, lib, dist=None, fixed=None, elimination="gebauermoeller", rewards="additions",
                 sort_input=False, sort_reducers=True, handle=None):
        self.lib = lib
        if handle is not None:
            self.h = handle
        elif fixed is not None:
            pl = fixed if isinstance(fixed, PolyList) else PolyList(lib, fixed)
            self.h = lib.fn("env_new_fixed")(pl.h, ELIM[elimination], REWARDS[rewards], int(sort_input), int(sort_reducers))
        else:
            self.h = lib.fn("env_new")(dist.encode(), ELIM[elimination], REWARDS[rewards], int(sort_input), int(sort_reducers))
        if not self.h:
            raise ValueError("could not build env for %r" % (dist,))

    def __del__(self):
        try:
            self.lib.fn("env_free")(self.h)
        except Exception:
            pass

    def copy(self):
        return Env(self.lib, handle=self.lib.fn("env_copy")(self.h))

    def seed(self, s):
        self.lib.fn("env_seed")(self.h, int(s))

    def nvars(self):
        return self.lib.fn("env_nvars")(self.h)

    def reset(self):
        self.lib.fn("env_reset")(self.h)

    def step(self, action):
        return self.lib.fn("env_step")(self.h, int(action))

    def step_pair(self, i, j):
        return self.lib.fn("env_step_pair")(self.h, int(i), int(j))

    def value(self, strategy="degree", gamma=0.99):
        return self.lib.fn("env_value")(self.h, strategy.encode(), float(gamma))

    @property
    def nG(self):
        return self.lib.fn("env_nG")(self.h)

    @property
    def nP(self):
        return self.lib.fn("env_nP")(self.h)

    def pairs(self):
        out = np.zeros((max(self.nP, 1), 2), dtype=np.int32)
        self.lib.fn("env_pairs")(self.h, out.ctypes.data_as(_ip))
        return out[:self.nP].copy()

    def poly(self, i):
        n = self.lib.fn("env_poly_nterms")(self.h, i)
        coef = np.zeros(max(n, 1), dtype=np.int32)
        exps = np.zeros((max(n, 1), NV), dtype=np.int32)
        self.lib.fn("env_poly_get")(self.h, i, coef.ctypes.data_as(_ip), exps.ctypes.data_as(_ip))
        return coef[:n].copy(), exps[:n].copy()

    def poly_sugar(self, i):
        return self.lib.fn("env_poly_sugar")(self.h, i)

    def basis(self):
        return [self.poly(i) for i in range(self.nG)]

    def reducer_order(self):
        out = np.zeros(max(self.nG, 1), dtype=np.int32)
        self.lib.fn("env_reducer_order")(self.h, out.ctypes.data_as(_ip))
        return out[:self.nG].copy()

    def obs(self, k, n=None):
        n = self.nvars() if n is None else n
        out = np.zeros((max(self.nP, 1), 2 * n * k), dtype=np.int32)
        self.lib.fn("env_obs")(self.h, k, n, out.ctypes.data_as(_ip))
        return out[:self.nP].copy()

    def last_step_bytes(self):
        return self.lib.fn("env_last_step_bytes")(self.h)


class Lib:
    def __init__(self, kind):
        self.kind = kind
        self.pre = kind
        self.dll = C.CDLL(_PATHS[kind])
        _sig(self.dll, kind)

    def fn(self, name):
        return getattr(self.dll, self.pre + "_" + name)

    # ---- scalar helpers
    def coef(self, op, a, b=None):
        return self.fn("coef_" + op)(a) if b is None else self.fn("coef_" + op)(a, b)

    def mono_gt(self, a, b):
        a_, ap = _ia(pad_exps([a])[0]); b_, bp = _ia(pad_exps([b])[0])
        return bool(self.fn("mono_gt")(ap, bp))

    # ---- function-level wrappers on python term lists
    def polylist(self, polys=()):
        return PolyList(self, polys)

    def binop(self, name, f, g):
        pl, out = PolyList(self, [f, g]), PolyList(self)
        self.fn(name)(pl.h, 0, 1, out.h)
        return out.get(0)

    def spoly(self, f, g):
        return self.binop("spoly", f, g)

    def parse_polynomial(self, s):
        out = PolyList(self)
        self.fn("parse_polynomial")(s.encode(), out.h)
        return out.get(0)

    def reduce(self, g, F):
        plg, plF, out = PolyList(self, [g]), PolyList(self, F), PolyList(self)
        steps = self.fn("reduce")(plg.h, 0, plF.h, out.h)
        return out.get(0), steps

    def update(self, G, P, f, elimination="gebauermoeller"):
        plG, plf = PolyList(self, G), PolyList(self, [f])
        buf = np.zeros((len(P) + len(G) + 1, 2), dtype=np.int32)
        if len(P):
            buf[:len(P)] = np.asarray(P, dtype=np.int32)
        n = self.fn("update")(plG.h, buf.ctypes.data_as(_ip), len(P), plf.h, 0, ELIM[elimination])
        return plG.all(), [tuple(int(x) for x in r) for r in buf[:n]]

    def minimalize(self, G):
        pl, out = PolyList(self, G), PolyList(self)
        self.fn("minimalize")(pl.h, out.h)
        return out.all()

    def interreduce(self, G):
        pl, out = PolyList(self, G), PolyList(self)
        self.fn("interreduce")(pl.h, out.h)
        return out.all()

    def buchberger(self, F, S=None, selection="degree", elimination="gebauermoeller", rewards="additions",
                   sort_input=False, sort_reducers=True, gamma=0.99, seed=None, want_basis=True):
        pl = F if isinstance(F, PolyList) else PolyList(self, F)
        out = PolyList(self)
        stats = (C.c_double * 5)()
        if S is None:
            pp, n = None, -1
        else:
            arr, pp = _ia(np.asarray(S, dtype=np.int32).reshape(-1, 2)); n = len(S)
        self.fn("buchberger")(pl.h, pp, n, SELECTION[selection], ELIM[elimination], REWARDS[rewards], int(sort_input),
                              int(sort_reducers), float(gamma), int(seed is not None), int(seed or 0),
                              out.h if want_basis else None, stats)
        keys = ("zero_reductions", "nonzero_reductions", "polynomial_additions", "total_reward", "discounted_return")
        return (out.all() if want_basis else None), dict(zip(keys, list(stats)))

    def cyclic(self, n):
        out = PolyList(self)
        self.fn("cyclic")(n, out.h)
        return out.all()

    def basis(self, n, d):
        cap = 1 << 16
        buf = np.zeros((cap, NV), dtype=np.int32)
        m = self.fn("basis")(n, d, buf.ctypes.data_as(_ip), cap)
        return buf[:m].copy()

    def degree_distribution(self, n, d, dist="uniform", constants=False):
        buf = (C.c_double * (d + 2))()
        m = self.fn("degree_distribution")(n, d, DIST[dist], int(constants), buf)
        return list(buf)[:m]

    def generator(self, dist):
        return Generator(self, dist)

    def env(self, *a, **kw):
        return Env(self, *a, **kw)

    def run_random(self, dist, k, seed, agent_seed, nsteps, auto_reset=True, nobs=0):
        """One environment under the counter-hash agent (C restatement only): dict of counters + final-state hash.  The call
        releases the GIL, so a thread pool runs many environments on all host cores."""
        out = (C.c_longlong * 8)()
        if self.fn("run_random")(dist.encode(), k, int(seed), int(agent_seed), int(nsteps), int(auto_reset), int(nobs), out) != 0:
            raise ValueError("bad distribution %r" % (dist,))
        keys = ("steps", "additions", "bytes", "episodes", "nG", "nP", "state_hash", "zero_reductions")
        d = dict(zip(keys, [int(v) for v in out]))
        d["state_hash"] &= (1 << 64) - 1
        return d

    def run_random_many(self, dist, k, seeds, agent_seeds, nsteps, auto_reset=True, nobs=0, threads=None):
        from concurrent.futures import ThreadPoolExecutor
        threads = threads or min(16, os.cpu_count() or 1)
        with ThreadPoolExecutor(threads) as ex:
            return list(ex.map(lambda sa: self.run_random(dist, k, sa[0], sa[1], nsteps, auto_reset, nobs), zip(seeds, agent_seeds)))

    def bench_random(self, dist, k, nenvs, nsteps, seed0, agent_seed0):
        ts, ta, cs = C.c_longlong(), C.c_longlong(), C.c_ulonglong()
        sec = self.fn("bench_random")(dist.encode(), k, nenvs, nsteps, seed0, agent_seed0,
                                      C.byref(ts), C.byref(ta), C.byref(cs))
        return {"seconds": sec, "steps": ts.value, "additions": ta.value, "checksum": cs.value}


class Generator:
    def __init__(self, lib, dist, handle=None):
        self.lib = lib
        self.h = handle if handle is not None else lib.fn("gen_new")(dist.encode())
        if not self.h:
            raise ValueError("bad distribution string %r" % (dist,))

    def __del__(self):
        try:
            self.lib.fn("gen_free")(self.h)
        except Exception:
            pass

    def seed(self, s):
        self.lib.fn("gen_seed")(self.h, int(s))

    def nvars(self):
        return self.lib.fn("gen_nvars")(self.h)

    def next(self):
        out = PolyList(self.lib)
        self.lib.fn("gen_next")(self.h, out.h)
        return out.all()

    def copy(self):
        return Generator(self.lib, None, handle=self.lib.fn("gen_copy")(self.h))


_cache = {}


def load(kind):
    if kind not in _cache:
        if not available(kind):
            if kind == "bo" or os.path.isdir("/root/reference"):
                build()
        _cache[kind] = Lib(kind)
    return _cache[kind]


def agent_action(seed, t, rows):
    """Row chosen by the built-in random agent at its step t: multiply-shift range reduction of the counter hash
    (bbx_agent_action in include/bbx.h; same in oracle/bbx_oracle.c and oracle/ref_driver.cpp)."""
    return (agent_hash(seed, t) * int(rows)) >> 32


def agent_hash(seed, t):
    """The counter-based action hash shared by device, oracle and reference driver."""
    m = (1 << 64) - 1
    z = ((((seed & 0xFFFFFFFF) << 32) | (t & 0xFFFFFFFF)) + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z = z ^ (z >> 31)
    return z >> 32
