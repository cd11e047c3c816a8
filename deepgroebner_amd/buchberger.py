"""Gym-style Buchberger environments with the surface of the reference's
deepgroebner.wrapped.CLeadMonomialsEnv (wrapped.pyx:11-38) and
deepgroebner.buchberger.LeadMonomialsEnv (buchberger.py:448-542), running on libbbx (HIP, gfx950).

reset() -> int32 [rows, 2*n*k]; step(a) -> (state, reward, done, {}); seed(); value(); copy().
VecLeadMonomialsEnv steps a whole batch of independent environments per call.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .ideals import FixedIdealGenerator

NV = 8


def _caps(caps):
    if caps is None:
        return None
    c = _ffi.Caps()
    for key, val in caps.items():
        setattr(c, key, int(val))
    return C.byref(c)


class VecLeadMonomialsEnv:
    """`batch` independent LeadMonomialsEnv instances stepped together on one GPU.

    ideal_dist : distribution string (reference grammar) or FixedIdealGenerator
    mimic : 'cpp'  -> observation width follows the C++ class (FixedIdealGenerator's
                      largest-variable-index quirk, ideals.cpp:146-154)
            'python' -> width follows ring.ngens like buchberger.py:537
    """

    def __init__(self, ideal_dist="3-20-10-weighted", batch=1, elimination="gebauermoeller", rewards="additions",
                 sort_input=False, sort_reducers=True, k=1, device=0, caps=None, mimic="cpp"):
        L = _ffi.lib()
        self._h = C.c_void_p()
        self.batch, self.k = int(batch), int(k)
        el, rw = _ffi.ELIMINATION[elimination], _ffi.REWARDS[rewards]
        if isinstance(ideal_dist, (list, tuple)):             # a list of ideals: environment e gets ideals e, e+batch, ...
            ideals = [[[(int(c), tuple(int(x) for x in e)) for c, e in f] for f in F] for F in ideal_dist]
            npolys = np.array([len(F) for F in ideals], dtype=np.int32)
            nterms = np.array([len(f) for F in ideals for f in F], dtype=np.int32)
            coefs = np.array([c for F in ideals for f in F for c, _ in f], dtype=np.int32)
            exps = np.zeros((len(coefs), NV), dtype=np.int32)
            r = 0
            for F in ideals:
                for f in F:
                    for _, e in f:
                        exps[r, :len(e)] = e
                        r += 1
            used = np.flatnonzero(exps.any(axis=0))                # trailing all-zero slots are not variables
            nv = int(used[-1]) + 1 if len(used) else 1
            _ffi.check(L.bbx_create_ideals(len(ideals), _ffi.ptr(npolys), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), nv, el, rw,
                                           int(sort_input), int(sort_reducers), self.k, self.batch, int(device),
                                           _caps(caps), C.byref(self._h)))
        elif isinstance(ideal_dist, FixedIdealGenerator):
            F = ideal_dist.F
            nterms = np.array([len(f) for f in F], dtype=np.int32)
            coefs = np.array([c for f in F for c, _ in f], dtype=np.int32)
            exps = np.zeros((len(coefs), NV), dtype=np.int32)
            r = 0
            for f in F:
                for _, e in f:
                    exps[r, :len(e)] = e
                    r += 1
            nv = ideal_dist.nvars if mimic == "python" else 0
            _ffi.check(L.bbx_create_fixed(len(F), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), nv, el, rw,
                                          int(sort_input), int(sort_reducers), self.k, self.batch, int(device),
                                          _caps(caps), C.byref(self._h)))
        else:
            _ffi.check(L.bbx_create(str(ideal_dist).encode(), el, rw, int(sort_input), int(sort_reducers), self.k,
                                    self.batch, int(device), _caps(caps), C.byref(self._h)))
        self.cols = L.bbx_cols(self._h)
        self.nvars = L.bbx_nvars(self._h)
        self.rows = np.zeros(self.batch, dtype=np.int32)
        self._rewards = np.zeros(self.batch, dtype=np.float64)
        self._dones = np.zeros(self.batch, dtype=np.uint8)

    @classmethod
    def _from_handle(cls, h, src):
        self = cls.__new__(cls)
        self._h = h
        self.batch, self.k, self.cols, self.nvars = src.batch, src.k, src.cols, src.nvars
        self.rows = src.rows.copy()
        self._rewards = np.zeros(self.batch, dtype=np.float64)
        self._dones = np.zeros(self.batch, dtype=np.uint8)
        return self

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().bbx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- reference surface, batched
    def seed(self, seeds=None):
        if seeds is None:
            return
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.int64), (self.batch,)))
        _ffi.check(_ffi.lib().bbx_seed(self._h, _ffi.ptr(s)))

    def seed_agent(self, seeds):
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint32), (self.batch,)))
        _ffi.check(_ffi.lib().bbx_seed_agent(self._h, _ffi.ptr(s)))

    def seed_strategy(self, seeds):
        """Seed the "random_std" agent: the reference's seeded Random selection (buchberger.cpp:200-203, 244)."""
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.int64), (self.batch,)))
        _ffi.check(_ffi.lib().bbx_seed_strategy(self._h, _ffi.ptr(s)))

    def _step_obs(self, actions, auto_reset):
        """One library call: the step (actions None: none) and the ragged observation block -> (flat [sum rows, cols],
        offsets [batch + 1]), fresh copies of the handle's pinned buffers."""
        io = self.__dict__.get("_io")
        if io is None:                                       # ctypes plumbing built once per handle
            io = self._io = {"obs": C.POINTER(C.c_int32)(), "off": C.POINTER(C.c_int32)(), "views": {},
                             "act": np.zeros(self.batch, dtype=np.int32), "fn": _ffi.lib().bbx_step_obs}
            io["args"] = (_ffi.ptr(self._rewards), _ffi.ptr(self._dones), _ffi.ptr(self.rows), C.byref(io["obs"]), C.byref(io["off"]))
            io["pact"] = _ffi.ptr(io["act"])
        if actions is not None:
            io["act"][...] = actions
        _ffi.check(io["fn"](self._h, io["pact"] if actions is not None else None, int(auto_reset), *io["args"]))

        def view(ptr):                                       # numpy view of a pinned buffer, cached by its address
            addr = C.cast(ptr, C.c_void_p).value
            v = io["views"].get(addr)
            if v is None:
                if len(io["views"]) > 8:
                    io["views"].clear()
                v = io["views"][addr] = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32 * (1 << 28))).contents)
            return v
        off = view(io["off"])[:self.batch + 1].copy()
        total = int(off[-1])
        if total == 0:
            return np.zeros((0, self.cols), dtype=np.int32), off
        return view(io["obs"])[:total * self.cols].copy().reshape(total, self.cols), off

    def _step_one(self, action):
        """The single-environment step of CLeadMonomialsEnv.step with nothing the batch interface needs: one library
        call, one copy of the observation out of the pinned buffer -> (state, reward, done)."""
        io = self.__dict__.get("_io1")
        if io is None:
            obs, off = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
            act = np.zeros(1, dtype=np.int32)
            io = self._io1 = {"obs": obs, "off": off, "act": act, "fn": _ffi.lib().bbx_step_obs, "view": None, "addr": None,
                              "args": (self._h, _ffi.ptr(act), 0, _ffi.ptr(self._rewards), _ffi.ptr(self._dones), _ffi.ptr(self.rows),
                                       C.byref(obs), C.byref(off))}
        io["act"][0] = action
        rc = io["fn"](*io["args"])
        if rc:
            _ffi.check(rc)
        total = io["off"][1]
        addr = C.cast(io["obs"], C.c_void_p).value
        if addr != io["addr"]:                               # (the pinned buffer moves only when it has to grow)
            io["addr"] = addr
            io["view"] = np.ctypeslib.as_array(C.cast(io["obs"], C.POINTER(C.c_int32 * (1 << 28))).contents)
        cols = self.cols
        return io["view"][:total * cols].reshape(total, cols).copy(), float(self._rewards[0]), bool(self._dones[0])

    def _as_list(self, flat, off):
        return [flat] if self.batch == 1 else np.split(flat, off[1:-1])

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _ffi.check(_ffi.lib().bbx_reset(self._h, _ffi.ptr(m), _ffi.ptr(self.rows)))
        return self._as_list(*self._step_obs(None, False))

    def step(self, actions, auto_reset=False):
        """auto_reset=True: finished environments start their next episode inside the same call (VecEnv style).
        Returns (list of per-environment int32 [rows_e, cols] matrices, rewards, dones, infos)."""
        flat, off = self._step_obs(actions, auto_reset)
        return self._as_list(flat, off), self._rewards.copy(), self._dones.astype(bool), [{} for _ in range(self.batch)]

    def step_ragged(self, actions, auto_reset=False):
        """step() for vectorised consumers: (flat int32 [sum rows, cols], offsets int32 [batch + 1], rewards, dones);
        environment e owns flat[offsets[e]:offsets[e + 1]]."""
        flat, off = self._step_obs(actions, auto_reset)
        return flat, off, self._rewards.copy(), self._dones.astype(bool)

    def rollout(self, agent="random", nsteps=1, auto_reset=True):
        _ffi.check(_ffi.lib().bbx_rollout(self._h, _ffi.AGENTS[agent], int(nsteps), int(auto_reset),
                                          _ffi.ptr(self._rewards), _ffi.ptr(self._dones), _ffi.ptr(self.rows)))
        return self._rewards.copy(), self._dones.astype(bool), self.rows.copy()

    def observations(self, max_rows=None, fill=True):
        """List of per-environment int32 [rows_e, cols] matrices (max_rows=None) or the padded
        int32 [batch, max_rows, cols] block with -1 fill."""
        mr = int(max(1, self.rows.max())) if max_rows is None else int(max_rows)
        out = np.empty((self.batch, mr, self.cols), dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_obs(self._h, _ffi.ptr(out), mr, int(fill)))
        if max_rows is None:
            return [out[e, :self.rows[e]].copy() for e in range(self.batch)]
        return out

    def value(self, idx=0, strategy="degree", gamma=0.99):
        v = C.c_double()
        _ffi.check(_ffi.lib().bbx_value(self._h, int(idx), strategy.encode(), float(gamma), C.byref(v)))
        return v.value

    def values(self, strategy="degree", gamma=0.99, seeds=None):
        """value() of every environment at once (float64 [batch]).  seeds: the seeds of the Random rollouts behind "random"
        ([batch]) / "sample" ([batch, 100]) — buchberger(..., Random, seed) of the reference; None: the handle draws them."""
        out = np.zeros(self.batch, dtype=np.float64)
        if seeds is None:
            _ffi.check(_ffi.lib().bbx_values(self._h, strategy.encode(), float(gamma), _ffi.ptr(out)))
        else:
            s = np.ascontiguousarray(seeds, dtype=np.int64)
            assert s.size == self.batch * (100 if strategy == "sample" else 1)
            _ffi.check(_ffi.lib().bbx_values_seeded(self._h, strategy.encode(), float(gamma), _ffi.ptr(s), _ffi.ptr(out)))
        return out

    def copy(self):
        h = C.c_void_p()
        _ffi.check(_ffi.lib().bbx_copy(self._h, C.byref(h)))
        return VecLeadMonomialsEnv._from_handle(h, self)

    def clone_envs(self, src, dst):
        """Copy environments src[i] over dst[i] inside the batch (tree-search node pool): state, queued ideals, RNG."""
        s = np.ascontiguousarray(src, dtype=np.int32); d = np.ascontiguousarray(dst, dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_clone_envs(self._h, len(s), _ffi.ptr(s), _ffi.ptr(d)))
        self.rows[d] = self.rows[s]

    # ---- introspection
    def stats(self):
        out = np.zeros((self.batch, 8), dtype=np.int64)
        _ffi.check(_ffi.lib().bbx_stats(self._h, _ffi.ptr(out)))
        return out

    def capacities(self):
        """Current per-environment capacities (they grow on demand unless caps['no_growth']) and how often they grew."""
        out = np.zeros(5, dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_capacities(self._h, _ffi.ptr(out)))
        return dict(zip(("max_basis", "max_pairs", "arena_terms", "max_poly_terms", "grown"), (int(v) for v in out)))

    def state(self, idx=0):
        """(basis, pairs, reducer_order): basis = list of (coefs[int32 n], exps[int32 n,8])."""
        nG, nP, nT = C.c_int32(), C.c_int32(), C.c_int32()
        _ffi.check(_ffi.lib().bbx_state_sizes(self._h, int(idx), C.byref(nG), C.byref(nP), C.byref(nT)))
        nterms = np.zeros(max(nG.value, 1), dtype=np.int32)
        coefs = np.zeros(max(nT.value, 1), dtype=np.int32)
        exps = np.zeros((max(nT.value, 1), NV), dtype=np.int32)
        pairs = np.zeros((max(nP.value, 1), 2), dtype=np.int32)
        order = np.zeros(max(nG.value, 1), dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_state_get(self._h, int(idx), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), _ffi.ptr(pairs), _ffi.ptr(order)))
        basis, at = [], 0
        for g in range(nG.value):
            basis.append((coefs[at:at + nterms[g]].copy(), exps[at:at + nterms[g]].copy()))
            at += nterms[g]
        return basis, pairs[:nP.value].copy(), order[:nG.value].copy()

    def rollout_device(self, agent, nsteps, auto_reset=True, stream=0, rewards=None, dones=None, rows=None,
                       obs=None, obs_rows=0, obs_fill=False, obs_every_step=False):
        """Asynchronous rollout on caller-owned device buffers (raw device pointers or objects with
        .data_ptr(), e.g. torch tensors); call sync() before reading results."""
        def dp(x):
            return None if x is None else C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        _ffi.check(_ffi.lib().bbx_rollout_device(self._h, _ffi.AGENTS[agent], int(nsteps), int(auto_reset), dp(rewards), dp(dones),
                                                 dp(rows), dp(obs), int(obs_rows), int(obs_fill), int(obs_every_step), C.c_void_p(int(stream))))

    def step_device(self, actions, rewards=None, dones=None, rows=None, obs=None, obs_rows=0, obs_fill=True, stream=0, auto_reset=True):
        """One asynchronous vector step with the actions taken from a device buffer (what a device-side policy wrote) and
        rewards / dones / rows / the -1-padded observation block written to device buffers: no host round trip."""
        def dp(x):
            return None if x is None else C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        fn = _ffi.lib().bbx_step_device_autoreset if auto_reset else _ffi.lib().bbx_step_device
        _ffi.check(fn(self._h, dp(actions), dp(rewards), dp(dones), dp(rows), dp(obs), int(obs_rows), int(obs_fill), C.c_void_p(int(stream))))   # obs_fill: 0 / 1 / 2 (include/bbx.h)

    def policy_step_device(self, prepared, hidden, u, actions, logprobs, rewards, dones, rows, obs, obs_rows, obs_fill=2, stream=0):
        """One asynchronous vector step with the PMLP policy in the loop (bbx_policy_step_device): the policy reads the
        block of the previous call in obs / rows, its draw is the action; obs / rows are rewritten.  `prepared`: the
        weights as bbx_pmlp_prepare left them; every buffer on the device (tensors or addresses)."""
        def dp(x):
            if x is None or isinstance(x, C.c_void_p):
                return x
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        _ffi.check(_ffi.lib().bbx_policy_step_device(self._h, dp(prepared), int(hidden), dp(u), dp(actions), dp(logprobs), dp(rewards), dp(dones),
                                                    dp(rows), dp(obs), int(obs_rows), int(obs_fill), C.c_void_p(int(stream))))

    def policy_rollout_device(self, prepared, hidden, nsteps, u, actions, logprobs, rewards=None, dones=None, rows=None, obs=None, obs_rows=0,
                              obs_step_stride=0, stream=0):
        """nsteps vector steps in one launch with the PMLP policy inside the step kernel (bbx_policy_rollout_device): all
        per-step arrays are [nsteps, batch] device buffers; obs (optional) receives the observation of step t at element
        offset t * obs_step_stride.  Raises BbxError(BBX_E_UNSUPPORTED) where the batch's kernel class has no built-in
        policy."""
        def dp(x):
            if x is None or isinstance(x, C.c_void_p):
                return x
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        _ffi.check(_ffi.lib().bbx_policy_rollout_device(self._h, dp(prepared), int(hidden), int(nsteps), dp(u), dp(actions), dp(logprobs), dp(rewards),
                                                       dp(dones), dp(rows), dp(obs), int(obs_rows), int(obs_step_stride), C.c_void_p(int(stream))))

    def sync(self):
        _ffi.check(_ffi.lib().bbx_sync(self._h))

    def persistent(self, enable=True):
        """Persistent sessions (bbx_persistent): asynchronous rollout_device calls queued behind each other feed ONE running
        kernel through a device-visible step counter instead of becoming a kernel each; environments never wait for each
        other between calls.  Outputs are ready for the caller's stream after join(stream) or sync()."""
        _ffi.check(_ffi.lib().bbx_persistent(self._h, int(enable)))

    def session_stats(self):
        out = np.zeros(5, dtype=np.int64)
        _ffi.check(_ffi.lib().bbx_session_stats(self._h, _ffi.ptr(out)))
        return dict(zip(("sessions", "joined", "later_kernel_steps", "kernels", "spills"), (int(v) for v in out)))

    def kernels_launched(self):
        """Step / reset / observation kernels launched for this batch so far (bbx_kernels_launched)."""
        out = np.zeros(1, dtype=np.int64)
        _ffi.check(_ffi.lib().bbx_kernels_launched(self._h, _ffi.ptr(out)))
        return int(out[0])

    def join(self, stream=0):
        """Device-side end of the persistent session in flight: `stream` waits for it (the host does not)."""
        _ffi.check(_ffi.lib().bbx_join(self._h, C.c_void_p(int(stream))))

    def graph_replayed(self, stream=0):
        """After replaying a HIP graph that holds recorded step_device / rollout_device / policy_* calls of this batch
        (bbx_graph_replayed): the work is in flight on `stream`; the next sync() waits for it and reports its errors."""
        _ffi.check(_ffi.lib().bbx_graph_replayed(self._h, C.c_void_p(int(stream))))

    def accounting(self, enable):
        """Toggle per-step algorithmic-byte accounting (stats()[:, 6]); off selects the leanest kernel."""
        _ffi.check(_ffi.lib().bbx_accounting(self._h, int(enable)))

    def prefetch(self):
        """Generate and upload ideals until every environment's ring is full."""
        _ffi.check(_ffi.lib().bbx_prefetch(self._h))

    def timing(self, enable=True):
        """(kernel milliseconds, launches) of the step kernel since the last call (HIP events on its stream)."""
        ms, n = C.c_double(), C.c_int32()
        _ffi.check(_ffi.lib().bbx_timing(self._h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reduced_basis(self, idx=0):
        """interreduce(minimalize(G)) of environment idx (the reduced Groebner basis once its episode is over), as a list
        of term lists [(coef, exps8), ...]."""
        n, nt = C.c_int32(), C.c_int32()
        _ffi.check(_ffi.lib().bbx_reduced_basis(self._h, int(idx), C.byref(n), C.byref(nt), None, None, None))
        nterms = np.zeros(max(n.value, 1), dtype=np.int32)
        coefs = np.zeros(max(nt.value, 1), dtype=np.int32)
        exps = np.zeros((max(nt.value, 1), NV), dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_reduced_basis(self._h, int(idx), C.byref(n), C.byref(nt), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps)))
        out, at = [], 0
        for g in range(n.value):
            out.append([(int(coefs[at + t]), tuple(int(x) for x in exps[at + t])) for t in range(nterms[g])])
            at += nterms[g]
        return out

    def trace_enable(self, capacity):
        _ffi.check(_ffi.lib().bbx_trace_enable(self._h, int(capacity)))

    def trace_read(self, env, first, count):
        out = np.zeros(count, dtype=_ffi.TRACE_DTYPE)
        _ffi.check(_ffi.lib().bbx_trace_read(self._h, int(env), int(first), int(count), _ffi.ptr(out)))
        return out


class CLeadMonomialsEnv:
    """Drop-in for deepgroebner.wrapped.CLeadMonomialsEnv (wrapped.pyx:11-38): one environment,
    always Gebauer-Moeller / additions like the C++ class (the Cython class ignores those two
    arguments, wrapped.pyx:16 -> buchberger.cpp:377)."""

    _mimic = "cpp"

    def __init__(self, ideal_dist="3-20-10-weighted", elimination="gebauermoeller", rewards="additions",
                 sort_input=False, sort_reducers=True, k=1, device=0, caps=None, _vec=None):
        if _vec is not None:
            self._vec = _vec
            return
        if self._mimic == "cpp":
            elimination, rewards = "gebauermoeller", "additions"
        self._vec = VecLeadMonomialsEnv(ideal_dist, 1, elimination, rewards, sort_input, sort_reducers, k, device, caps, self._mimic)
        # (no algorithmic-byte counting behind the gym surface — the reference has nothing like it —: the lean kernels, and
        # with them the resident kernel that serves a loop of step() calls through a host mailbox, bbx_api.cpp mbox_step)
        self._vec.accounting(False)

    def reset(self):
        return self._vec.reset()[0]

    def step(self, action):
        obs, r, d = self._vec._step_one(int(action))  # accepts numpy / python scalars like action.numpy()
        return obs, r, d, {}

    def seed(self, seed=None):
        if seed is not None:
            self._vec.seed([seed])

    def value(self, strategy="degree", gamma=0.99):
        return self._vec.value(0, strategy, gamma)

    def copy(self):
        return type(self)(_vec=self._vec.copy())


class LeadMonomialsEnv(CLeadMonomialsEnv):
    """Drop-in for the pure-Python deepgroebner.buchberger.LeadMonomialsEnv (buchberger.py:448-542):
    honours elimination/rewards, accepts a FixedIdealGenerator, observation width = ring variables."""

    _mimic = "python"

    def value(self, gamma=0.99):
        return self._vec.value(0, "degree", gamma)


P_MOD = 32003


def _monic(coefs, exps, n):
    """Terms of a basis element the way the Python reference keeps them: divided by the lead coefficient
    (f.monic(), buchberger.py:342,363), exponent tuples of the ring's n variables."""
    inv = pow(int(coefs[0]), -1, P_MOD)
    return [(int(c) * inv % P_MOD, tuple(int(x) for x in e[:n])) for c, e in zip(coefs, exps)]


class BuchbergerEnv:
    """Drop-in for the reference's pure-Python deepgroebner.buchberger.BuchbergerEnv (buchberger.py:243-394): the
    low-level surface underneath LeadMonomialsEnv.

        reset() -> (G, P)                          step((i, j)) -> ((G, P), reward, done, {})

    G is the list of basis polynomials in insertion order, each a list of (coefficient, exponent-tuple) terms in
    descending grevlex order, MONIC like the reference's (the device keeps the C++ path's non-monic elements; the two
    differ by the unit 1/LC); P is the list of pairs (i, j) in the reference's order.  The state lives
    on the device: one environment of a libbbx batch.  Only grevlex over GF(32003) exists on the device path."""

    def __init__(self, ideal_dist="3-20-10-uniform", elimination="gebauermoeller", rewards="additions",
                 sort_input=False, sort_reducers=True, device=0, caps=None):
        self._vec = VecLeadMonomialsEnv(ideal_dist, 1, elimination, rewards, sort_input, sort_reducers, 1, device, caps, "python")
        self.elimination, self.rewards = elimination, rewards
        self.sort_input, self.sort_reducers = sort_input, sort_reducers
        self.nvars = self._vec.nvars
        self.G, self.P = [], []

    def _sync(self, fresh):
        basis, pairs, _ = self._vec.state(0)
        if fresh:
            self.G = []
        for c, e in basis[len(self.G):]:                  # (a step appends at most one element)
            self.G.append(_monic(c, e, self.nvars))
        self.P = [(int(i), int(j)) for i, j in pairs]

    def reset(self):
        """New ideal from the generator (redrawn while its pair set is empty, buchberger.py:353) -> (G, P)."""
        self._vec.reset()
        self._sync(True)
        return self.G, self.P

    def step(self, action):
        """One S-polynomial reduction of the pair `action` = (i, j), which must be in P (list.remove semantics)."""
        i, j = action
        idx = self.P.index((int(i), int(j)))              # ValueError when absent, like self.P.remove(action)
        _, r, _, _ = self._vec.step(np.array([idx], dtype=np.int32))
        self._sync(False)
        return (self.G, self.P), float(r[0]), len(self.P) == 0, {}

    def seed(self, seed=None):
        if seed is not None:
            self._vec.seed([seed])

    def value(self, gamma=0.99):
        return self._vec.value(0, "degree", gamma)

    def copy(self):
        other = BuchbergerEnv.__new__(BuchbergerEnv)
        other.__dict__.update(self.__dict__)
        other._vec = self._vec.copy()
        other.G, other.P = [list(g) for g in self.G], list(self.P)
        return other


# ---- the reference's free functions (buchberger.py:11-240; buchberger.cpp:18-122), on the device ----------------------------
# Polynomials are term lists [(coefficient, exponent tuple), ...] in descending grevlex order over GF(32003), the form
# BuchbergerEnv hands out.  Every call marshals its operands into device-resident lists (bbx_alg_*, include/bbx.h) and runs
# the kernels of csrc/bbx_algebra.hip; the *_many forms take a batch of independent problems per launch.

class PolyLists:
    """A batch of independent polynomial lists (std::vector<Polynomial>) on the device."""

    def __init__(self, lists, device=0):
        self.n = len(lists)
        self._h = C.c_void_p()
        npolys = np.array([len(L) for L in lists], dtype=np.int32)
        nterms = np.array([len(f) for L in lists for f in L], dtype=np.int32)
        coefs = np.array([int(c) for L in lists for f in L for c, _ in f], dtype=np.int32)
        exps = np.zeros((max(len(coefs), 1), NV), dtype=np.int32)
        r = 0
        for L in lists:
            for f in L:
                for _, e in f:
                    exps[r, :len(e)] = e
                    r += 1
        _ffi.check(_ffi.lib().bbx_alg_create(int(device), self.n, _ffi.ptr(npolys), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), C.byref(self._h)))

    @classmethod
    def from_envs(cls, vec_env, envs=None):
        """The bases of environments of a VecLeadMonomialsEnv (None: all) as device-resident lists, copied on the device."""
        self = cls.__new__(cls)
        idx = np.arange(vec_env.batch, dtype=np.int32) if envs is None else np.ascontiguousarray(envs, dtype=np.int32)
        self.n = len(idx)
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().bbx_alg_from_envs(vec_env._h, self.n, _ffi.ptr(idx), C.byref(self._h)))
        return self

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().bbx_alg_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _ij(self, pairs):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(pairs, dtype=np.int32), (self.n, 2)))

    def binop(self, op, ij):
        """op in {'add', 'sub', 'mul', 'spoly'} of elements ij = (i, j) (one pair for all lists or one per list): appended."""
        _ffi.check(_ffi.lib().bbx_alg_binop(self._h, {"add": 0, "sub": 1, "mul": 2, "spoly": 3}[op], _ffi.ptr(self._ij(ij))))

    def reduce(self, g_and_nF):
        steps = np.zeros(self.n, dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_alg_reduce(self._h, _ffi.ptr(self._ij(g_and_nF)), _ffi.ptr(steps)))
        return steps

    def update(self, pairs, elimination="gebauermoeller"):
        """pairs: one list of (i, j) per polynomial list -> the new pair lists."""
        npairs = np.array([len(P) for P in pairs], dtype=np.int32)
        flat = np.array([x for P in pairs for p in P for x in p], dtype=np.int32).reshape(-1, 2)
        sizes = np.zeros(self.n, dtype=np.int32); tot = np.zeros(self.n, dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_alg_sizes(self._h, _ffi.ptr(sizes), _ffi.ptr(tot)))
        cap = int(npairs.sum() + sizes.sum()) + 1
        nout = np.zeros(self.n, dtype=np.int32); out = np.zeros((cap, 2), dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_alg_update(self._h, _ffi.ELIMINATION[elimination], _ffi.ptr(npairs), _ffi.ptr(flat) if len(flat) else None,
                                             _ffi.ptr(nout), _ffi.ptr(out), cap))
        res, at = [], 0
        for k in range(self.n):
            res.append([(int(i), int(j)) for i, j in out[at:at + nout[k]]]); at += int(nout[k])
        return res

    def minimalize(self):
        _ffi.check(_ffi.lib().bbx_alg_minimalize(self._h))

    def interreduce(self):
        _ffi.check(_ffi.lib().bbx_alg_interreduce(self._h))

    def get(self, k):
        """List k as term lists (exponent tuples of all 8 slots)."""
        sizes = np.zeros(self.n, dtype=np.int32); tot = np.zeros(self.n, dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_alg_sizes(self._h, _ffi.ptr(sizes), _ffi.ptr(tot)))
        nterms = np.zeros(max(int(sizes[k]), 1), dtype=np.int32)
        coefs = np.zeros(max(int(tot[k]), 1), dtype=np.int32); exps = np.zeros((max(int(tot[k]), 1), NV), dtype=np.int32)
        _ffi.check(_ffi.lib().bbx_alg_get(self._h, int(k), _ffi.ptr(nterms), _ffi.ptr(coefs), _ffi.ptr(exps), None))
        out, at = [], 0
        for g in range(int(sizes[k])):
            out.append([(int(coefs[at + t]), tuple(int(x) for x in exps[at + t])) for t in range(int(nterms[g]))])
            at += int(nterms[g])
        return out


def _nv(*polys):
    return max([len(e) for f in polys for _, e in f] + [1])


def _cut(f, n):
    return [(c, e[:n]) for c, e in f]


def spoly_many(pairs, device=0):
    """[(f, g), ...] -> [spoly(f, g), ...], one launch."""
    L = PolyLists([[f, g] for f, g in pairs], device)
    L.binop("spoly", (0, 1))
    return [_cut(L.get(k)[2], _nv(*pairs[k])) for k in range(len(pairs))]


def spoly(f, g, lmf=None, lmg=None):
    """The s-polynomial of f and g (buchberger.py:11-19; buchberger.cpp:18-21: both are divided by their lead coefficients)."""
    return spoly_many([(f, g)])[0]


def reduce_many(problems, device=0):
    """[(g, F), ...] -> [(remainder, {'steps': s}), ...], one launch."""
    L = PolyLists([list(F) + [g] for g, F in problems], device)
    steps = L.reduce([(len(F), len(F)) for _, F in problems])
    return [(_cut(L.get(k)[-1], _nv(g, *F)), {"steps": int(steps[k])}) for k, (g, F) in enumerate(problems)]


def reduce(g, F, lmF=None):
    """Remainder and stats when g is divided by the polynomials F, first divisor in list order (buchberger.py:22-67;
    buchberger.cpp:24-49) -> (r, {'steps': successful reductions})."""
    return reduce_many([(g, F)])[0]


def update(G, P, f, strategy="gebauermoeller", lmG=None):
    """The updated lists of polynomials and pairs when f is added to the basis G; G and P are modified, like the reference's
    (buchberger.py:70-147; buchberger.cpp:52-99)."""
    if strategy not in _ffi.ELIMINATION:
        raise ValueError("unknown elimination strategy")
    L = PolyLists([list(G) + [f]])
    P[:] = L.update([list(P)], strategy)[0]
    G.append(f)
    return G, P


def minimalize(G):
    """A minimal Groebner basis from the Groebner basis G (buchberger.py:150-157; buchberger.cpp:102-111)."""
    if not G:
        return []
    L = PolyLists([list(G)])
    L.minimalize()
    return [_cut(f, _nv(*G)) for f in L.get(0)]


def interreduce(G):
    """The reduced Groebner basis from the minimal Groebner basis G (buchberger.py:160-166; buchberger.cpp:114-122)."""
    if not G:
        return []
    L = PolyLists([list(G)])
    L.interreduce()
    return [_cut(f, _nv(*G)) for f in L.get(0)]


def buchberger(F, S=None, elimination="gebauermoeller", rewards="additions", sort_reducers=True, gamma=0.99, selection="degree",
               sort_input=False, device=0):
    """The reduced Groebner basis of the ideal generated by F by Buchberger's algorithm and its statistics
    (buchberger.py:169-240 — Degree selection there; buchberger.cpp:125-266 for the others): the run is a device rollout
    to completion, minimalize / interreduce of the final basis run on the device too.  S (a partially processed pair set)
    is not supported: the environments start from the generators."""
    if S is not None:
        raise NotImplementedError("buchberger(F, S): start from the generators (S=None)")
    if selection not in ("first", "degree", "normal", "sugar"):
        raise ValueError("buchberger(): selection must be first, degree, normal or sugar (strategy_stats covers the others)")
    F = [list(f) for f in F]
    if not F:
        return [], {"zero_reductions": 0, "nonzero_reductions": 0, "polynomial_additions": 0, "total_reward": 0.0, "discounted_return": 0.0}
    env = VecLeadMonomialsEnv([F], batch=1, elimination=elimination, rewards=rewards, sort_input=sort_input, sort_reducers=sort_reducers,
                              k=1, device=device)
    env.reset()
    ret = env.value(0, selection, gamma) if int(env.rows[0]) > 0 else 0.0     # (a rollout from a clone: the discounted return)
    if int(env.rows[0]) > 0:
        env.rollout(selection if selection != "random" else "random", 1 << 30, auto_reset=False)
    st = env.stats()[0]
    n = _nv(*F)
    zero, steps, adds = int(st[3]), int(st[0]), int(st[1])
    stats = {"zero_reductions": zero, "nonzero_reductions": steps - zero, "polynomial_additions": adds,
             "total_reward": float(-adds if rewards == "additions" else -steps), "discounted_return": float(ret)}
    L = PolyLists.from_envs(env, [0])                       # minimalize / interreduce of the final basis, device to device
    L.minimalize(); L.interreduce()
    return [_cut(f, n) for f in L.get(0)], stats


def _pair_key_columns(G, P, strategy):
    """Integer / float key columns of every pair of P under one selection strategy, most significant column first.
    'first': (j, i).  'degree': total degree of lcm(LM_i, LM_j).  'normal': that lcm in grevlex order — larger degree is
    larger, and among equal degrees the monomial whose LAST differing exponent is smaller is larger, i.e. ascending order of
    (degree, -e[n-1], ..., -e[0]).  'random': one uniform draw per pair."""
    ij = np.asarray([(int(p[0]), int(p[1])) for p in P], dtype=np.int64).reshape(len(P), 2)
    if strategy == "first":
        return ij[:, ::-1].astype(np.float64)
    if strategy == "random":
        return np.random.rand(len(P), 1)
    if strategy not in ("normal", "degree"):
        raise ValueError("unknown selection strategy")
    width = max(len(f[0][1]) for f in G)
    lead = np.zeros((len(G), width), dtype=np.int64)
    for r, f in enumerate(G):
        e = f[0][1]
        lead[r, :len(e)] = e
    lcm = np.maximum(lead[ij[:, 0]], lead[ij[:, 1]])
    deg = lcm.sum(axis=1, keepdims=True)
    if strategy == "degree":
        return deg.astype(np.float64)
    return np.concatenate([deg, -lcm[:, ::-1]], axis=1).astype(np.float64)


def select(G, P, strategy="normal"):
    """The pair of P that a selection strategy — or a list of strategies, later ones breaking the ties of earlier ones —
    ranks first; among pairs that tie under all of them, the one that comes first in P (what `min` over P gives the
    reference's select, buchberger.py:415-439).  G: term lists as BuchbergerEnv returns them; P: pairs (i, j)."""
    if len(G) == 0 or len(P) == 0:
        raise AssertionError("polynomial list must be nonempty" if len(G) == 0 else "pair set must be nonempty")
    names = [strategy] if isinstance(strategy, str) else list(strategy)
    keys = np.concatenate([_pair_key_columns(G, P, s) for s in names], axis=1)
    alive = np.arange(len(P))
    for c in range(keys.shape[1]):                          # lexicographic minimum: column by column among the survivors
        col = keys[alive, c]
        alive = alive[col == col.min()]
        if len(alive) == 1:
            break
    return P[int(alive[0])]


class BuchbergerAgent:
    """Agent over (G, P) states that always takes select()'s pair (reference BuchbergerAgent, buchberger.py:397-412)."""

    def __init__(self, selection="normal"):
        self.strategy = selection

    def act(self, state):
        return select(state[0], state[1], strategy=self.strategy)


def lead_monomials_vector(f, n, k=2, dtype=np.int32):
    """Concatenated exponent vectors of the k lead monomials of the term list f in n variables, zero padded
    (reference lead_monomials_vector, buchberger.py:442-445; the second argument is the ring there, its number of
    variables here)."""
    n = getattr(n, "ngens", n)
    rows = [tuple(e[:n]) + (0,) * (n - len(e[:n])) for _, e in f[:k]]
    rows += [(0,) * n] * (k - len(rows))
    return np.array(rows).flatten().astype(dtype)


class LeadMonomialsAgent:
    """Agent over the lead-monomial matrix, one row per pair (reference LeadMonomialsAgent, buchberger.py:545-567): 'first'
    takes row 0, 'degree' the first row whose two LEAD monomials (the first of the k per half) have the lcm of least
    degree, 'random' a uniformly drawn row.  Rows are in P's order, so 'first' / 'degree' pick what the device selectors
    BBX_AGENT_FIRST / BBX_AGENT_DEGREE pick (ties resolve to the earliest row on both sides)."""

    _RULES = {
        "first": lambda self, state: 0,
        "degree": lambda self, state: int(self._lcm_degrees(state).argmin()),
        "random": lambda self, state: np.random.choice(len(state)),
    }

    def __init__(self, selection="degree", k=1):
        self.strategy = selection
        self.k = k

    def _lcm_degrees(self, state):
        half = state.shape[1] // 2                          # columns of one pair member: k monomials of n exponents
        n = half // self.k
        return np.maximum(state[:, :n], state[:, half:half + n]).sum(axis=1)

    def act(self, state):
        rule = self._RULES.get(self.strategy)
        return None if rule is None else rule(self, state)   # (the reference falls through to None for anything else)


def strategy_stats(ideals, strategy="degree", elimination="gebauermoeller", sort_reducers=True, device=0, caps=None,
                   seed=None):
    """Full Buchberger runs of one selection strategy over a list of ideals, all at once on the GPU: the columns
    scripts/make_strat.cpp:44-70 of the reference writes per ideal.  Returns int64 [len(ideals), 3] =
    (ZeroReductions, NonzeroReductions, PolynomialAdditions).  strategy is one of first / degree / normal / sugar /
    random / last / codegree / strange / spice; "random" with a seed is the reference's seeded Random selection (the
    same seed for every ideal, make_strat.cpp:66), without one it is the counter-hash agent (the reference then seeds
    from std::random_device, so there is nothing to reproduce)."""
    env = VecLeadMonomialsEnv(list(ideals), batch=len(ideals), elimination=elimination, sort_reducers=sort_reducers,
                              k=1, device=device, caps=caps)
    if strategy == "random" and seed is not None:
        strategy = "random_std"
        env.seed_strategy(int(seed))
    env.reset()
    if int(env.rows.max()) > 0:
        env.rollout(strategy, 1 << 30, auto_reset=False)
    st = env.stats()
    return np.stack([st[:, 3], st[:, 0] - st[:, 3], st[:, 1]], axis=1)
