"""The consumer side of the batched environment: policy, trajectory buffer and rollout loop, all on the
device, replacing the per-step host round trip of the reference's agents.

    PMLPPolicy                 ParallelMultilayerPerceptron           networks.py:522-571 (:49-95, :414-460)
    discount_rewards, compute_advantages                              pg.py:20-76
    DeviceTrajectoryBuffer     TrajectoryBuffer (store/finish/get)    pg.py:79-240
    run_rollout                PGAgent.run_episode(s)                 pg.py:451-503   (one library call per vector step)
    run_rollout_fused          the same with the policy INSIDE the step kernel, 256 vector steps per launch

PyTorch is the plumbing here (device memory, autograd for training); the policy evaluation + sampling of the default
one-hidden-layer network runs in hand-written HIP: on the matrix cores, either as a kernel of its own fed by the padded
observation block (bbx_pmlp_act), in front of the step in the same launch (bbx_policy_step_device), or inside the step
loop (bbx_policy_rollout_device); deeper networks take the torch path.  Actions never visit the host.
"""
import contextlib
import ctypes as C
import functools
import gc

import numpy as np
import torch

from . import _ffi


@contextlib.contextmanager
def _gc_paused():
    """The step loops below enqueue a few kernels per iteration and must not fall behind the device: a generation-2 pass of
    Python's cyclic collector over a process that has torch loaded takes ~40 ms (measured: one in ~900 iterations, +40 us per
    vector step averaged over a 1000-step rollout), so collection is paused for the duration of a rollout."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _with_gc_paused(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with _gc_paused():
            return fn(*args, **kwargs)
    return wrapper


def discount_rewards(rewards, gam):
    """Discounted rewards-to-go (pg.py:20-47), float64."""
    out = np.zeros(len(rewards), dtype=np.float64)
    c = 0.0
    for i in reversed(range(len(rewards))):
        c = rewards[i] + gam * c
        out[i] = c
    return out


def compute_advantages(rewards, values, gam, lam):
    """Generalized advantage estimates of one complete trajectory (pg.py:50-76), float64."""
    rewards = np.array(rewards, dtype=np.float64)
    values = np.array(values, dtype=np.float64)
    delta = rewards - values
    delta[:-1] += gam * values[1:]
    return discount_rewards(delta, gam * lam)


class PMLPPolicy(torch.nn.Module):
    """ParallelMultilayerPerceptron(hidden_layers) of the reference: every row of the -1-padded [batch, rows, cols] int
    block is embedded by the same MLP (relu), scored by one linear unit, padded rows get -1e9, log-softmax over rows."""

    def __init__(self, cols, hidden_layers=(128,)):
        super().__init__()
        dims = [cols] + list(hidden_layers)
        self.embedding = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.deciding = torch.nn.Linear(dims[-1], 1)
        self.cols = cols

    def forward(self, batch):
        """int [B, R, cols] with -1 padding -> log-probabilities float32 [B, R] (about -1e9 on padded rows)."""
        mask = batch[:, :, -1] != -1                                  # ParallelEmbeddingLayer.compute_mask
        x = batch.to(torch.float32)
        for layer in self.embedding:
            x = torch.relu(layer(x))
        x = self.deciding(x).squeeze(-1)
        x = x + (~mask).to(torch.float32) * -1e9
        return torch.log_softmax(x, dim=-1)

    @torch.no_grad()
    def act(self, obs, rows, u, actions=None, logprobs=None, stream=None):
        """Sample one action per environment by inverse CDF from the uniforms u [B] -> (actions int32 [B], logprobs
        float32 [B]) on the device.  One hidden layer: the fused HIP kernel (bbx_pmlp_act); two or three hidden layers of at
        most 128 units: bbx_pmlp2_act / bbx_pmlp3_act; otherwise torch ops."""
        B, R, cols = obs.shape
        if actions is None:
            actions = torch.empty(B, dtype=torch.int32, device=obs.device)
        if logprobs is None:
            logprobs = torch.empty(B, dtype=torch.float32, device=obs.device)
        if len(self.embedding) == 1 and self.fused_ok(cols, self.embedding[0].out_features) and R <= 2048:   # (the kernels score <= 2048 rows)
            w = self._fused_weights()
            s = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
            _ffi.check(_ffi.lib().bbx_pmlp_act(C.c_void_p(obs.data_ptr()), C.c_void_p(rows.data_ptr()), B, R, cols, w["prepared"], w["hidden"],
                                               C.c_void_p(u.data_ptr()), C.c_void_p(actions.data_ptr()), C.c_void_p(logprobs.data_ptr()), C.c_void_p(s)))
            return actions, logprobs
        if self.deep_ok(cols) and R <= self.deep_max_rows():
            w = self._deep_weights()
            s = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
            fn = _ffi.lib().bbx_pmlp2_act if len(w["hidden"]) == 2 else _ffi.lib().bbx_pmlp3_act
            _ffi.check(fn(C.c_void_p(obs.data_ptr()), C.c_void_p(rows.data_ptr()), B, R, cols, w["prepared"], *w["hidden"],
                          C.c_void_p(u.data_ptr()), C.c_void_p(actions.data_ptr()), C.c_void_p(logprobs.data_ptr()), C.c_void_p(s)))
            return actions, logprobs
        return self.act_torch(obs, rows, u, actions, logprobs)

    def deep_ok(self, cols):
        """Two or three hidden layers of at most 128 units: the shapes bbx_pmlp2_act / bbx_pmlp3_act are built for."""
        return len(self.embedding) in (2, 3) and all(1 <= l.out_features <= 128 for l in self.embedding) and 1 <= cols <= 64

    def deep_max_rows(self):
        """Rows per environment the two- / three-layer kernels score: the logits of a wave's environment live in LDS beside the
        staged weights — 2048 rows, 1024 where three layers wider than 64 units (128 KB of weights) leave less room."""
        return 1024 if len(self.embedding) == 3 and max(l.out_features for l in self.embedding) > 64 else 2048

    def _deep_weights(self):
        """The two- / three-layer kernel's view of the weights (bbx_pmlp2_prepare / bbx_pmlp3_prepare), rebuilt only when a
        parameter changed, in the same buffer (a recorded graph keeps reading it)."""
        key = tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())
        c = self.__dict__.get("_deep_cache")
        if c is None or c["key"] != key:
            cols = self.embedding[0].in_features
            hidden = [l.out_features for l in self.embedding]
            t = []
            for l in self.embedding:
                t += [l.weight.detach().t().contiguous().float(), l.bias.detach().contiguous().float()]
            t += [self.deciding.weight.detach().reshape(-1).contiguous().float(), self.deciding.bias.detach().reshape(-1).contiguous().float()]
            lib = _ffi.lib()
            nfl = lib.bbx_pmlp2_prepared_floats(cols, *hidden) if len(hidden) == 2 else lib.bbx_pmlp3_prepared_floats(cols, *hidden)
            _ffi.check(min(nfl, 0))
            prep = c["keep"][0] if c is not None and c["keep"][0].numel() == nfl and c["keep"][0].device == t[0].device else \
                torch.empty(nfl, dtype=torch.float32, device=t[0].device)
            prepare = lib.bbx_pmlp2_prepare if len(hidden) == 2 else lib.bbx_pmlp3_prepare
            _ffi.check(prepare(*[C.c_void_p(x.data_ptr()) for x in t], cols, *hidden, C.c_void_p(prep.data_ptr()),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            c = {"key": key, "keep": (prep, t), "prepared": C.c_void_p(prep.data_ptr()), "hidden": hidden}
            self.__dict__["_deep_cache"] = c
        return c

    @staticmethod
    def fused_ok(cols, hidden):
        """Shapes the policy kernel is built for (bbx_pmlp_prepared_floats >= 0)."""
        return 1 <= hidden <= 256 and 1 <= cols <= 64

    def _fused_weights(self):
        """The kernels' view of the weights (bbx_pmlp_prepare: transposed, zero-padded to the tile sizes), rebuilt only when
        a parameter changed (an optimiser step bumps the tensors' version counters): no per-step transposes, no per-step
        host read of b2."""
        lin = self.embedding[0]
        key = tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())
        c = self.__dict__.get("_fused_cache")
        if c is None or c["key"] != key:
            cols, hidden = lin.in_features, lin.out_features
            w1 = lin.weight.detach().t().contiguous().float()
            b1 = lin.bias.detach().contiguous().float()
            w2 = self.deciding.weight.detach().reshape(-1).contiguous().float()
            nfl = _ffi.lib().bbx_pmlp_prepared_floats(cols, hidden)
            _ffi.check(min(nfl, 0))
            prep = torch.empty(nfl, dtype=torch.float32, device=w1.device)
            _ffi.check(_ffi.lib().bbx_pmlp_prepare(C.c_void_p(w1.data_ptr()), C.c_void_p(b1.data_ptr()), C.c_void_p(w2.data_ptr()),
                                                   C.c_float(float(self.deciding.bias.item())), cols, hidden, C.c_void_p(prep.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            c = {"key": key, "keep": prep, "prepared": C.c_void_p(prep.data_ptr()), "hidden": hidden}
            self.__dict__["_fused_cache"] = c
        return c

    def act_torch(self, obs, rows, u, actions=None, logprobs=None):
        """The same draw with torch ops (the reference of the fused kernel; any depth)."""
        lp = self.forward(obs)
        n = torch.clamp(rows, max=obs.shape[1]).to(torch.int64)
        valid = torch.arange(obs.shape[1], device=obs.device)[None, :] < n[:, None]
        p = torch.where(valid, torch.exp(lp - lp.max(dim=1, keepdim=True).values), torch.zeros_like(lp))
        cdf = torch.cumsum(p, dim=1)
        target = u * cdf[:, -1]
        a = torch.minimum((cdf <= target[:, None]).sum(dim=1), n - 1).clamp(min=0)
        out_a = a.to(torch.int32)
        out_l = lp.gather(1, a[:, None]).squeeze(1)
        if actions is not None:
            actions.copy_(out_a); out_a = actions
        if logprobs is not None:
            logprobs.copy_(out_l); out_l = logprobs
        return out_a, out_l


class DeviceTrajectoryBuffer:
    """TrajectoryBuffer (pg.py:79-240) for a batch of environments stepping in lockstep, kept on the device: per step the
    -1-padded state block [B, R, cols] (optional), action, reward, log-probability, value and done flag of every
    environment.  finish() turns rewards into discounted rewards-to-go and values into GAE advantages per EPISODE
    (episodes end where done is set; an episode still running at the end of the buffer is incomplete and left out, like
    the reference's `[:self.start]`); get() returns the training tensors with the reference's filtering (states with a
    single row are dropped) and advantage normalisation."""

    def __init__(self, nsteps, batch, gam=0.99, lam=0.97, obs_shape=None, device="cuda"):
        self.T, self.B, self.gam, self.lam = nsteps, batch, gam, lam
        self.actions = torch.zeros((nsteps, batch), dtype=torch.int32, device=device)
        self.rewards = torch.zeros((nsteps, batch), dtype=torch.float64, device=device)
        self.logprobs = torch.zeros((nsteps, batch), dtype=torch.float32, device=device)
        self.values = torch.zeros((nsteps, batch), dtype=torch.float64, device=device)
        self.dones = torch.zeros((nsteps, batch), dtype=torch.bool, device=device)
        self.rows = torch.zeros((nsteps, batch), dtype=torch.int32, device=device)
        self.states = torch.empty((nsteps, batch) + tuple(obs_shape), dtype=torch.int32, device=device) if obs_shape else None
        self.t = 0
        self.returns = self.advantages = self.complete = None

    def store(self, state, rows, action, reward, logprob, value, done):
        t = self.t
        if self.states is not None:
            self.states[t].copy_(state)
        self.rows[t].copy_(rows); self.actions[t].copy_(action); self.rewards[t].copy_(reward)
        self.logprobs[t].copy_(logprob); self.dones[t].copy_(done.to(torch.bool))
        if value is not None:
            self.values[t].copy_(value)
        self.t += 1

    def finish(self):
        """Reverse scan over time with episode boundaries: ret_t = r_t + gam ret_{t+1}; delta_t = r_t - v_t + gam v_{t+1};
        adv_t = delta_t + gam lam adv_{t+1}; nothing crosses a done flag (pg.py:20-76 per trajectory)."""
        T = self.t
        r, v, d = self.rewards[:T], self.values[:T], self.dones[:T]
        ret = torch.zeros_like(r); adv = torch.zeros_like(r); comp = torch.zeros_like(d)
        nret = torch.zeros(self.B, dtype=torch.float64, device=r.device); nadv = torch.zeros_like(nret)
        nval = torch.zeros_like(nret); ncomp = torch.zeros(self.B, dtype=torch.bool, device=r.device)
        for t in range(T - 1, -1, -1):
            last = d[t]                                              # step t ends its episode: nothing follows it
            nret = torch.where(last, torch.zeros_like(nret), nret); nadv = torch.where(last, torch.zeros_like(nadv), nadv)
            nval = torch.where(last, torch.zeros_like(nval), nval); ncomp = ncomp | last
            ret[t] = r[t] + self.gam * nret
            adv[t] = (r[t] - v[t] + self.gam * nval) + self.gam * self.lam * nadv
            comp[t] = ncomp
            nret, nadv, nval = ret[t], adv[t], v[t]
        self.returns, self.advantages, self.complete = ret, adv, comp
        return ret, adv, comp

    def get(self, batch_size=None, normalize_advantages=True, sort=False, drop_remainder=False):
        """Training data of all complete-episode steps whose state had more than one row (pg.py:162-240): (states or None,
        actions, logprobs, advantages, values-to-fit), advantages normalised by their mean and population std over the
        complete steps (pg.py:176-178: before the single-row filter, like the reference), steps ordered trajectory after
        trajectory (environment-major).  batch_size=None: one tuple of whole tensors.  Otherwise the reference's
        padded_batch: a list of such tuples of at most batch_size steps each, the state block of a batch cut to the most
        rows any of its states has (-1 padding beyond a state's own rows, as stored); sort=True orders the steps by their
        number of rows first (less padding); drop_remainder=True leaves a short last batch out."""
        if self.returns is None:
            self.finish()
        T = self.t
        em = lambda x: x[:T].transpose(0, 1)                        # [B, T, ...]: trajectory after trajectory
        comp = em(self.complete)
        adv = em(self.advantages)[comp].to(torch.float32)
        if normalize_advantages and adv.numel():
            adv = (adv - adv.mean()) / adv.std(unbiased=False)
        rows = em(self.rows)[comp]
        keep = rows != 1
        st = em(self.states)[comp][keep] if self.states is not None else None
        out = [st, em(self.actions)[comp][keep], em(self.logprobs)[comp][keep], adv[keep], em(self.returns)[comp].to(torch.float32)[keep]]
        rows = rows[keep]
        if batch_size is None and not sort:
            return tuple(out)
        if sort:
            order = torch.argsort(rows, stable=True)
            out = [None if x is None else x[order] for x in out]
            rows = rows[order]
        if batch_size is None:
            return tuple(out)
        n = int(rows.numel())
        batches = []
        for i in range(0, n, batch_size):
            j = min(i + batch_size, n)
            if drop_remainder and j - i < batch_size:
                break
            mr = int(rows[i:j].max()) if self.states is not None else 0
            batches.append(tuple(None if x is None else (x[i:j, :mr] if k == 0 else x[i:j]) for k, x in enumerate(out)))
        return batches


@_with_gc_paused
@torch.no_grad()
def run_rollout_fused(env, policy, nsteps, buffer=None, obs_rows=256, generator=None, chunk=256):
    """run_rollout with the policy INSIDE the step kernel (bbx_policy_rollout_device): `chunk` vector steps per launch,
    environments never wait for each other between steps.  Per-step outputs land in the trajectory buffer's own arrays
    (no copies).  Raises BbxError (BBX_E_UNSUPPORTED) where the batch's kernel class has no built-in policy — callers
    fall back to run_rollout.  Returns (total reward per environment, finished episodes) like run_rollout.
    The rollout kernels are the lean ones (no algorithmic-byte accounting): the handle's accounting is switched off here."""
    env.accounting(False)
    B, cols = env.batch, env.cols
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream()
    w = policy._fused_weights()
    keep_states = buffer is not None and buffer.states is not None
    if keep_states:
        obs_rows = buffer.states.shape[2]
    if buffer is None:
        act = torch.empty((chunk, B), dtype=torch.int32, device=dev); logp = torch.empty((chunk, B), dtype=torch.float32, device=dev)
    obs1 = None if keep_states else torch.empty((B, obs_rows, cols), dtype=torch.int32, device=dev)
    st0 = env.stats()
    if buffer is not None:
        # the kernel writes n x B elements behind raw pointers into the buffer's arrays: a short slice would be overrun
        if buffer.t + nsteps > buffer.T:
            raise IndexError("trajectory buffer holds %d steps, %d are stored already: no room for %d more" % (buffer.T, buffer.t, nsteps))
        if keep_states and tuple(buffer.states.shape[2:]) != (obs_rows, cols):
            raise ValueError("buffer.states has blocks of %s, the environment writes (%d, %d)" % (tuple(buffer.states.shape[2:]), obs_rows, cols))
    for t0 in range(0, nsteps, chunk):
        n = min(chunk, nsteps - t0)
        u = torch.rand((n, B), device=dev, generator=generator)
        if buffer is not None:
            t = buffer.t
            obs = buffer.states[t:t + n] if keep_states else obs1
            if keep_states:
                obs.fill_(-1)                           # the kernel writes the live rows only; the rest is the reference's padding
            env.policy_rollout_device(w["prepared"], w["hidden"], n, u, buffer.actions[t:t + n], buffer.logprobs[t:t + n], buffer.rewards[t:t + n],
                                      buffer.dones[t:t + n], buffer.rows[t:t + n], obs, obs_rows, B * obs_rows * cols if keep_states else 0,
                                      stream.cuda_stream)
            buffer.t += n
        else:
            env.policy_rollout_device(w["prepared"], w["hidden"], n, u, act, logp, None, None, None, obs1, obs_rows, 0, stream.cuda_stream)
        env.sync()
    d = env.stats() - st0
    total = torch.tensor(-d[:, 1].astype(np.float64), device=dev)
    episodes = torch.tensor(d[:, 2], device=dev)
    return total, episodes


@_with_gc_paused
@torch.no_grad()
def run_rollout(env, policy, nsteps, buffer=None, obs_rows=256, generator=None, sync_every=64, graph=False):
    """nsteps vector steps of `env` (a VecLeadMonomialsEnv already reset) under `policy`, everything on the device
    (obs_rows: rows of the observation block per environment — a pair set with more rows makes env.sync() raise
    BBX_E_CAPACITY, it is never cut silently; 256 is what the register/LDS-resident class holds):
    observation block -> policy.act (log-softmax + inverse-CDF draw) -> bbx_step_device_autoreset -> next block.  The
    host only enqueues kernels; it waits (env.sync: errors, environments that outgrew a kernel class) every
    `sync_every` steps and at the end.  Returns (total reward per environment float64 [B] — for `additions` rewards —,
    finished episodes int64 [B]).
    graph=True (policies without a fused kernel, i.e. more than one hidden layer; no buffer): the vector step — the
    policy's torch ops and the step kernels — is recorded once into a HIP graph (kept on `env`) and replayed, one graph
    launch per step instead of ~25 kernel launches from Python; same draws, same results."""
    B, cols = env.batch, env.cols
    dev = torch.device("cuda", torch.cuda.current_device())
    one_call = len(policy.embedding) == 1 and policy.fused_ok(cols, policy.embedding[0].out_features) and hasattr(env, "policy_step_device")
    if graph and not one_call and buffer is None:
        return _run_rollout_graph(env, policy, nsteps, obs_rows, generator, sync_every)
    stream = torch.cuda.current_stream()
    obs = torch.empty((B, obs_rows, cols), dtype=torch.int32, device=dev)
    rew = torch.zeros(B, dtype=torch.float64, device=dev); done = torch.zeros(B, dtype=torch.uint8, device=dev)
    rows = torch.zeros(B, dtype=torch.int32, device=dev); act = torch.zeros(B, dtype=torch.int32, device=dev)
    logp = torch.zeros(B, dtype=torch.float32, device=dev)
    st0 = env.stats()
    # the current observation: a zero-step launch writes the padded block and the row counts
    env.rollout_device("first", 0, False, stream.cuda_stream, rew, done, rows, obs, obs_rows, True, False)
    env.sync()
    keep_states = buffer is not None and buffer.states is not None
    w = policy._fused_weights() if one_call else None
    nsync = 0
    for t0 in range(0, nsteps, sync_every):
        n = min(sync_every, nsteps - t0)
        u_all = torch.rand((n, B), device=dev, generator=generator)   # one generator launch per chunk of steps
        for i in range(n):
            if buffer is not None:                      # what the policy is about to see
                t = buffer.t
                if keep_states:
                    buffer.states[t].copy_(obs)
                buffer.rows[t].copy_(rows)
            if one_call:                                # policy + step: one library call (one kernel where the class has it)
                env.policy_step_device(w["prepared"], w["hidden"], u_all[i], act, logp, rew, done, rows, obs, obs_rows, 2, stream.cuda_stream)   # (2: incremental padding)
            else:
                policy.act(obs, rows, u_all[i], act, logp, stream)
                env.step_device(act, rew, done, rows, obs, obs_rows, 2, stream.cuda_stream, auto_reset=True)
            if buffer is not None:
                buffer.actions[t].copy_(act); buffer.logprobs[t].copy_(logp)
                buffer.rewards[t].copy_(rew); buffer.dones[t].copy_(done)
                buffer.t += 1
        env.sync()
        nsync += 1
    # totals from the environments' own counters (additions rewards: reward = -additions, buchberger.cpp:328)
    d = env.stats() - st0
    total = torch.tensor(-d[:, 1].astype(np.float64), device=dev)
    episodes = torch.tensor(d[:, 2], device=dev)
    return total, episodes


def _run_rollout_graph(env, policy, nsteps, obs_rows, generator, sync_every):
    """run_rollout with the vector step replayed from a HIP graph (see there).  The graph and its buffers live on `env`
    (one per policy object and block height) and are reused by later calls; the policy's parameters are read in place, so
    optimiser steps between calls are seen."""
    B, cols = env.batch, env.cols
    dev = torch.device("cuda", torch.cuda.current_device())
    cache = env.__dict__.setdefault("_step_graphs", {})
    key = (id(policy), obs_rows, sync_every)
    c = cache.get(key)
    if c is None:
        c = {"obs": torch.empty((B, obs_rows, cols), dtype=torch.int32, device=dev),
             "rew": torch.zeros(B, dtype=torch.float64, device=dev), "done": torch.zeros(B, dtype=torch.uint8, device=dev),
             "rows": torch.ones(B, dtype=torch.int32, device=dev), "act": torch.zeros(B, dtype=torch.int32, device=dev),
             "logp": torch.zeros(B, dtype=torch.float32, device=dev),
             "u": torch.zeros((sync_every, B), dtype=torch.float32, device=dev), "i": torch.zeros(1, dtype=torch.int64, device=dev),
             "policy": policy}
        c["obs"].fill_(-1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # the library GEMMs pick their workspaces outside the capture
            for _ in range(3):
                policy.act(c["obs"], c["rows"], c["u"][0], c["act"], c["logp"], side)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        env.sync()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cs = torch.cuda.current_stream()
            u = c["u"].index_select(0, c["i"]).squeeze(0)
            policy.act(c["obs"], c["rows"], u, c["act"], c["logp"], cs)
            env.step_device(c["act"], c["rew"], c["done"], c["rows"], c["obs"], obs_rows, 2, cs.cuda_stream, auto_reset=True)
            c["i"].add_(1)
        c["graph"] = g
        cache[key] = c
    stream = torch.cuda.current_stream()
    if policy.deep_ok(cols) and obs_rows <= policy.deep_max_rows():
        policy._deep_weights()                       # (weights changed since the recording: the prepared copy is refilled in place)
    st0 = env.stats()
    env.rollout_device("first", 0, False, stream.cuda_stream, c["rew"], c["done"], c["rows"], c["obs"], obs_rows, True, False)
    env.sync()
    for t0 in range(0, nsteps, sync_every):
        n = min(sync_every, nsteps - t0)
        c["u"][:n].copy_(torch.rand((n, B), device=dev, generator=generator))
        c["i"].zero_()
        for _ in range(n):
            c["graph"].replay()
        try:
            env.graph_replayed(stream.cuda_stream)
            env.sync()                                 # (records that were enlarged inside the replays are reported here)
        except _ffi.BbxError:
            # the records live at a new address now, or an environment sat out part of the chain: the recording steps a
            # retired copy from here on — dropped, the next call records again
            cache.pop(key, None)
            raise
    d = env.stats() - st0
    return torch.tensor(-d[:, 1].astype(np.float64), device=dev), torch.tensor(d[:, 2], device=dev)
