"""Latency of the single-environment drop-in (what scripts/train.py of the reference drives): env.step(action) per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import CLeadMonomialsEnv
env = CLeadMonomialsEnv("3-20-10-weighted", k=2)
env.seed(123)
s = env.reset()
for _ in range(600):                        # (one-time costs out of the way: streams, pinned blocks, code objects of every kernel variant)
    s, r, d, _ = env.step(0)
    if d:
        s = env.reset()
n = 0; t0 = time.perf_counter()
while n < 20000:
    s, r, d, _ = env.step(0)
    n += 1
    if d:
        s = env.reset()
t1 = time.perf_counter()
print("single-environment drop-in: %.1f us per step+observation, %.0f steps/s" % ((t1 - t0) / n * 1e6, n / (t1 - t0)), env._vec.session_stats())
import ctypes as C
from deepgroebner_amd import _ffi
L = _ffi.lib(); v = env._vec
act = np.zeros(1, dtype=np.int32); obs_p, off_p = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
args = (v._h, _ffi.ptr(act), 1, _ffi.ptr(v._rewards), _ffi.ptr(v._dones), _ffi.ptr(v.rows), C.byref(obs_p), C.byref(off_p))
t0 = time.perf_counter()
for _ in range(20000):
    L.bbx_step_obs(*args)
t1 = time.perf_counter()
print("  of which the library call itself (bbx_step_obs, auto-reset): %.1f us" % ((t1 - t0) / 20000 * 1e6))
env = CLeadMonomialsEnv("3-20-10-weighted", k=2); env.seed(123); env.reset()
n = 0; t0 = time.perf_counter()
while n < 300:
    val = env.value("degree", 0.99)
    s, r, d, _ = env.step(0); n += 1
    if d:
        env.reset()
t1 = time.perf_counter()
print("value('degree') + step, as pg.py:461-465 calls them with --value_model degree: %.1f us per step" % ((t1 - t0) / n * 1e6))
st = env._vec.session_stats() if hasattr(env._vec, "session_stats") else None
print("  sessions of the value()+step loop's handle:", st)
