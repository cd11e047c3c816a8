"""PCIe-inclusive rate of the Gym-style drop-in path: host actions in, padded observation block out, every step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
import ctypes as C
B, T = 4096, 200
env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
env.seed(np.arange(B) + 1000); env.reset()
L = _ffi.lib()
acts = np.zeros(B, dtype=np.int32); rew = np.zeros(B); done = np.zeros(B, dtype=np.uint8); rows = env.rows
obs = np.empty((B, 128, env.cols), dtype=np.int32)
rng = np.random.default_rng(0)
t0 = time.perf_counter()
for t in range(T):
    acts[:] = (rng.random(B) * np.maximum(rows, 1)).astype(np.int32)
    _ffi.check(L.bbx_step(env._h, _ffi.ptr(acts), _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(rows)))
    _ffi.check(L.bbx_obs(env._h, _ffi.ptr(obs), 128, 1))
    if done.any():
        _ffi.check(L.bbx_reset(env._h, _ffi.ptr(done), _ffi.ptr(rows)))
t1 = time.perf_counter()
print("gym path: %.3f ms per vector step of %d envs incl. padded obs block D2H (%.1f MB) -> %.1f M env-steps/s" % ((t1 - t0) / T * 1e3, B, obs.nbytes / 1e6, B * T / (t1 - t0) / 1e6))
t0 = time.perf_counter()
for t in range(T):
    acts[:] = 0
    _ffi.check(L.bbx_step(env._h, _ffi.ptr(acts), _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(rows)))
    if done.any():
        _ffi.check(L.bbx_reset(env._h, _ffi.ptr(done), _ffi.ptr(rows)))
t1 = time.perf_counter()
print("gym path without the observation copy: %.3f ms per vector step -> %.1f M env-steps/s" % ((t1 - t0) / T * 1e3, B * T / (t1 - t0) / 1e6))
t0 = time.perf_counter()
for t in range(T):
    acts[:] = 0
    _ffi.check(L.bbx_step_autoreset(env._h, _ffi.ptr(acts), _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(rows)))
t1 = time.perf_counter()
print("gym path, auto-reset, no observation copy: %.3f ms per vector step -> %.1f M env-steps/s" % ((t1 - t0) / T * 1e3, B * T / (t1 - t0) / 1e6))
t0 = time.perf_counter()
for t in range(T):
    acts[:] = 0
    _ffi.check(L.bbx_step_autoreset(env._h, _ffi.ptr(acts), _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(rows)))
    _ffi.check(L.bbx_obs(env._h, _ffi.ptr(obs), 128, 1))
t1 = time.perf_counter()
print("gym path, auto-reset, with padded observation block: %.3f ms per vector step -> %.1f M env-steps/s" % ((t1 - t0) / T * 1e3, B * T / (t1 - t0) / 1e6))
t0 = time.perf_counter()
for t in range(T):
    acts[:] = 0
    flat, off, r, d = env.step_ragged(acts, auto_reset=True)
t1 = time.perf_counter()
print("python step_ragged (fused step + ragged observation, %.1f MB): %.3f ms per vector step -> %.1f M env-steps/s" % (flat.nbytes / 1e6, (t1 - t0) / T * 1e3, B * T / (t1 - t0) / 1e6))
t0 = time.perf_counter()
for t in range(20):
    obs_list, r, d, _ = env.step(acts, auto_reset=True)
t1 = time.perf_counter()
print("python step (list of %d per-environment matrices): %.3f ms per vector step" % (B, (t1 - t0) / 20 * 1e3))
obs_p, off_p = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
t0 = time.perf_counter()
for t in range(T):
    _ffi.check(L.bbx_step_obs(env._h, _ffi.ptr(acts), 1, _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(rows), C.byref(obs_p), C.byref(off_p)))
t1 = time.perf_counter()
print("bbx_step_obs (C ABI only: step + ragged observation into pinned memory): %.3f ms per vector step -> %.1f M env-steps/s" % ((t1 - t0) / T * 1e3, B * T / (t1 - t0) / 1e6))
