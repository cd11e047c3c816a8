"""Experiment: timeline of a few waves of the step kernel ({shader-clock counter, 100 MHz counter} every 1024 steps), normal
launches or one persistent session.  Library built with -DBBX_DRIFT_DEBUG.   python scripts/exp_drift.py MODE TOTAL K"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgroebner_amd import VecLeadMonomialsEnv, _ffi
mode, total, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
B, R = 4096, 256
env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
obs = torch.empty((B, R, env.cols), dtype=torch.int32, device="cuda")
rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda"); rows = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream()
env.persistent(True); env.persistent(mode == "persistent")          # (allocates the block the samples go to)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
ev0.record(s)
for i in range(total // K):
    env.rollout_device("random", K, True, s.cuda_stream, rew, done, rows, obs, R, False, True)
if os.environ.get("JOIN"):
    env.join(s.cuda_stream)
ev1.record(s)
t_issue = time.perf_counter() - t0
def cpustat():
    try:
        return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat")}
    except Exception:
        return {}
c0 = cpustat()
if os.environ.get("SPIN"):                                 # host polls the event itself instead of blocking in the runtime
    while not ev1.query():
        pass
    t_spin = time.perf_counter() - t0
    print("event seen complete after %.1f ms" % (t_spin * 1e3))
env.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
c1 = cpustat()
print("cgroup during the wait: throttled %d times, %.1f ms; cpu used %.1f ms" % (c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0), (c1.get("throttled_usec", 0) - c0.get("throttled_usec", 0)) / 1e3, (c1.get("usage_usec", 0) - c0.get("usage_usec", 0)) / 1e3))
print("%s K=%d total %d steps: host %.1f ms (issue %.1f ms), %.3f us/step; events %.1f ms" % (mode, K, total, dt * 1e3, t_issue * 1e3, dt / total * 1e6, ev0.elapsed_time(ev1)), env.session_stats())
dll = C.CDLL(os.path.join(os.path.dirname(_ffi.__file__), "libbbx.so"))
dll.bbx_session_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros(8 * 256 * 2, dtype=np.uint64)
dll.bbx_session_debug(env._h, out.ctypes.data_as(C.c_void_p), len(out))
d = out.reshape(8, 256, 2).astype(np.int64)
ns = min(255, total // 1024)
base = d[:, 1:ns + 1, 1][d[:, 1:ns + 1, 1] > 0].min()
for w in range(8):
    rt = (d[w, 1:ns + 1, 1] - base) / 1e5                   # ms
    st = np.diff(rt)
    med = float(np.median(st))
    gaps = [(i, rt[i], st[i]) for i in range(len(st)) if st[i] > 2.5 * med]
    print("env %4d: first sample %.1f ms, last %.1f ms, median %.2f ms per 1024 steps (%.2f us/step); gaps: %s" % (
        w * 512, rt[0], rt[-1], med, med * 1e3 / 1024, " ".join("@%.0fms+%.1fms" % (g[1], g[2]) for g in gaps[:12])))
