"""Experiment: effective shader clock over time under the headline workload, (a) back-to-back K-step launches,
(b) one persistent session.  Library built with -DBBX_DRIFT_DEBUG.   python scripts/exp_clock.py MODE TOTAL K"""
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
dll = C.CDLL(os.path.join(os.path.dirname(_ffi.__file__), "libbbx.so"))
n = 1200
probe = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
dll.bbx_debug_clock_probe(C.c_void_p(probe.data_ptr()), n, 100000)        # every 1 ms for 1.2 s
time.sleep(0.05)                                                          # 50 ms of idle first
if mode == "persistent":
    env.persistent(True)
t0 = time.perf_counter()
for _ in range(total // K):
    env.rollout_device("random", K, True, s.cuda_stream, rew, done, rows, obs, R, False, True)
env.sync(); dt = time.perf_counter() - t0
print("%s K=%d total %d steps: %.1f ms, %.3f us/step" % (mode, K, total, dt * 1e3, dt / total * 1e6), env.session_stats() if mode == "persistent" else "")
torch.cuda.synchronize()
p = probe.cpu().numpy().reshape(n, 2)
mhz = np.diff(p[:, 0]) / np.diff(p[:, 1]) * 100.0
t_ms = (p[1:, 1] - p[0, 1]) / 1e5
print("shader clock (MHz) by time (ms): " + " ".join("%d:%.0f" % (t_ms[i], mhz[i]) for i in range(0, min(len(mhz), int(dt * 1e3) + 120), 20)))
