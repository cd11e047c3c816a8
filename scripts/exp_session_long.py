"""Persistent sessions under launches of different lengths: the same 8192 steps issued as 8 x 1024, 1 x 8192, 82 x 100, 410 x 20
(events on the launch stream around the launches and the join behind them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgroebner_amd import VecLeadMonomialsEnv
B = 4096
torch.cuda.init()
env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset(); env.accounting(False)
R = 256
d_obs = torch.empty((B, R, env.cols), dtype=torch.int32, device="cuda")
d_rew = torch.empty(B, dtype=torch.float64, device="cuda"); d_done = torch.empty(B, dtype=torch.uint8, device="cuda"); d_rows = torch.empty(B, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream()
def launch(n): env.rollout_device("random", n, True, stream.cuda_stream, d_rew, d_done, d_rows, d_obs, R, False, True)
for pers in (True, False):
    env.persistent(pers)
    launch(256); env.sync()
    for k, r in ((1024, 8), (8192, 1), (100, 82), (20, 410), (1024, 8)):
        launch(64); env.join(stream.cuda_stream)
        s0 = env.session_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(r): launch(k)
        env.join(stream.cuda_stream); e1.record(stream)
        env.sync(); torch.cuda.synchronize()
        s1 = env.session_stats()
        ms = e0.elapsed_time(e1)
        print("persistent=%d  %4d x %5d steps: %7.2f ms  %7.1f M env-steps/s   session stats delta %s" % (pers, r, k, ms, r * k * B / ms / 1e3, {a: s1[a] - s0[a] for a in s1}))
