"""Long-horizon soak of the headline batch: many launches of the device rollout (ideals drawn in-kernel, spills to the
HBM-resident pass and back at the next launch), counters of a few environments against the compiled reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi
B, CH, N = 4096, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.cuda.init()
env = VecLeadMonomialsEnv("3-20-10-weighted", batch=B, k=2)
env.seed(np.arange(B) + 1000); env.seed_agent(np.arange(B)); env.reset()
env.accounting(False)
if len(sys.argv) > 2 and sys.argv[2] == "persistent":
    env.persistent(True)                      # the launches feed one session (bbx_persistent); every 10th is synchronised
d_obs = torch.empty((B, 256, env.cols), dtype=torch.int32, device="cuda")
d_rew = torch.empty(B, dtype=torch.float64, device="cuda"); d_done = torch.empty(B, dtype=torch.uint8, device="cuda")
d_rows = torch.empty(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream()
t0 = time.perf_counter()
for i in range(N):
    env.rollout_device("random", CH, True, s.cuda_stream, d_rew, d_done, d_rows, d_obs, 256, False, True)
    if i % 10 == 9 or not (len(sys.argv) > 2 and sys.argv[2] == "persistent"):
        env.sync()
env.sync()
t1 = time.perf_counter()
st = env.stats()
assert (st[:, 0] == CH * N).all() and (st[:, 4] == 0).all()
lib = ffi.load("ref" if ffi.available("ref") else "bo")
res = lib.bench_random("3-20-10-weighted", 2, 8, CH * N, 1000, 0)
print("soak: %d envs x %d steps in %.2f s (%.0f M env-steps/s sustained, no host generation); episodes %d; max basis %d; "
      "additions of envs 0..7 match the %s: %s" % (B, CH * N, t1 - t0, B * CH * N / (t1 - t0) / 1e6, int(st[:, 2].sum()), int(st[:, 7].max()),
                                                    lib.kind, res["additions"] == int(st[:8, 1].sum())))
assert res["additions"] == int(st[:8, 1].sum())
