"""Reproduction: after bbx_policy_step_device, rows[] in the caller's buffer against the library's own row counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgroebner_amd import VecLeadMonomialsEnv
from deepgroebner_amd.rollout import PMLPPolicy
B, R, k = 64, 512, 2
persistent = len(sys.argv) > 1 and sys.argv[1] == "session"
env = VecLeadMonomialsEnv("3-3-7-uniform", batch=B, k=k, caps={"lds_max_basis": 16})
env.seed(np.arange(B) + 51); env.seed_agent(np.arange(B) + 3); env.reset(); env.accounting(False)
if persistent:
    env.persistent(True)
torch.manual_seed(5003)
policy = PMLPPolicy(env.cols, [64]).cuda()
pw = policy._fused_weights()
s = torch.cuda.current_stream().cuda_stream
rew = torch.zeros(B, dtype=torch.float64, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
rows = torch.zeros(B, dtype=torch.int32, device="cuda"); obs = torch.full((B, R, env.cols), -1, dtype=torch.int32, device="cuda")
act = torch.zeros(B, dtype=torch.int32, device="cuda"); logp = torch.zeros(B, dtype=torch.float32, device="cuda")
U = torch.rand((400, B), device="cuda")
env.rollout_device("first", 0, False, s, rew, done, rows, obs, R, True, False); env.sync()
bad_total = 0
rows2 = torch.zeros(B, dtype=torch.int32, device="cuda"); obs2 = torch.full((B, R, env.cols), -1, dtype=torch.int32, device="cuda")
rew2 = torch.zeros(B, dtype=torch.float64, device="cuda"); done2 = torch.zeros(B, dtype=torch.uint8, device="cuda")
for t in range(400):
    env.policy_step_device(pw["prepared"], pw["hidden"], U[t], act, logp, rew, done, rows, obs, R, 2, s)
    env.sync()
    got = rows.cpu().numpy().copy(); dn = done.cpu().numpy().copy(); og = obs.clone()
    env.rollout_device("first", 0, False, s, rew2, done2, rows2, obs2, R, True, False); env.sync()    # the observation of the state as it is
    true = rows2.cpu().numpy()
    live = torch.arange(R, device="cuda")[None, :] < rows2[:, None]
    obs_bad = (~((og == obs2) | ~live[:, :, None]).all(dim=2).all(dim=1)).cpu().numpy()
    if not np.array_equal(got, true) or obs_bad.any():
        bad = np.flatnonzero((got != true) | obs_bad)
        bad_total += len(bad)
        if bad_total <= 12:
            print("step %d: envs %s: rows the call left %s, rows of the state %s, done %s, observation differs %s, basis %s" % (
                t, bad.tolist(), got[bad].tolist(), true[bad].tolist(), dn[bad].tolist(), obs_bad[bad].tolist(), env.stats()[bad, 7].tolist()))
print("mode %s: %d (step, env) where the block / row count the call left is not the state's, in 400 steps" % ("session" if persistent else "launch per step", bad_total))
