import sys, numpy as np
sys.path.insert(0, '.')
from deepgroebner_amd import VecLeadMonomialsEnv
from oracle import ffi
bo = ffi.load('bo')
dist = sys.argv[1] if len(sys.argv) > 1 else '3-8-6-maximum-pure-homog'
for caps in ({"lds_max_basis": -1}, None):
    print("caps", caps)
    env = VecLeadMonomialsEnv(dist, batch=1, k=2, caps=caps)
    env.seed([5]); env.seed_agent([5]); env.trace_enable(8)
    env.reset()
    o = bo.env(dist); o.seed(5); o.reset()
    basis, pairs, order = env.state(0)
    ob = o.basis()
    print("nG", len(basis), o.nG, "nP", len(pairs), o.nP)
    for g in range(len(basis)):
        ok = np.array_equal(basis[g][0], ob[g][0]) and np.array_equal(basis[g][1], ob[g][1])
        if not ok: print("poly", g, basis[g], ob[g])
    print("pairs eq", np.array_equal(pairs, o.pairs()), "order eq", np.array_equal(order, o.reducer_order()), order, o.reducer_order())
    try:
        env.rollout("random", 3, auto_reset=True)
    except Exception as ex:
        print("rollout failed", ex)
    print(env.trace_read(0, 0, 3))
    for t in range(3):
        a = ffi.agent_action(5, t, o.nP)
        r = o.step(a); print("oracle", a, r, o.nP, o.nG)
