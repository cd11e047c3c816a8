"""INTEGRATION.md section 3 shows the whole ctypes binding a maintainer of the reference would write over the C ABI
(the equivalent of deepgroebner/wrapped.pyx:11-38).  This test EXECUTES that code block as printed — nothing of
deepgroebner_amd's Python is involved — and drives the resulting class against the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3."):text.index("## 4.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1, "section 3 holds exactly one python block"
    return blocks[0]


def test_stub_is_self_contained():
    src = stub_source()
    assert "deepgroebner_amd/libbbx.so" in src and "import deepgroebner_amd" not in src and "from deepgroebner_amd" not in src
    compile(src, "INTEGRATION.md#3", "exec")


@pytest.mark.gpu
def test_stub_runs_and_matches_the_oracle():
    from oracle import ffi
    bo = ffi.load("bo")
    src = stub_source().replace('"deepgroebner_amd/libbbx.so"', repr(os.path.join(ROOT, "deepgroebner_amd", "libbbx.so")))
    ns = {}
    exec(compile(src, "INTEGRATION.md#3", "exec"), ns)
    Env = ns["CLeadMonomialsEnv"]
    env = Env("3-20-10-weighted", k=2)
    env.seed(123)
    o = bo.env("3-20-10-weighted"); o.seed(123); o.reset()
    state = env.reset()
    assert state.dtype == np.int32 and state.shape == (19, 12) and np.array_equal(state, o.obs(2))
    twin = None
    for t in range(40):
        if t == 10:
            twin = env.copy()
            otwin = o.copy()
        a = (5 * t + 2) % len(state)
        state, r, done, info = env.step(a)
        assert r == o.step(a) and done == (o.nP == 0) and info == {} and np.array_equal(state, o.obs(2))
        if done:
            state = env.reset(); o.reset()
            assert np.array_equal(state, o.obs(2))
    s2 = twin.reset() if False else None      # (the copy continues independently from step 10)
    st = twin._state()
    assert np.array_equal(st, otwin.obs(2))
    st, r, done, _ = twin.step(0)
    assert r == otwin.step(0) and np.array_equal(st, otwin.obs(2))
