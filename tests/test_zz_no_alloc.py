"""fa_reduce_opt_static -- the entry point that allocates nothing -- against fa_reduce_opt with a
caller-provided workspace: same inputs, bit-identical g / m / v, on two calls in a row (the second
with a smaller input, so stale contents of the library's array lie beyond what it may read)."""

import ctypes as C

import numpy as np
import pytest
import torch

from appfl_amd import _native as N
from appfl_amd import ops

pytestmark = pytest.mark.gpu

HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
OPT = {"none": 0, "avgm": 1, "adam": 2}


def _call(static, rows, w, g, m, v, n, opt, step_ws=None):
    lib = N.load()
    K = len(rows)
    x = N.ptr_array([r.data_ptr() for r in rows])
    wd = N.double_array(w)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mp = None if m is None else m.data_ptr()
    vp = None if v is None else v.data_ptr()
    if static:
        return lib.fa_reduce_opt_static(x, wd, K, g.data_ptr(), mp, vp, n, OPT[opt], HP["lr"],
                                        HP["beta1"], HP["beta2"], HP["tau"], None, st)
    return lib.fa_reduce_opt(x, wd, K, g.data_ptr(), mp, vp, n, OPT[opt], HP["lr"], HP["beta1"],
                             HP["beta2"], HP["tau"], None, step_ws.data_ptr(), st)


def _same(a, b):
    return np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("opt", ["none", "adam"])
def test_static_workspace_matches_caller_workspace(opt):
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    K = 2 * ops.kchunk() + 5                     # three client chunks: the carry is read twice
    w = [(1.0 + (k % 7)) / (4.0 * K) for k in range(K)]
    for n in ((1 << 20) + 7, 300001):            # second call smaller than the first
        base = torch.empty(n, dtype=torch.float32, device=dev)
        ops.synth_fill(base, 7, scale=0.05)
        rows = []
        for k in range(K):
            r = torch.empty(n, dtype=torch.float32, device=dev)
            ops.synth_fill(r, 1000 + k, scale=0.01, base=base)
            rows.append(r)
        state = {}
        for static in (False, True):
            g = base.clone()
            m = torch.zeros_like(g) if opt == "adam" else None
            v = torch.full_like(g, HP["tau"] ** 2) if opt == "adam" else None
            ws = torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
            for _ in range(2):                   # two rounds, so m / v feed back
                rc = _call(static, rows, w, g, m, v, n, opt, ws)
                assert rc == 0, N.load().fa_error_string(rc)
            torch.cuda.synchronize()
            state[static] = (g, m, v)
        for a, b, name in zip(state[False], state[True], "gmv"):
            if a is not None:
                assert _same(a, b), f"n={n} {opt}: {name} differs between the two entry points"
        assert not _same(state[True][0], base)   # the round did change g


def test_static_workspace_rejects_oversized_rounds_and_allocates_nothing():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    K = ops.kchunk() + 1
    small = torch.zeros(1024, dtype=torch.float32, device=dev)
    rows = [small] * K
    w = [1.0 / K] * K
    too_many = (32 << 20) + 4
    rc = _call(True, rows, w, small, None, None, too_many, "none")
    assert rc == -4 and b"FA_STATIC_WS_FLOATS" in N.load().fa_error_string(rc)
    # a round of at most FA_KCHUNK clients needs no workspace: any n goes through fa_reduce_opt's path
    n = 4096
    g = torch.zeros(n, dtype=torch.float32, device=dev)
    xs = [torch.full((n,), float(k), dtype=torch.float32, device=dev) for k in range(4)]
    ops.fill(g, 0.0)          # loads the library's code object (and with it the array) on this GPU
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
    assert _call(True, xs, [0.25] * 4, g, None, None, n, "none") == 0
    rows2 = [xs[k % 4] for k in range(K)]
    assert _call(True, rows2, w, g, None, None, n, "none") == 0
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
    assert after[0] == before[0]
    assert after[1] >= before[1] - (1 << 20), "device memory shrank during calls that must not allocate"
