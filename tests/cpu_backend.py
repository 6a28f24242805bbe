"""TEST INFRASTRUCTURE: the flat-kernel interface of appfl_amd.ops implemented on CPU tensors
with the C oracle, so the multi-rank logic of appfl_amd.sharded can be rehearsed with gloo on a
machine without a GPU.  Never used by the product package."""

import ctypes as C

import numpy as np
import torch

from oracle import c_oracle as OC

_CODE = {"none": OC.OPT_NONE, "avgm": OC.OPT_AVGM, "adam": OC.OPT_ADAM, "yogi": OC.OPT_YOGI,
         "adagrad": OC.OPT_ADAGRAD}


def _np(t, n=None):
    if t is None:
        return None
    a = t.numpy()
    return a if n is None else a[:n]


_CT = {np.float32: C.c_float, np.int64: C.c_int64, np.int32: C.c_int32}


def _rows(x, n, dtype=np.float32):
    """Client buffers as numpy arrays: CPU tensors, or raw host addresses (rows of a CPU slab)."""
    out = []
    for t in x:
        if isinstance(t, int):
            out.append(np.ctypeslib.as_array((_CT[dtype] * n).from_address(t)))
        else:
            out.append(np.ascontiguousarray(_np(t, n)))
    return out


def kchunk():
    return 128


def fill(p, value, n=None):
    _np(p, n)[...] = np.float32(value)


def reduce_opt(x, w, g, m=None, v=None, opt="none", lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
               step_out=None, step_ws=None, n=None):
    n = g.numel() if n is None else n
    OC.reduce_opt(_rows(x, n), w, _np(g, n), _np(m, n), _np(v, n),
                  _CODE[opt], lr=lr, beta1=beta1, beta2=beta2, tau=tau, step_out=_np(step_out, n))


def reduce_step(x, w, g, step_out, n=None):
    n = g.numel() if n is None else n
    OC.reduce_step(_rows(x, n), w, _np(g, n), _np(step_out, n))


def server_opt_step(step, m, v, opt, lr=0.0, beta1=0.0, beta2=0.0, tau=0.0, n=None):
    n = step.numel() if n is None else n
    OC.server_opt_step(_np(step, n), _np(m, n), _np(v, n), _CODE[opt], lr=lr, beta1=beta1,
                       beta2=beta2, tau=tau)


def apply_step(g, step, n=None):
    n = g.numel() if n is None else n
    OC.apply_step(_np(g, n), _np(step, n))


def mean(x, g, n=None):
    n = g.numel() if n is None else n
    OC.mean(_rows(x, n, _np(g, n).dtype.type), _np(g, n))
