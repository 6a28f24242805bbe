"""TEST INFRASTRUCTURE: the flat-kernel interface of appfl_amd.ops implemented on CPU tensors
with the C oracle, so the multi-rank logic of appfl_amd.sharded can be rehearsed with gloo on a
machine without a GPU.  Never used by the product package."""

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


def fill(p, value, n=None):
    _np(p, n)[...] = np.float32(value)


def reduce_opt(x, w, g, m=None, v=None, opt="none", lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
               step_out=None, step_ws=None, n=None):
    n = g.numel() if n is None else n
    OC.reduce_opt([np.ascontiguousarray(_np(t, n)) for t in x], w, _np(g, n), _np(m, n), _np(v, n),
                  _CODE[opt], lr=lr, beta1=beta1, beta2=beta2, tau=tau, step_out=_np(step_out, n))


def reduce_step(x, w, g, step_out, n=None):
    n = g.numel() if n is None else n
    OC.reduce_step([np.ascontiguousarray(_np(t, n)) for t in x], w, _np(g, n), _np(step_out, n))


def server_opt_step(step, m, v, opt, lr=0.0, beta1=0.0, beta2=0.0, tau=0.0, n=None):
    n = step.numel() if n is None else n
    OC.server_opt_step(_np(step, n), _np(m, n), _np(v, n), _CODE[opt], lr=lr, beta1=beta1,
                       beta2=beta2, tau=tau)


def apply_step(g, step, n=None):
    n = g.numel() if n is None else n
    OC.apply_step(_np(g, n), _np(step, n))


def mean(x, g, n=None):
    n = g.numel() if n is None else n
    OC.mean([np.ascontiguousarray(_np(t, n)) for t in x], _np(g, n))
