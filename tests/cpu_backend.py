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


_CT = {np.float32: C.c_float, np.float64: C.c_double, np.int64: C.c_int64, np.int32: C.c_int32}


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
    a = _np(p, n)
    a[...] = a.dtype.type(value)


def _reduce_opt64(xs, w, g, m, v, opt, lr, beta1, beta2, tau):
    """float64 entries: the op sequence of oracle/aggregators.py in numpy doubles (scalars stay
    double, as PyTorch keeps them for float64 tensors)."""
    s = np.zeros_like(g)
    for x, wk in zip(xs, w):
        s = s + (x - g) * np.float64(wk)
    if opt == "avgm":
        v[...] = v * beta1 + s
        s = v.copy()
    elif opt != "none":
        m[...] = m * beta1 + s * (1 - beta1)
        s2 = s * s
        if opt == "adam":
            v[...] = v * beta2 + s2 * (1 - beta2)
        elif opt == "yogi":
            v[...] = v - (s2 * np.sign(v - s2)) * (1 - beta2)
        else:
            v[...] = v + s2
        s = (lr * m) / (np.sqrt(v) + tau)
    g[...] = g + s


def reduce_opt(x, w, g, m=None, v=None, opt="none", lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
               step_out=None, step_ws=None, n=None, persistent=True):
    n = g.numel() if n is None else n
    if g.dtype == torch.float64:
        return _reduce_opt64(_rows(x, n, np.float64), w, _np(g, n), _np(m, n), _np(v, n), opt, lr,
                             beta1, beta2, tau)
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
    out = _np(g, n)
    if out.dtype == np.float64:
        acc = np.zeros_like(out)
        for a in _rows(x, n, np.float64):
            acc = acc + a
        out[...] = acc / np.float64(len(x))
        return
    OC.mean(_rows(x, n, out.dtype.type), out)


_MIX_MODES = {"replace": 0, "convex": 1, "subtract": 2, "convex_one": 3, "sum_add": 4, "sum_set": 5}


def weighted_mix(x, w, g, mode, coeff=0.0, acc_ws=None, n=None):
    n = g.numel() if n is None else n
    OC.weighted_mix(_rows(x, n), w, _np(g, n), _MIX_MODES[mode], coeff)


class _HalfBackend:
    """The same interface for bfloat16 / float16 vectors, in PyTorch's own CPU arithmetic on
    16-bit tensors (the arithmetic the reference's server runs for such a model): the op
    sequences of algorithm/aggregator/fedavg_aggregator.py:276-368 and of the optimizer
    subclasses.  Calls on other tensors go to the functions above."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.half = {torch.bfloat16: 1, torch.float16: 2}[dtype]

    def for_half(self, dtype):
        return self if dtype == self.dtype else _HalfBackend(dtype)

    kchunk = staticmethod(kchunk)

    def _rows(self, x, n):
        out = []
        for t in x:
            if isinstance(t, int):
                t = torch.frombuffer((C.c_char * (2 * n)).from_address(t), dtype=self.dtype)
            out.append(t[:n])
        return out

    def fill(self, p, value, n=None):
        if p.dtype != self.dtype:
            return fill(p, value, n)
        (p if n is None else p[:n]).fill_(value)

    def _step(self, x, w, g, n):
        s = torch.zeros_like(g[:n])
        for xk, wk in zip(self._rows(x, n), w):
            s += wk * (xk - g[:n])
        return s

    def _opt(self, s, m, v, opt, lr, beta1, beta2, tau, n):
        if opt == "none":
            return s
        v = v[:n]
        if opt == "avgm":
            v.copy_(beta1 * v + s)
            return v.clone()
        m = m[:n]
        m.copy_(beta1 * m + (1 - beta1) * s)
        s2 = s * s
        if opt == "adam":
            v.copy_(beta2 * v + (1 - beta2) * s2)
        elif opt == "yogi":
            v.copy_(v - (s2 * torch.sign(v - s2)) * (1 - beta2))
        else:
            v.copy_(v + s2)
        return torch.div(lr * m, torch.sqrt(v) + tau)

    def reduce_opt(self, x, w, g, m=None, v=None, opt="none", lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
                   step_out=None, step_ws=None, n=None, persistent=True):
        if g.dtype != self.dtype:
            return reduce_opt(x, w, g, m, v, opt, lr, beta1, beta2, tau, step_out, step_ws, n)
        n = g.numel() if n is None else n
        s = self._opt(self._step(x, w, g, n), m, v, opt, lr, beta1, beta2, tau, n)
        if step_out is not None:
            step_out[:n].copy_(s)
        g[:n] += s

    def reduce_step(self, x, w, g, step_out, n=None):
        if g.dtype != self.dtype:
            return reduce_step(x, w, g, step_out, n)
        n = g.numel() if n is None else n
        step_out[:n].copy_(self._step(x, w, g, n))

    def server_opt_step(self, step, m, v, opt, lr=0.0, beta1=0.0, beta2=0.0, tau=0.0, n=None):
        if step.dtype != self.dtype:
            return server_opt_step(step, m, v, opt, lr, beta1, beta2, tau, n)
        n = step.numel() if n is None else n
        step[:n].copy_(self._opt(step[:n].clone(), m, v, opt, lr, beta1, beta2, tau, n))

    def apply_step(self, g, step, n=None):
        if g.dtype != self.dtype:
            return apply_step(g, step, n)
        n = g.numel() if n is None else n
        g[:n] += step[:n]

    def mean(self, x, g, n=None):
        if g.dtype != self.dtype:
            return mean(x, g, n)
        n = g.numel() if n is None else n
        s = torch.zeros_like(g[:n])
        for xk in self._rows(x, n):
            s += xk
        g[:n].copy_(torch.div(s, len(x)))

    def weighted_mix(self, x, w, g, mode, coeff=0.0, acc_ws=None, n=None):
        if g.dtype != self.dtype:
            return weighted_mix(x, w, g, mode, coeff, acc_ws, n)
        raise NotImplementedError("the 16-bit stand-in covers the FedAvg family only")


def for_half(dtype):
    return _HalfBackend(dtype)
