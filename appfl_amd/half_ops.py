"""The flat-vector operations for bfloat16 / float16 state dicts.

A 16-bit model stays 16 bits wide on the device: client rows, the global state, the moments and
the steps are bfloat16 / float16 tensors, half the bytes of an fp32 model everywhere.
:class:`HalfOps` has the function names and arguments of :mod:`appfl_amd.ops` and sends every
call on tensors of its 16-bit type to the ``fa_half_*`` entry points, which widen on load, round
each operation's result to the 16-bit type exactly where PyTorch's CPU kernels round it in the
reference (the float path of misc/memory_utils.py:112-131 on a 16-bit tensor) and narrow on
store; calls on integer or float64 tensors pass through to :mod:`appfl_amd.ops`.  `round_sync`
is here a sequence of launches (weighted entries, then each plain-mean segment) with the same
`k_ready` / `k_done` contract as the fp32 one, so the aggregators' stage-on-arrival folding and
their pipelined last arrival serve 16-bit models too; there is no `RoundPlan` (the aggregators
test for it before keeping a round's arguments).
"""

from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _native as N
from . import ops
from .ops import OPT_CODES, _check_rows, _dev, _stream

HALF_CODES = {torch.bfloat16: N.FA_HALF_BF16, torch.float16: N.FA_HALF_F16}


class HalfOps:
    def __init__(self, dtype: torch.dtype):
        self.dtype = dtype
        self.half = HALF_CODES[dtype]

    kchunk = staticmethod(ops.kchunk)

    def reduce_kernel_name(self, K: int, n: int, opt: str = "none") -> str:
        """Name of the kernel `reduce_opt` launches for K clients of n aligned 16-bit elements."""
        return N.load().fa_half_reduce_kernel_name(K, n, OPT_CODES[opt], self.half).decode()

    def round_scalar(self, x: float) -> float:
        """`x` as the 16-bit type holds it (fa_half_round_scalar)."""
        return float(N.load().fa_half_round_scalar(float(x), self.half))

    def reduce_opt(self, x: Sequence, w: Sequence[float], g: torch.Tensor, m=None, v=None,
                   opt: str = "none", lr: float = 0.0, beta1: float = 0.0, beta2: float = 0.0,
                   tau: float = 0.0, step_out=None, step_ws=None, n: Optional[int] = None,
                   persistent: bool = True) -> None:
        """`persistent=False` (fa_half_reduce_opt_ex with FA_ROUTE_NO_PERSISTENT) keeps this call on
        the tile-per-block kernel: for a round that runs beside other kernels on the same GPU."""
        if g.dtype != self.dtype:
            return ops.reduce_opt(x, w, g, m, v, opt, lr, beta1, beta2, tau, step_out, step_ws, n, persistent)
        dt = self.dtype
        n = g.numel() if n is None else n
        if len(x) != len(w):
            raise ValueError("one weight per client buffer")
        for t in (m, v, step_out, step_ws):
            if t is not None and t.numel() < n:
                raise ValueError("state / workspace tensor shorter than n")
        rc = N.load().fa_half_reduce_opt_ex(
            _check_rows(x, n, dt), N.double_array([float(t) for t in w]), len(x), _dev(g, dt),
            _dev(m, dt), _dev(v, dt), n, OPT_CODES[opt], float(lr), float(beta1), float(beta2),
            float(tau), _dev(step_out, dt), _dev(step_ws, dt), self.half,
            N.FA_ROUTE_DEFAULT if persistent else N.FA_ROUTE_NO_PERSISTENT, _stream(g))
        N.check(rc, "fa_half_reduce_opt_ex")

    def reduce_step(self, x, w, g, step_out, n: Optional[int] = None, k_done: int = 0) -> None:
        """Weighted step alone into `step_out`; `k_done` > 0: `step_out` holds the partial step of
        the first `k_done` clients already (fa_half_reduce_step_resume)."""
        if g.dtype != self.dtype:
            return ops.reduce_step(x, w, g, step_out, n, k_done)
        dt = self.dtype
        n = g.numel() if n is None else n
        if step_out.numel() < n:
            raise ValueError("step_out shorter than n")
        rc = N.load().fa_half_reduce_step_resume(
            _check_rows(x, n, dt), N.double_array([float(t) for t in w]), len(x), int(k_done), _dev(g, dt),
            _dev(step_out, dt), n, self.half, _stream(g))
        N.check(rc, "fa_half_reduce_step_resume")

    def round_sync(self, x: Sequence, w: Sequence[float], g: torch.Tensor, m=None, v=None, opt: str = "none",
                   lr: float = 0.0, beta1: float = 0.0, beta2: float = 0.0, tau: float = 0.0,
                   step_ws: Optional[torch.Tensor] = None, n: Optional[int] = None, means: Sequence = (),
                   k_ready: int = 0, rest_event: Optional["torch.cuda.Event"] = None, k_done: int = 0) -> None:
        """One synchronous round over K client rows, arguments as `ops.round_sync`: the weighted
        entries ``g[:n]`` (16-bit) and the plain mean of every ``(row_offset_bytes, out, count)``
        in `means`, as separate launches.  `k_ready` / `rest_event`: the first `k_ready` clients
        are folded at once, the others after `rest_event`; `k_done`: the first `k_done` clients are
        in `step_ws` already."""
        dt = self.dtype
        n = g.numel() if n is None else n
        K = len(x)
        if K != len(w):
            raise ValueError("one weight per client buffer")
        if k_ready and k_done:
            raise ValueError("k_ready and k_done exclude each other")
        for t in (m, v, step_ws):
            if t is not None and t.numel() < n:
                raise ValueError("state / workspace tensor shorter than n")
        stream = torch.cuda.current_stream(g.device)
        split = 0 < k_ready < K
        if n:
            if split:
                if step_ws is None:
                    raise ValueError("folding the rows that have landed needs step_ws")
                self.reduce_step(x[:k_ready], w[:k_ready], g, step_ws, n=n)
                k_done = k_ready
            if split and rest_event is not None:
                stream.wait_event(rest_event)
            rc = N.load().fa_half_reduce_opt_resume(
                _check_rows(x, n, dt), N.double_array([float(t) for t in w]), K, int(k_done), _dev(g, dt),
                _dev(m, dt), _dev(v, dt), n, OPT_CODES[opt], float(lr), float(beta1), float(beta2), float(tau),
                _dev(step_ws, dt), self.half, _stream(g))
            N.check(rc, "fa_half_reduce_opt_resume")
        elif split and rest_event is not None:
            stream.wait_event(rest_event)
        ptrs = [t if isinstance(t, int) else t.data_ptr() for t in x]
        for off, out, count in means:
            if count:
                self.mean([p + int(off) for p in ptrs], out, n=int(count))

    def server_opt_step(self, step, m, v, opt: str, lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
                        n: Optional[int] = None) -> None:
        if step.dtype != self.dtype:
            return ops.server_opt_step(step, m, v, opt, lr, beta1, beta2, tau, n)
        dt = self.dtype
        n = step.numel() if n is None else n
        for t in (m, v):
            if t is not None and t.numel() < n:
                raise ValueError("moment tensor shorter than n")
        rc = N.load().fa_half_server_opt_step(
            _dev(step, dt), _dev(m, dt), _dev(v, dt), n, OPT_CODES[opt], float(lr), float(beta1),
            float(beta2), float(tau), self.half, _stream(step))
        N.check(rc, "fa_half_server_opt_step")

    def apply_step(self, g, step, n: Optional[int] = None) -> None:
        if g.dtype != self.dtype:
            return ops.apply_step(g, step, n)
        n = g.numel() if n is None else n
        if step.numel() < n:
            raise ValueError("step shorter than n")
        N.check(N.load().fa_half_apply_step(_dev(g, self.dtype), _dev(step, self.dtype), n, self.half,
                                            _stream(g)), "fa_half_apply_step")

    def mean(self, x: Sequence, g: torch.Tensor, n: Optional[int] = None) -> None:
        if g.dtype != self.dtype:
            return ops.mean(x, g, n)
        n = g.numel() if n is None else n
        N.check(N.load().fa_half_mean(_check_rows(x, n, self.dtype), len(x), _dev(g, self.dtype), n,
                                      self.half, _stream(g)), "fa_half_mean")

    def weighted_mix(self, x: Sequence, w: Sequence[float], g: torch.Tensor, mode: str, coeff: float = 0.0,
                     acc_ws: Optional[torch.Tensor] = None, n: Optional[int] = None) -> None:
        if g.dtype != self.dtype:
            return ops.weighted_mix(x, w, g, mode, coeff, acc_ws, n)
        n = g.numel() if n is None else n
        if len(x) != len(w):
            raise ValueError("one weight per client buffer")
        rc = N.load().fa_half_weighted_mix(
            _check_rows(x, n, self.dtype), N.double_array([float(t) for t in w]), len(x), _dev(g, self.dtype),
            float(coeff), ops.MIX_MODES[mode], _dev(acc_ws, self.dtype), n, self.half, _stream(g))
        N.check(rc, "fa_half_weighted_mix")

    def async_mix(self, x, alpha_t: float, g, gradient_based: bool = False, sub_form: bool = True,
                  step_out=None, n: Optional[int] = None) -> None:
        if g.dtype != self.dtype:
            return ops.async_mix(x, alpha_t, g, gradient_based, sub_form, step_out, n)
        dt = self.dtype
        n = g.numel() if n is None else n
        if x.numel() < n:
            raise ValueError("x shorter than n")
        rc = N.load().fa_half_async_mix(_dev(x, dt), float(alpha_t), int(bool(gradient_based)),
                                        int(bool(sub_form)), _dev(g, dt), _dev(step_out, dt), n,
                                        self.half, _stream(g))
        N.check(rc, "fa_half_async_mix")

    def buff_accum(self, x, alpha_t: float, g, step, first: bool, gradient_based: bool = False,
                   flush: bool = False, n: Optional[int] = None) -> None:
        if g.dtype != self.dtype:
            return ops.buff_accum(x, alpha_t, g, step, first, gradient_based, flush, n)
        dt = self.dtype
        n = g.numel() if n is None else n
        if x.numel() < n or step.numel() < n:
            raise ValueError("x / step shorter than n")
        rc = N.load().fa_half_buff_accum(_dev(x, dt), float(alpha_t), int(bool(gradient_based)),
                                         _dev(g, dt), _dev(step, dt), int(bool(first)),
                                         int(bool(flush)), n, self.half, _stream(g))
        N.check(rc, "fa_half_buff_accum")

    def buff_accum_raw(self, x, step, first: bool, n: Optional[int] = None) -> None:
        if step.dtype != self.dtype:
            return ops.buff_accum_raw(x, step, first, n)
        n = step.numel() if n is None else n
        if x.numel() < n:
            raise ValueError("x shorter than n")
        N.check(N.load().fa_half_buff_accum_raw(_dev(x, self.dtype), _dev(step, self.dtype),
                                                int(bool(first)), n, self.half, _stream(step)),
                "fa_half_buff_accum_raw")

    def buff_flush_mean(self, g, step, K: int, n: Optional[int] = None) -> None:
        if g.dtype != self.dtype:
            return ops.buff_flush_mean(g, step, K, n)
        n = g.numel() if n is None else n
        if step.numel() < n:
            raise ValueError("step shorter than n")
        N.check(N.load().fa_half_buff_flush_mean(_dev(g, self.dtype), _dev(step, self.dtype), int(K), n,
                                                 self.half, _stream(g)), "fa_half_buff_flush_mean")

    def fill(self, p: torch.Tensor, value: float, n: Optional[int] = None) -> None:
        """p[:n] = value as the 16-bit type holds it (zeros_like(...) + tau**2 on a 16-bit tensor,
        fedadam_aggregator.py:45-49)."""
        if p.dtype != self.dtype:
            return ops.fill(p, value, n)
        n = p.numel() if n is None else n
        if n:
            N.check(N.load().fa_half_fill(_dev(p, self.dtype), float(value), n, self.half, _stream(p)),
                    "fa_half_fill")


def for_dtype(dtype: torch.dtype, base=ops):
    """The operations for vectors of the 16-bit type `dtype`.  `base` is :mod:`appfl_amd.ops` in
    the product; the CPU stand-in with which the tests rehearse the multi-rank logic under gloo
    brings its own 16-bit counterpart (`base.for_half(dtype)`)."""
    if base is ops:
        return HalfOps(dtype)
    if hasattr(base, "for_half"):
        return base.for_half(dtype)
    raise NotImplementedError("bfloat16 / float16 state dicts need the HIP library's own operations")


def for_layout(layout, base):
    """The operations an aggregator uses for `layout`: `base` (normally :mod:`appfl_amd.ops`)
    when its float entries are fp32 / fp64, the 16-bit ones when they are bfloat16 / float16."""
    if layout is None or layout.half_dtype is None:
        return base
    return for_dtype(layout.half_dtype, base)
