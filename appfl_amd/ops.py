"""Flat-vector operations on device tensors, one function per C-ABI entry point.

Every function enqueues its kernel on the CURRENT torch stream of the tensors' device and
returns immediately; tensors must be contiguous 1-D device tensors.  PyTorch is only used for
device memory and streams -- the arithmetic is in ``libappfl_amd.so``.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _native as N

OPT_CODES = {
    "none": N.FA_OPT_NONE,
    "avgm": N.FA_OPT_AVGM,
    "adam": N.FA_OPT_ADAM,
    "yogi": N.FA_OPT_YOGI,
    "adagrad": N.FA_OPT_ADAGRAD,
}


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dev(t: Optional[torch.Tensor], dtype=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise N.NativeLibraryError("libappfl_amd kernels take device tensors; got a CPU tensor")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _check_rows(x: Sequence, n: int, dtype) -> "C.Array":
    """Client buffers as device tensors, or as raw device addresses (ints) the caller vouches
    for (rows of a staging slab)."""
    ptrs = []
    for t in x:
        if isinstance(t, int):
            ptrs.append(t)
            continue
        if t.numel() < n:
            raise ValueError(f"client buffer has {t.numel()} elements, need {n}")
        ptrs.append(_dev(t, dtype))
    return N.ptr_array(ptrs)


def _float_entry(name: str, t: torch.Tensor):
    """(entry point, dtype) of the fp32 or fp64 flavour of a flat kernel, by `t`'s dtype."""
    if t.dtype == torch.float64:
        return getattr(N.load(), name + "_f64"), torch.float64
    return getattr(N.load(), name), torch.float32


def kchunk() -> int:
    return N.load().fa_kchunk()


def reduce_kernel_name(K: int, n: int, opt: str = "none") -> str:
    return N.load().fa_reduce_kernel_name(K, n, OPT_CODES[opt]).decode()


def tune(vpt: int = 0, unroll: int = 8, nt: int = 1, max_blocks: int = 0, stash: int = 2) -> None:
    """Benchmark knob (fa_tune): launch shape of the reduce kernels for the launches the CALLING
    THREAD makes from now on; launches of other threads (a scheduler's) are not affected.  Called
    without arguments it restores every default.  Product code never calls it: a caller that
    needs one round kept off the persistent kernels passes ``persistent=False`` to `reduce_opt`."""
    N.check(N.load().fa_tune(vpt, unroll, nt, max_blocks, stash), "fa_tune")


def scribble_meeting_counters(device=None) -> None:
    """Test hook (fa_debug_scribble_meeting_counters) on the current stream of `device`."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        N.check(N.load().fa_debug_scribble_meeting_counters(st), "fa_debug_scribble_meeting_counters")


def reduce_opt(x: Sequence[torch.Tensor], w: Sequence[float], g: torch.Tensor,
               m: Optional[torch.Tensor] = None, v: Optional[torch.Tensor] = None,
               opt: str = "none", lr: float = 0.0, beta1: float = 0.0, beta2: float = 0.0,
               tau: float = 0.0, step_out: Optional[torch.Tensor] = None,
               step_ws: Optional[torch.Tensor] = None, n: Optional[int] = None,
               persistent: bool = True) -> None:
    """Fused weighted step + server optimizer + apply over ``g[:n]`` (fa_reduce_opt).
    ``persistent=False`` (fp32 only; fa_reduce_opt_ex with FA_ROUTE_NO_PERSISTENT) keeps this call
    off the one-block-per-CU kernels, for rounds that run beside other kernels on the GPU."""
    n = g.numel() if n is None else n
    if len(x) != len(w):
        raise ValueError("one weight per client buffer")
    for t in (m, v, step_out, step_ws):
        if t is not None and t.numel() < n:
            raise ValueError("state / workspace tensor shorter than n")
    fn, dt = _float_entry("fa_reduce_opt", g)
    head = (_check_rows(x, n, dt), N.double_array([float(t) for t in w]), len(x),
            _dev(g, dt), _dev(m, dt), _dev(v, dt), n,
            OPT_CODES[opt], float(lr), float(beta1), float(beta2), float(tau),
            _dev(step_out, dt), _dev(step_ws, dt))
    if not persistent and dt == torch.float32:
        rc = N.load().fa_reduce_opt_ex(*head, N.FA_ROUTE_NO_PERSISTENT, _stream(g))
    else:
        rc = fn(*head, _stream(g))
    N.check(rc, "fa_reduce_opt")


_MEAN_DT = {torch.float32: N.FA_DT_F32, torch.int64: N.FA_DT_I64, torch.int32: N.FA_DT_I32}


def round_sync(x: Sequence, w: Sequence[float], g: torch.Tensor,
               m: Optional[torch.Tensor] = None, v: Optional[torch.Tensor] = None,
               opt: str = "none", lr: float = 0.0, beta1: float = 0.0, beta2: float = 0.0,
               tau: float = 0.0, step_ws: Optional[torch.Tensor] = None, n: Optional[int] = None,
               means: Sequence = (), k_ready: int = 0,
               rest_event: Optional["torch.cuda.Event"] = None, k_done: int = 0) -> None:
    """One synchronous round over K client rows in as few launches as possible (fa_round):
    `reduce_opt` over ``g[:n]`` plus the plain mean of every ``(row_offset_bytes, out, count)``
    in `means` -- segments found `row_offset_bytes` after ``x[k]`` in every client's row, whose
    means go to ``out[:count]`` (f32 / i64 / i32 by `out`'s dtype).  `k_ready` / `rest_event`:
    only the first `k_ready` clients' rows have landed; they are folded at once, the rest after
    `rest_event` (recorded behind the remaining transfers) -- needs `step_ws`.  `k_done`: the
    first `k_done` clients were folded into `step_ws` earlier (`reduce_step(..., k_done=)`); the
    round is finished from there (fa_round_resume)."""
    n = g.numel() if n is None else n
    if len(x) != len(w):
        raise ValueError("one weight per client buffer")
    for t in (m, v, step_ws):
        if t is not None and t.numel() < n:
            raise ValueError("state / workspace tensor shorter than n")
    segs = (N.MeanSegment * max(len(means), 1))()
    for i, (off, out, count) in enumerate(means):
        if out.numel() < count:
            raise ValueError("mean output shorter than its segment")
        segs[i] = N.MeanSegment(int(off), _dev(out), int(count), _MEAN_DT[out.dtype])
    if k_done:
        if k_ready:
            raise ValueError("k_ready and k_done exclude each other")
        rc = N.load().fa_round_resume(_check_rows(x, n, torch.float32), N.double_array([float(t) for t in w]),
                                      len(x), int(k_done), _dev(g, torch.float32), _dev(m, torch.float32),
                                      _dev(v, torch.float32), n, OPT_CODES[opt], float(lr), float(beta1),
                                      float(beta2), float(tau), _dev(step_ws, torch.float32),
                                      segs, len(means), _stream(g))
        N.check(rc, "fa_round_resume")
        return
    rc = N.load().fa_round(_check_rows(x, n, torch.float32), N.double_array([float(t) for t in w]),
                           len(x), _dev(g, torch.float32), _dev(m, torch.float32),
                           _dev(v, torch.float32), n, OPT_CODES[opt], float(lr), float(beta1),
                           float(beta2), float(tau), None, _dev(step_ws, torch.float32),
                           segs, len(means), int(k_ready),
                           C.c_void_p(rest_event.cuda_event) if rest_event is not None else None,
                           _stream(g))
    N.check(rc, "fa_round")


class RoundPlan:
    """The arguments of one `round_sync` call (fa_round without `k_ready`), kept for re-use: a
    server whose federation is steady launches the same round -- same rows of the same slab, same
    weights, same state vectors -- every time, and for a small model building the ctypes arrays of
    K pointers and K weights and converting 19 arguments again costs more than the kernel runs
    (MNIST-CNN size, 2-10 clients).  The arguments live in a library-side plan
    (fa_round_plan_create); `launch()` only fetches the current stream and hands over two words
    (fa_round_plan_launch).  The plan holds references to the tensors it points into; the caller
    drops it when any of them is replaced."""

    __slots__ = ("_launch", "_destroy", "_handle", "_keep", "_device", "key")

    def __init__(self, x: Sequence, w: Sequence[float], g: torch.Tensor,
                 m: Optional[torch.Tensor] = None, v: Optional[torch.Tensor] = None,
                 opt: str = "none", lr: float = 0.0, beta1: float = 0.0, beta2: float = 0.0,
                 tau: float = 0.0, step_ws: Optional[torch.Tensor] = None, n: Optional[int] = None,
                 means: Sequence = (), key=None):
        self._handle = None
        n = g.numel() if n is None else n
        if len(x) != len(w):
            raise ValueError("one weight per client buffer")
        for t in (m, v, step_ws):
            if t is not None and t.numel() < n:
                raise ValueError("state / workspace tensor shorter than n")
        segs = (N.MeanSegment * max(len(means), 1))()
        for i, (off, out, count) in enumerate(means):
            if out.numel() < count:
                raise ValueError("mean output shorter than its segment")
            segs[i] = N.MeanSegment(int(off), _dev(out), int(count), _MEAN_DT[out.dtype])
        f32 = torch.float32
        lib = N.load()
        handle = C.c_void_p()
        N.check(lib.fa_round_plan_create(
            _check_rows(x, n, f32), N.double_array([float(t) for t in w]), len(x), _dev(g, f32), _dev(m, f32),
            _dev(v, f32), n, OPT_CODES[opt], float(lr), float(beta1), float(beta2), float(tau),
            _dev(step_ws, f32), segs, len(means), C.byref(handle)), "fa_round_plan_create")
        self._handle = handle.value
        self._launch, self._destroy = lib.fa_round_plan_launch, lib.fa_round_plan_destroy
        self._keep = (x, g, m, v, step_ws, [out for _, out, _ in means])
        self._device = g.device
        self.key = key

    def launch(self) -> None:
        rc = self._launch(self._handle, torch.cuda.current_stream(self._device).cuda_stream)
        if rc:
            N.check(rc, "fa_round_plan_launch")

    def __del__(self):
        if self._handle is not None:
            self._destroy(self._handle)
            self._handle = None


def reduce_opt_slab(slab: torch.Tensor, stride: int, w: Sequence[float], g: torch.Tensor,
                    m=None, v=None, opt: str = "none", lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
                    step_out=None, step_ws=None, n: Optional[int] = None) -> None:
    """Same with the K client rows in one slab, row k at ``slab[k*stride:]``."""
    n = g.numel() if n is None else n
    K = len(w)
    if K < 1 or slab.numel() < (K - 1) * stride + n:
        raise ValueError("slab too small for K rows")
    rc = N.load().fa_reduce_opt_slab(
        _dev(slab, torch.float32), stride, N.double_array([float(t) for t in w]), K,
        _dev(g, torch.float32), _dev(m, torch.float32), _dev(v, torch.float32), n,
        OPT_CODES[opt], float(lr), float(beta1), float(beta2), float(tau),
        _dev(step_out, torch.float32), _dev(step_ws, torch.float32), _stream(g))
    N.check(rc, "fa_reduce_opt_slab")


def reduce_step(x, w, g, step_out, n: Optional[int] = None, k_done: int = 0) -> None:
    """Weighted step alone into `step_out` (fa_reduce_step).  `k_done` > 0: `step_out` holds the
    partial step of the first `k_done` clients already; the others are folded onto it
    (fa_reduce_step_resume, fp32 only)."""
    n = g.numel() if n is None else n
    if step_out.numel() < n:
        raise ValueError("step_out shorter than n")
    if k_done:
        if g.dtype != torch.float32:
            raise NotImplementedError("resumed steps are fp32 only")
        rc = N.load().fa_reduce_step_resume(_check_rows(x, n, torch.float32),
                                            N.double_array([float(t) for t in w]), len(x), int(k_done),
                                            _dev(g, torch.float32), _dev(step_out, torch.float32), n,
                                            _stream(g))
        N.check(rc, "fa_reduce_step_resume")
        return
    fn, dt = _float_entry("fa_reduce_step", g)
    rc = fn(_check_rows(x, n, dt), N.double_array([float(t) for t in w]), len(x),
            _dev(g, dt), _dev(step_out, dt), n, _stream(g))
    N.check(rc, "fa_reduce_step")


def server_opt_step(step, m, v, opt: str, lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
                    n: Optional[int] = None) -> None:
    """Optimizer epilogue alone on an already averaged step (fa_server_opt_step)."""
    n = step.numel() if n is None else n
    for t in (m, v):
        if t is not None and t.numel() < n:
            raise ValueError("moment tensor shorter than n")
    fn, dt = _float_entry("fa_server_opt_step", step)
    rc = fn(_dev(step, dt), _dev(m, dt), _dev(v, dt), n, OPT_CODES[opt], float(lr),
            float(beta1), float(beta2), float(tau), _stream(step))
    N.check(rc, "fa_server_opt_step")


def apply_step(g, step, n: Optional[int] = None) -> None:
    n = g.numel() if n is None else n
    if step.numel() < n:
        raise ValueError("step shorter than n")
    fn, dt = _float_entry("fa_apply_step", g)
    N.check(fn(_dev(g, dt), _dev(step, dt), n, _stream(g)), "fa_apply_step")


MIX_MODES = {"replace": 0, "convex": 1, "subtract": 2, "convex_one": 3, "sum_add": 4,
             "sum_set": 5}


def weighted_mix(x: Sequence, w: Sequence[float], g: torch.Tensor, mode: str, coeff: float = 0.0,
                 acc_ws: Optional[torch.Tensor] = None, n: Optional[int] = None) -> None:
    """FedCompass / FedQueue mixes over ``g[:n]`` (fa_weighted_mix)."""
    n = g.numel() if n is None else n
    if len(x) != len(w):
        raise ValueError("one weight per client buffer")
    fn, dt = _float_entry("fa_weighted_mix", g)
    rc = fn(_check_rows(x, n, dt), N.double_array([float(t) for t in w]), len(x), _dev(g, dt),
            float(coeff), MIX_MODES[mode], _dev(acc_ws, dt), n, _stream(g))
    N.check(rc, "fa_weighted_mix")


def admm_carry_floats(n: int) -> int:
    """Floats of the carry workspace an ADMM round of more than FA_ADMM_KCHUNK clients needs."""
    return 2 * ((n + 3) // 4 * 4)


def admm_round(alg: str, primal: Sequence, dual: Sequence, prev: Optional[Sequence],
               penalty: Sequence[float], g: torch.Tensor, norms: torch.Tensor, res_ws: torch.Tensor,
               num_clients: int = 1, carry_ws: Optional[torch.Tensor] = None,
               n: Optional[int] = None) -> None:
    """IIADMM ("ii": fa_admm_ii, the `dual` rows are updated in place) or ICEADMM ("ice":
    fa_admm_ice) server update over ``g[:n]``; `norms` (2 float64) receives the primal and dual
    residual norms, `res_ws` is the FA_ADMM_RES_WS-double workspace of the norm sums."""
    n = g.numel() if n is None else n
    K = len(primal)
    if len(dual) != K or len(penalty) != K or (prev is not None and len(prev) != K):
        raise ValueError("one dual row, one penalty (and one previous primal row) per client")
    if norms.numel() < 2 or res_ws.numel() < N.FA_ADMM_RES_WS:
        raise ValueError("norms needs 2 doubles, res_ws FA_ADMM_RES_WS doubles")
    if carry_ws is not None and carry_ws.numel() < admm_carry_floats(n):
        raise ValueError("carry workspace shorter than 2 * round_up(n, 4) floats")
    f32, f64 = torch.float32, torch.float64
    common = (_check_rows(primal, n, f32), _check_rows(dual, n, f32),
              _check_rows(prev, n, f32) if prev is not None else None,
              N.double_array([float(t) for t in penalty]), K)
    tail = (_dev(g, f32), n, _dev(carry_ws, f32), _dev(res_ws, f64), _dev(norms, f64), _stream(g))
    if alg == "ii":
        rc = N.load().fa_admm_ii(*common, int(num_clients), *tail)
    elif alg == "ice":
        rc = N.load().fa_admm_ice(*common, *tail)
    else:
        raise ValueError(f"unknown ADMM flavour {alg!r}")
    N.check(rc, "fa_admm_" + alg)


_MEAN = {torch.float32: "fa_mean_f32", torch.float64: "fa_mean_f64", torch.int64: "fa_mean_i64",
         torch.int32: "fa_mean_i32"}
_RAW = {torch.float32: "fa_buff_accum_raw_f32", torch.float64: "fa_buff_accum_raw_f64",
        torch.int64: "fa_buff_accum_raw_i64", torch.int32: "fa_buff_accum_raw_i32"}
_FLUSH = {torch.float32: "fa_buff_flush_mean_f32", torch.float64: "fa_buff_flush_mean_f64",
          torch.int64: "fa_buff_flush_mean_i64", torch.int32: "fa_buff_flush_mean_i32"}


def mean(x: Sequence[torch.Tensor], g: torch.Tensor, n: Optional[int] = None) -> None:
    """Plain mean over clients into ``g[:n]`` (f32 / i64 / i32)."""
    n = g.numel() if n is None else n
    fn = getattr(N.load(), _MEAN[g.dtype])
    N.check(fn(_check_rows(x, n, g.dtype), len(x), _dev(g), n, _stream(g)), _MEAN[g.dtype])


def async_mix(x, alpha_t: float, g, gradient_based: bool = False, sub_form: bool = True,
              step_out=None, n: Optional[int] = None) -> None:
    n = g.numel() if n is None else n
    if x.numel() < n:
        raise ValueError("x shorter than n")
    fn, dt = _float_entry("fa_async_mix", g)
    rc = fn(_dev(x, dt), float(alpha_t), int(bool(gradient_based)), int(bool(sub_form)), _dev(g, dt),
            _dev(step_out, dt), n, _stream(g))
    N.check(rc, "fa_async_mix")


def buff_accum(x, alpha_t: float, g, step, first: bool, gradient_based: bool = False,
               flush: bool = False, n: Optional[int] = None) -> None:
    n = g.numel() if n is None else n
    if x.numel() < n or step.numel() < n:
        raise ValueError("x / step shorter than n")
    fn, dt = _float_entry("fa_buff_accum_flush" if flush else "fa_buff_accum", g)
    rc = fn(_dev(x, dt), float(alpha_t), int(bool(gradient_based)),
            _dev(g, dt), _dev(step, dt), int(bool(first)), n, _stream(g))
    N.check(rc, "fa_buff_accum")


def buff_accum_raw(x, step, first: bool, n: Optional[int] = None) -> None:
    n = step.numel() if n is None else n
    if x.numel() < n:
        raise ValueError("x shorter than n")
    fn = getattr(N.load(), _RAW[step.dtype])
    N.check(fn(_dev(x, step.dtype), _dev(step), int(bool(first)), n, _stream(step)),
            _RAW[step.dtype])


def buff_flush_mean(g, step, K: int, n: Optional[int] = None) -> None:
    n = g.numel() if n is None else n
    if step.numel() < n:
        raise ValueError("step shorter than n")
    fn = getattr(N.load(), _FLUSH[g.dtype])
    N.check(fn(_dev(g), _dev(step, g.dtype), int(K), n, _stream(g)), _FLUSH[g.dtype])


def fill(p: torch.Tensor, value: float, n: Optional[int] = None) -> None:
    n = p.numel() if n is None else n
    if n == 0:
        return
    if p.dtype == torch.float64:
        N.check(N.load().fa_fill_f64(_dev(p, torch.float64), C.c_double(value), n, _stream(p)),
                "fa_fill_f64")
        return
    N.check(N.load().fa_fill_f32(_dev(p, torch.float32), C.c_float(value), n, _stream(p)),
            "fa_fill_f32")


def synth_fill(p: torch.Tensor, seed: int, first: int = 0, scale: float = 1.0,
               base: Optional[torch.Tensor] = None, n: Optional[int] = None) -> None:
    """Fill ``p[:n]`` with the shared counter-based synthetic update (see fa_synth_fill)."""
    n = p.numel() if n is None else n
    if base is not None and base.numel() < n:
        raise ValueError("base shorter than n")
    rc = N.load().fa_synth_fill(_dev(p, torch.float32), _dev(base, torch.float32), n,
                                C.c_uint64(seed), C.c_uint64(first), C.c_float(scale), _stream(p))
    N.check(rc, "fa_synth_fill")
