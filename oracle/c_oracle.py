"""ctypes view of ``oracle/_build/libfedagg_oracle.so`` (see ``fedagg_oracle.c``).

TEST INFRASTRUCTURE ONLY -- imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the ``appfl_amd`` package.  All arrays are numpy
arrays in host memory; functions mutate ``g`` / ``m`` / ``v`` / ``step`` in place like the
device calls they check.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libfedagg_oracle.so")

OPT_NONE, OPT_AVGM, OPT_ADAM, OPT_YOGI, OPT_ADAGRAD = range(5)

_lib: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "fedagg_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ptrs(arrs: Sequence[np.ndarray]):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _chk(a: np.ndarray, dtype) -> np.ndarray:
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a


def reduce_opt(x, w, g, m=None, v=None, opt=OPT_NONE, lr=0.0, beta1=0.0, beta2=0.0, tau=0.0,
               step_out=None) -> None:
    n = g.size
    for a in x:
        _chk(a, np.float32)
        assert a.size == n
    wa = (C.c_double * len(w))(*[float(t) for t in w])
    rc = load().or_reduce_opt(_ptrs(x), wa, C.c_int(len(x)), _ptr(_chk(g, np.float32)), _ptr(m),
                              _ptr(v), C.c_size_t(n), C.c_int(opt), C.c_double(lr),
                              C.c_double(beta1), C.c_double(beta2), C.c_double(tau),
                              _ptr(step_out))
    assert rc == 0, rc


def reduce_step(x, w, g, step_out) -> None:
    wa = (C.c_double * len(w))(*[float(t) for t in w])
    rc = load().or_reduce_step(_ptrs(x), wa, C.c_int(len(x)), _ptr(_chk(g, np.float32)),
                               _ptr(_chk(step_out, np.float32)), C.c_size_t(g.size))
    assert rc == 0, rc


def server_opt_step(step, m, v, opt, lr=0.0, beta1=0.0, beta2=0.0, tau=0.0) -> None:
    rc = load().or_server_opt_step(_ptr(_chk(step, np.float32)), _ptr(m), _ptr(v),
                                   C.c_size_t(step.size), C.c_int(opt), C.c_double(lr),
                                   C.c_double(beta1), C.c_double(beta2), C.c_double(tau))
    assert rc == 0, rc


def weighted_mix(x, w, g, mode, coeff=0.0) -> None:
    wa = (C.c_double * len(w))(*[float(t) for t in w])
    rc = load().or_weighted_mix(_ptrs(x), wa, C.c_int(len(x)), _ptr(_chk(g, np.float32)),
                                C.c_double(coeff), C.c_int(mode), C.c_size_t(g.size))
    assert rc == 0, rc


def apply_step(g, step) -> None:
    load().or_apply_step(_ptr(_chk(g, np.float32)), _ptr(_chk(step, np.float32)),
                         C.c_size_t(g.size))


def mean(x, g) -> None:
    fn = {np.dtype(np.float32): "or_mean_f32", np.dtype(np.int64): "or_mean_i64",
          np.dtype(np.int32): "or_mean_i32"}[g.dtype]
    for a in x:
        _chk(a, g.dtype)
    getattr(load(), fn)(_ptrs(x), C.c_int(len(x)), _ptr(g), C.c_size_t(g.size))


def async_mix(x, alpha_t, gradient_based, sub_form, g, step_out=None) -> None:
    load().or_async_mix(_ptr(_chk(x, np.float32)), C.c_double(alpha_t), C.c_int(gradient_based),
                        C.c_int(sub_form), _ptr(_chk(g, np.float32)), _ptr(step_out),
                        C.c_size_t(g.size))


def buff_accum(x, alpha_t, gradient_based, g, step, first, flush) -> None:
    load().or_buff_accum(_ptr(_chk(x, np.float32)), C.c_double(alpha_t), C.c_int(gradient_based),
                         _ptr(_chk(g, np.float32)), _ptr(_chk(step, np.float32)), C.c_int(first),
                         C.c_int(flush), C.c_size_t(g.size))


def buff_accum_raw(x, step, first) -> None:
    fn = {np.dtype(np.float32): "or_buff_accum_raw_f32", np.dtype(np.int64): "or_buff_accum_raw_i64",
          np.dtype(np.int32): "or_buff_accum_raw_i32"}[step.dtype]
    getattr(load(), fn)(_ptr(_chk(x, step.dtype)), _ptr(step), C.c_int(first),
                        C.c_size_t(step.size))


def buff_flush_mean(g, step, K) -> None:
    fn = {np.dtype(np.float32): "or_buff_flush_mean_f32", np.dtype(np.int64): "or_buff_flush_mean_i64",
          np.dtype(np.int32): "or_buff_flush_mean_i32"}[g.dtype]
    getattr(load(), fn)(_ptr(g), _ptr(_chk(step, g.dtype)), C.c_int(K), C.c_size_t(g.size))


def admm(alg: str, primal, dual, prev, penalty, num_clients: int, g):
    """IIADMM ("ii": `dual` rows are updated in place) / ICEADMM ("ice") server update on flat
    vectors; returns (primal residual, dual residual)."""
    n = g.size
    for a in list(primal) + list(dual) + list(prev or []):
        _chk(a, np.float32)
        assert a.size == n
    res = (C.c_double * 2)()
    pa = (C.c_double * len(penalty))(*[float(t) for t in penalty])
    rc = load().or_admm(C.c_int({"ii": 0, "ice": 1}[alg]), _ptrs(primal), _ptrs(dual),
                        _ptrs(prev) if prev else None, pa, C.c_int(len(primal)), C.c_int(num_clients),
                        _ptr(_chk(g, np.float32)), C.c_size_t(n), res)
    assert rc == 0, rc
    return float(res[0]), float(res[1])


def synth_fill(n: int, seed: int, first: int = 0, scale: float = 1.0,
               base: Optional[np.ndarray] = None) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    load().or_synth_fill(_ptr(out), _ptr(base), C.c_size_t(n), C.c_uint64(seed),
                         C.c_uint64(first), C.c_float(scale))
    return out
