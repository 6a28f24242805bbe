"""ctypes binding of ``libappfl_amd.so`` (the C ABI declared in ``include/appfl_amd.h``).

The library is the only compute path of this package: if it cannot be loaded, importing
:data:`lib` raises and nothing falls back to the CPU.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
# APPFL_AMD_LIB selects another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("APPFL_AMD_LIB") or os.path.join(_HERE, "lib", "libappfl_amd.so")

FA_OK = 0
FA_OPT_NONE, FA_OPT_AVGM, FA_OPT_ADAM, FA_OPT_YOGI, FA_OPT_ADAGRAD = range(5)

_p = C.c_void_p
_sz = C.c_size_t
_i = C.c_int
_d = C.c_double
_f = C.c_float
_u64 = C.c_uint64

FA_DT_F32, FA_DT_I64, FA_DT_I32 = range(3)
FA_ROUTE_DEFAULT, FA_ROUTE_NO_PERSISTENT = 0, 1
FA_HALF_BF16, FA_HALF_F16 = 1, 2
FA_ADMM_KCHUNK, FA_ADMM_BLOCKS = 64, 2048
FA_ADMM_RES_WS = 2 + 2 * FA_ADMM_BLOCKS


class MeanSegment(C.Structure):
    """``fa_mean_segment``: one plain-mean segment of the client rows (fa_round)."""
    _fields_ = [("row_offset", C.c_size_t), ("out", C.c_void_p), ("n", C.c_size_t),
                ("dtype", C.c_int)]

# name -> (restype, argtypes); mirrors include/appfl_amd.h one to one.
SIGNATURES = {
    "fa_version": (_i, []),
    "fa_error_string": (C.c_char_p, [_i]),
    "fa_kchunk": (_i, []),
    "fa_reduce_kernel_name": (C.c_char_p, [_i, _sz, _i]),
    "fa_tune": (_i, [_i, _i, _i, _i, _i]),
    "fa_reduce_opt": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _p]),
    "fa_reduce_opt_ex": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, C.c_uint, _p]),
    "fa_debug_scribble_meeting_counters": (_i, [_p]),
    "fa_round": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _p, _i, _i, _p, _p]),
    "fa_round_plan_create": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _i, _p]),
    "fa_round_plan_launch": (_i, [_p, _p]),
    "fa_round_plan_destroy": (None, [_p]),
    "fa_reduce_apply": (_i, [_p, _p, _i, _p, _sz, _p, _p]),
    "fa_reduce_opt_static": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p]),
    "fa_reduce_opt_slab": (_i, [_p, _sz, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _p]),
    "fa_reduce_step": (_i, [_p, _p, _i, _p, _p, _sz, _p]),
    "fa_reduce_step_resume": (_i, [_p, _p, _i, _i, _p, _p, _sz, _p]),
    "fa_round_resume": (_i, [_p, _p, _i, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _i, _p]),
    "fa_server_opt_step": (_i, [_p, _p, _p, _sz, _i, _d, _d, _d, _d, _p]),
    "fa_apply_step": (_i, [_p, _p, _sz, _p]),
    "fa_weighted_mix": (_i, [_p, _p, _i, _p, _d, _i, _p, _sz, _p]),
    "fa_admm_ii": (_i, [_p, _p, _p, _p, _i, _i, _p, _sz, _p, _p, _p, _p]),
    "fa_admm_ice": (_i, [_p, _p, _p, _p, _i, _p, _sz, _p, _p, _p, _p]),
    "fa_mean_f32": (_i, [_p, _i, _p, _sz, _p]),
    "fa_mean_i64": (_i, [_p, _i, _p, _sz, _p]),
    "fa_mean_i32": (_i, [_p, _i, _p, _sz, _p]),
    "fa_async_mix": (_i, [_p, _d, _i, _i, _p, _p, _sz, _p]),
    "fa_buff_accum": (_i, [_p, _d, _i, _p, _p, _i, _sz, _p]),
    "fa_buff_accum_flush": (_i, [_p, _d, _i, _p, _p, _i, _sz, _p]),
    "fa_buff_accum_raw_f32": (_i, [_p, _p, _i, _sz, _p]),
    "fa_buff_accum_raw_i64": (_i, [_p, _p, _i, _sz, _p]),
    "fa_buff_accum_raw_i32": (_i, [_p, _p, _i, _sz, _p]),
    "fa_buff_flush_mean_f32": (_i, [_p, _p, _i, _sz, _p]),
    "fa_buff_flush_mean_i64": (_i, [_p, _p, _i, _sz, _p]),
    "fa_buff_flush_mean_i32": (_i, [_p, _p, _i, _sz, _p]),
    "fa_fill_f32": (_i, [_p, _f, _sz, _p]),
    "fa_reduce_opt_f64": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _p]),
    "fa_reduce_step_f64": (_i, [_p, _p, _i, _p, _p, _sz, _p]),
    "fa_server_opt_step_f64": (_i, [_p, _p, _p, _sz, _i, _d, _d, _d, _d, _p]),
    "fa_apply_step_f64": (_i, [_p, _p, _sz, _p]),
    "fa_mean_f64": (_i, [_p, _i, _p, _sz, _p]),
    "fa_weighted_mix_f64": (_i, [_p, _p, _i, _p, _d, _i, _p, _sz, _p]),
    "fa_async_mix_f64": (_i, [_p, _d, _i, _i, _p, _p, _sz, _p]),
    "fa_buff_accum_f64": (_i, [_p, _d, _i, _p, _p, _i, _sz, _p]),
    "fa_buff_accum_flush_f64": (_i, [_p, _d, _i, _p, _p, _i, _sz, _p]),
    "fa_buff_accum_raw_f64": (_i, [_p, _p, _i, _sz, _p]),
    "fa_buff_flush_mean_f64": (_i, [_p, _p, _i, _sz, _p]),
    "fa_fill_f64": (_i, [_p, _d, _sz, _p]),
    "fa_half_round_scalar": (_f, [_d, _i]),
    "fa_half_reduce_kernel_name": (C.c_char_p, [_i, _sz, _i, _i]),
    "fa_half_reduce_opt_ex": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _i, C.c_uint, _p]),
    "fa_half_reduce_opt": (_i, [_p, _p, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _p, _i, _p]),
    "fa_half_reduce_step": (_i, [_p, _p, _i, _p, _p, _sz, _i, _p]),
    "fa_half_reduce_step_resume": (_i, [_p, _p, _i, _i, _p, _p, _sz, _i, _p]),
    "fa_half_reduce_opt_resume": (_i, [_p, _p, _i, _i, _p, _p, _p, _sz, _i, _d, _d, _d, _d, _p, _i, _p]),
    "fa_half_server_opt_step": (_i, [_p, _p, _p, _sz, _i, _d, _d, _d, _d, _i, _p]),
    "fa_half_apply_step": (_i, [_p, _p, _sz, _i, _p]),
    "fa_half_mean": (_i, [_p, _i, _p, _sz, _i, _p]),
    "fa_half_weighted_mix": (_i, [_p, _p, _i, _p, _d, _i, _p, _sz, _i, _p]),
    "fa_half_async_mix": (_i, [_p, _d, _i, _i, _p, _p, _sz, _i, _p]),
    "fa_half_buff_accum": (_i, [_p, _d, _i, _p, _p, _i, _i, _sz, _i, _p]),
    "fa_half_buff_accum_raw": (_i, [_p, _p, _i, _sz, _i, _p]),
    "fa_half_buff_flush_mean": (_i, [_p, _p, _i, _sz, _i, _p]),
    "fa_half_fill": (_i, [_p, _d, _sz, _i, _p]),
    "fa_synth_fill": (_i, [_p, _p, _sz, _u64, _u64, _f, _p]),
    "fa_pack_host": (_i, [_p, _p, _p, _p, _i, _i]),
    "fa_pack_host_window": (_i, [_p, _p, _p, _p, _i, _sz, _sz, _i]),
    "fa_host_copy_fresh": (_i, [_p, _p, _sz, _i]),
    "fa_host_copy_crc32": (_i, [_p, _p, _sz, _sz, _p, _i]),
}


class NativeLibraryError(RuntimeError):
    """The HIP library is missing or a call into it failed."""


_lock = threading.Lock()
_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load ``libappfl_amd.so`` once and attach the prototypes.  Raises if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(code: int, what: str = "") -> None:
    if code != FA_OK:
        msg = load().fa_error_string(code).decode()
        raise NativeLibraryError(f"{what or 'libappfl_amd call'} failed: {msg} (code {code})")


def ptr_array(ptrs: Sequence[int]):
    """Host array of device pointers for the ``x`` argument of the reduce / mean calls."""
    return (C.c_void_p * len(ptrs))(*ptrs)


def double_array(vals: Sequence[float]):
    return (C.c_double * len(vals))(*vals)
