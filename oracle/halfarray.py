"""bfloat16 / float16 tensors for the numpy oracle.

TEST INFRASTRUCTURE ONLY (see oracle/aggregators.py).

PyTorch's CPU kernels for the two 16-bit float types compute every element-wise operation in
fp32 and round the result to the 16-bit type (round to nearest even); nothing is fused across
operations.  A Python scalar that MULTIPLIES or DIVIDES such a tensor enters as the fp32 value of
the scalar; one that is ADDED or SUBTRACTED is first rounded to the tensor's own type.
:class:`HalfArray` restates exactly that on top of numpy: the values are carried in an fp32
array that only ever holds values representable in the 16-bit type, every operator rounds its
result, and numpy's ufuncs / `zeros_like` dispatch here, so the formulas of
oracle/aggregators.py run unchanged on it.  Pinned, like the rest of the oracle, by golden
vectors produced by the reference's classes on bfloat16 / float16 state dicts.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


def round_bf16(a: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bfloat16 (ties to even), returned as fp32."""
    shape = np.shape(a)
    a = np.ascontiguousarray(a, F32).reshape(-1)          # (ascontiguousarray makes 0-d arrays 1-d)
    u = a.view(np.uint32)
    r = (u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) & np.uint32(0xFFFF0000)
    out = r.view(F32).copy()
    nan = np.isnan(a)
    if nan.any():
        out[nan] = np.nan
    return out.reshape(shape)


def round_f16(a: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return np.asarray(a, F32).astype(np.float16).astype(F32)


_ROUND = {"bf16": round_bf16, "f16": round_f16}


def round_scalar(x, kind: str) -> np.float32:
    """A Python scalar as the 16-bit type holds it (double -> fp32 -> 16-bit), as fp32."""
    return _ROUND[kind](np.array([F32(x)]))[0]


ITEMSIZE = 2


class _HalfDType:
    """Stands where numpy code asks an array for its dtype: `dtype.type(x)` makes the scalar the
    arithmetic will see (an fp32 value; HalfArray's add / subtract round it further)."""

    def __init__(self, kind: str):
        self.kind_name = kind
        self.itemsize = ITEMSIZE

    def type(self, x):
        return F32(x)

    def __eq__(self, other):
        return isinstance(other, _HalfDType) and other.kind_name == self.kind_name

    def __hash__(self):
        return hash(("half", self.kind_name))

    def __repr__(self):
        return self.kind_name


class HalfArray:
    __array_priority__ = 1000

    def __init__(self, values: np.ndarray, kind: str, _rounded: bool = False):
        self.kind = kind
        self.v = np.asarray(values, F32) if _rounded else _ROUND[kind](np.asarray(values, F32))

    # -- what the oracle asks of an array -------------------------------------------------------
    @property
    def dtype(self):
        return _HalfDType(self.kind)

    @property
    def shape(self):
        return self.v.shape

    @property
    def size(self):
        return self.v.size

    @property
    def itemsize(self):
        return ITEMSIZE

    def copy(self):
        return HalfArray(self.v.copy(), self.kind, True)

    def astype(self, dtype, copy=True):
        if isinstance(dtype, _HalfDType) and dtype.kind_name == self.kind:
            return self.copy() if copy else self
        raise TypeError(f"cannot cast a {self.kind} oracle array to {dtype}")

    def bits(self) -> np.ndarray:
        """The 16-bit patterns (uint16), which is how records hold these tensors."""
        v = np.where(np.isnan(self.v), F32(np.nan), self.v).astype(F32)      # one quiet NaN for all
        if self.kind == "bf16":
            flat = np.ascontiguousarray(v).reshape(-1)
            return (flat.view(np.uint32) >> np.uint32(16)).astype(np.uint16).reshape(v.shape)
        return v.astype(np.float16).reshape(-1).view(np.uint16).reshape(v.shape)

    def to_torch(self):
        import torch
        return torch.from_numpy(self.v.copy()).to(torch.bfloat16 if self.kind == "bf16" else torch.float16)

    # -- arithmetic ---------------------------------------------------------------------------------
    def _wrap(self, a):
        return HalfArray(a, self.kind)

    def _operand(self, o, additive: bool):
        if isinstance(o, HalfArray):
            if o.kind != self.kind:
                raise TypeError("bf16 and f16 oracle arrays do not mix")
            return o.v
        s = F32(o)                                     # Python / numpy scalar: fp32 first
        return round_scalar(s, self.kind) if additive else s

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs:
            return NotImplemented
        mine = next(x for x in inputs if isinstance(x, HalfArray))
        additive = ufunc in (np.add, np.subtract)
        if ufunc in (np.add, np.subtract, np.multiply, np.true_divide, np.divide):
            a, b = (mine._operand(x, additive) for x in inputs)
            with np.errstate(all="ignore"):
                return mine._wrap(ufunc(a, b, dtype=F32))
        if ufunc in (np.negative, np.sqrt, np.sign, np.square):
            with np.errstate(all="ignore"):
                return mine._wrap(ufunc(mine.v))
        return NotImplemented

    def __array_function__(self, func, types, args, kwargs):
        if func is np.zeros_like:
            return HalfArray(np.zeros_like(args[0].v), args[0].kind, True)
        return NotImplemented

    def __add__(self, o):
        return np.add(self, o)

    def __radd__(self, o):
        return np.add(o, self)

    def __sub__(self, o):
        return np.subtract(self, o)

    def __rsub__(self, o):
        return np.subtract(o, self)

    def __mul__(self, o):
        return np.multiply(self, o)

    def __rmul__(self, o):
        return np.multiply(o, self)

    def __truediv__(self, o):
        return np.true_divide(self, o)

    def __neg__(self):
        return np.negative(self)


def zeros_axpy(alpha, b: HalfArray, sign: int) -> HalfArray:
    """zeros_like(b).add_(b, alpha=alpha) (sign +1) / .sub_(b, alpha=alpha) (sign -1): ONE tensor
    operation -- 0 +/- alpha * b in fp32 (alpha cast to the tensor's type first), rounded once.
    Differs from rounding the product first in the sign of a zero result."""
    a = round_scalar(alpha, b.kind)
    with np.errstate(all="ignore"):
        p = (a * b.v).astype(F32)
        return HalfArray(F32(0.0) + p if sign > 0 else F32(0.0) - p, b.kind)


def from_torch(t):
    """A torch bfloat16 / float16 tensor as a HalfArray (exact)."""
    import torch
    kind = {torch.bfloat16: "bf16", torch.float16: "f16"}[t.dtype]
    return HalfArray(t.detach().to(torch.float32).cpu().numpy(), kind, True)
