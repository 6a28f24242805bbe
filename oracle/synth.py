"""Counter-based synthetic data shared by the tests, the golden-vector generator and the
benchmark (TEST INFRASTRUCTURE).  Pure integer hashing plus exactly representable float steps,
so numpy here, the C oracle and the device generator (fa_synth_fill) produce identical bits.
"""

from __future__ import annotations

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)
NOISE_SCALE = np.float32(2.6428998e-05)  # 1 / std of the sum of four 16-bit uniforms


def noise(seed: int, n: int, first: int = 0) -> np.ndarray:
    """Approximately N(0,1) float32 values for counters first..first+n-1 of stream `seed`."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + n, dtype=np.uint64)
        z = np.uint64(seed) * _M1 + i
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        z = z ^ (z >> np.uint64(31))
    m = np.uint64(0xFFFF)
    s = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m)
         + ((z >> np.uint64(48)) & m)).astype(np.int64) - 131070
    return s.astype(np.float32) * NOISE_SCALE


def fill(n: int, seed: int, first: int = 0, scale: float = 1.0, base=None) -> np.ndarray:
    """base + scale * noise, each step one fp32 rounding (matches fa_synth_fill)."""
    t = np.float32(scale) * noise(seed, n, first)
    return t if base is None else (np.asarray(base, dtype=np.float32) + t).astype(np.float32)


def ints(seed: int, n: int, lo: int, hi: int, dtype=np.int64) -> np.ndarray:
    """Integers in [lo, hi) from the same hash (for integer state-dict entries)."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        z = np.uint64(seed) * _M1 + i
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        z = z ^ (z >> np.uint64(31))
    return (lo + (z % np.uint64(hi - lo)).astype(np.int64)).astype(dtype)
