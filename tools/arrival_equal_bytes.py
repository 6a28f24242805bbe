"""Development tool: the FedAsync arrival kernel in fp32 and in bfloat16 at EQUAL bytes (the
bfloat16 vector twice as long), to tell the kernels apart from the cache the shorter vector fits."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from appfl_amd import half_ops, ops  # noqa: E402


def time_ms(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    h = half_ops.HalfOps(torch.bfloat16)
    for mbytes in (173, 346, 692, 1384):
        n32 = mbytes * 1000000 // 4
        n16 = 2 * n32
        g32 = torch.randn(n32, device="cuda") * 0.05
        x32 = g32 + torch.randn(n32, device="cuda") * 0.01
        t32 = time_ms(lambda: ops.async_mix(x32, 0.05, g32))
        del g32, x32
        g16 = (torch.randn(n16, device="cuda") * 0.05).to(torch.bfloat16)
        x16 = (g16.float() + torch.randn(n16, device="cuda") * 0.01).to(torch.bfloat16)
        t16 = time_ms(lambda: h.async_mix(x16, 0.05, g16))
        del g16, x16
        b = 3 * mbytes * 1e6
        print(f"{mbytes:5d} MB per vector: fp32 async_mix {t32:.4f} ms ({b / t32 / 1e6 / 8000:.3f} of peak)   "
              f"bf16 arrival {t16:.4f} ms ({b / t16 / 1e6 / 8000:.3f} of peak)", flush=True)


if __name__ == "__main__":
    main()
