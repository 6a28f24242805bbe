"""Development tool: times fa_admm_ii / fa_admm_ice over flat device rows for a few shapes
(events around batches of calls) and prints algorithmic TB/s.

    [APPFL_AMD_LIB=...] python tools/admm_probe.py [n] [K,K,...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from appfl_amd import _native as N, ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_557_032
    Ks = [int(k) for k in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 16, 32]
    dev = torch.device("cuda")
    stride = (n + 63) // 64 * 64
    for K in Ks:
        g = torch.empty(n, device=dev)
        ops.synth_fill(g, 7, scale=0.05)
        slabs = torch.empty((3, K, stride), device=dev)
        for k in range(K):
            ops.synth_fill(slabs[0, k], 1000 + k, scale=0.01, base=g, n=n)
            ops.synth_fill(slabs[1, k], 3000 + k, scale=0.5, n=n)
            ops.synth_fill(slabs[2, k], 2000 + k, scale=0.01, base=g, n=n)
        rows = [[slabs[j, k] for k in range(K)] for j in range(3)]
        pen = [500.0 / (1 + k % 4) for k in range(K)]
        norms = torch.zeros(2, dtype=torch.float64, device=dev)
        ws = torch.empty(int(os.environ.get('ADMM_WS', N.FA_ADMM_RES_WS)), dtype=torch.float64, device=dev)
        carry = torch.empty(ops.admm_carry_floats(n), device=dev) if K > N.FA_ADMM_KCHUNK else None
        for alg in ("ii", "ice"):
            for prev in (True, False):
                def call():
                    ops.admm_round(alg, rows[0], rows[1], rows[2] if prev else None, pen, g, norms, ws,
                                   num_clients=K, carry_ws=carry, n=n)
                for _ in range(3):
                    call()
                reps = 20
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                streams = (3 if alg == "ii" else 2) + (1 if prev else 0)
                nbytes = 4 * (streams * K + 2) * n
                print(f"n={n} K={K:3d} {alg:3s} prev={int(prev)}  {ms:.4f} ms  {nbytes / ms / 1e9:.3f} TB/s "
                      f"({nbytes / ms / 1e9 / 8:.3f} of peak)", flush=True)
        del slabs, g


if __name__ == "__main__":
    main()
