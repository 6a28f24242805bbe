"""Development tool: times fa_weighted_mix (FedCompass / FedQueue combine) over flat device rows.

    python tools/mix_probe.py [n] [K,K,...] [vptxunroll,...]   (0x8 = the default routing)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from appfl_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_557_032
    Ks = [int(k) for k in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 12, 16, 24, 32]
    stride = (n + 63) // 64 * 64
    for K in Ks:
        g = torch.empty(n, device="cuda")
        ops.synth_fill(g, 7, scale=0.05)
        slab = torch.empty((K, stride), device="cuda")
        for k in range(K):
            ops.synth_fill(slab[k], 1000 + k, scale=0.01, base=g, n=n)
        rows = [slab[k] for k in range(K)]
        w = [0.5 / K] * K
        shapes = [tuple(int(t) for t in sh.split("x")) for sh in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(0, 8)]
        for mode, streams in (("replace", K + 1), ("convex", K + 2)):
            line = f"n={n} K={K:3d} {mode:8s}"
            for vpt, unroll in shapes:
                ops.tune(vpt=vpt, unroll=unroll)

                def call():
                    ops.weighted_mix(rows, w, g, mode, coeff=0.5, n=n)
                for _ in range(3):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                line += f"  [{vpt}x{unroll}] {ms:.4f} ms {4 * streams * n / ms / 8e9:.3f}"
            ops.tune()
            print(line, flush=True)


if __name__ == "__main__":
    main()
