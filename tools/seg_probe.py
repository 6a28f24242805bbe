"""Development tool: one plain-FedAvg launch over a very long vector against the same vector cut
into 2 / 3 / 4 / 8 stretches launched back to back (kernel time by events, ms and fraction of the
8 TB/s peak).  python tools/seg_probe.py [elements] [clients] [slab]"""
import sys

import torch

sys.path.insert(0, ".")
from appfl_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 536_870_912
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    if len(sys.argv) > 3 and sys.argv[3] == "slab":      # one allocation, rows back to back
        slab = torch.empty((K, n), device=dev)
        rows = [slab[k] for k in range(K)]
    else:
        rows = [torch.empty(n, device=dev) for _ in range(K)]
    for k, r in enumerate(rows):
        ops.synth_fill(r, seed=100 + k, scale=0.01)
    g = torch.zeros(n, device=dev)
    w = [1.0 / K] * K
    for cuts in (1, 2, 3, 4, 5, 8, 1):
        per = (n // cuts + 255) // 256 * 256
        spans = [(o, min(per, n - o)) for o in range(0, n, per)]

        def run():
            for o, l in spans:
                ops.reduce_opt([r[o:o + l] for r in rows], w, g[o:o + l], None, None, "none")
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        best, tot, reps = 1e9, 0.0, 6
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best, tot = min(best, t), tot + t
        gb = n * 4.0 * (K + 2) / 1e9
        print(f"n={n} K={K} cuts={cuts} kernel={ops.reduce_kernel_name(K, spans[0][1], 'none')} "
              f"avg={tot / reps:.3f} ms min={best:.3f} ms frac={gb / (tot / reps) / 8000 * 1000:.3f}", flush=True)


if __name__ == "__main__":
    main()
