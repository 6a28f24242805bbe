"""Development sweep: the 16-bit tile-per-block reduce kernel against the persistent kernel with
the late epilogue (APPFL_AMD_HALF_LATE = 0 / 4, read by the library at every launch) over vector
lengths, client counts and optimizers; kernel time from events, both timed in turn in one process.
usage: python tools/half_late_sweep.py [bf16|f16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from appfl_amd import half_ops  # noqa: E402

HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
STATE = {"none": 1, "avgm": 2, "adam": 3, "yogi": 3, "adagrad": 3}


def timed(fn, reps=8):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts[:-2]) / (len(ts) - 2)


def main():
    dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.bfloat16
    o = half_ops.HalfOps(dt)
    for n in (4_500_000, 8_000_000, 11_689_512, 17_000_000, 25_557_032, 47_000_000, 86_567_656, 162_500_000):
        for K in (8, 12, 16, 24, 32, 48, 64, 100):
            if n * K > 6_000_000_000:
                continue
            rows = torch.empty((K, n), dtype=dt, device="cuda").normal_(0, 0.05)
            g = torch.empty(n, dtype=dt, device="cuda").normal_(0, 0.05)
            m, v = torch.zeros_like(g), torch.full_like(g, 1e-3)
            xs, w = [rows[k] for k in range(K)], [1.0 / K] * K
            for opt in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("adam", "avgm")):
                def run():
                    o.reduce_opt(xs, w, g, None if opt in ("avgm", "none") else m, None if opt == "none" else v, opt, **HP)
                out = []
                for mode in ("0", "4", "0", "4"):
                    os.environ["APPFL_AMD_HALF_LATE"] = mode
                    run()
                    out.append(timed(run))
                tile, late = min(out[0], out[2]), min(out[1], out[3])
                byt = 2 * (K + 2 * STATE[opt]) * n
                print(f"n={n:>11,} K={K:>3} {opt:5s}: tile {tile:.4f} ms {byt / tile / 8e9:.3f}   late {late:.4f} ms "
                      f"{byt / late / 8e9:.3f}   {100 * (late / tile - 1):+.1f} %", flush=True)
            del rows, xs


if __name__ == "__main__":
    main()
