"""Development tool: time the reduce kernel over launch shapes and workloads on one GPU.

usage: python tools/sweep.py --cfg "n,K,opt;n,K,opt" --shapes "vpt,unroll,nt,maxblk;..." [--reps R]
Interleaves the shapes round-robin inside one process (A/B comparisons are only meaningful
within one process on one device).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from appfl_amd import ops


def time_blocks(fns, reps, batch=1):
    """Like time_rounds, but every callable runs its `reps` samples in a row (after two warm-up
    calls of its own): what a server that repeats ONE kernel sees, without whatever the previous
    shape's traversal order left in the Infinity Cache."""
    out = []
    for f in fns:
        f()
        f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(batch):
                f()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) / batch)
        out.append(sorted(ts))
    return out


def time_rounds(fns, reps, batch=1):
    """fns: list of callables; returns per-fn sorted times (ms per call), interleaved round-robin;
    `batch` calls are enqueued back to back between the two events of one sample."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    out = [[] for _ in fns]
    for _ in range(reps):
        for i, f in enumerate(fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(batch):
                f()
            b.record()
            b.synchronize()
            out[i].append(a.elapsed_time(b) / batch)
    return [sorted(t) for t in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="11689512,100,none")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pad", type=int, default=0, help="extra floats between client rows")
    ap.add_argument("--shapes", default="4,8,1,65536")
    ap.add_argument("--batch", type=int, default=1, help="launches per timed sample (small kernels)")
    ap.add_argument("--blocks", action="store_true", help="time every shape in a row instead of interleaved")
    a = ap.parse_args()
    shapes = [tuple(int(t) for t in s.split(",")) for s in a.shapes.split(";")]
    hp = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
    for cfg in a.cfg.split(";"):
        n, K, opt = cfg.split(",")
        n, K = int(n), int(K)
        stride = (n + 63) // 64 * 64 + a.pad
        slab = torch.empty(K * stride, device="cuda")
        g = torch.empty(n, device="cuda")
        ops.synth_fill(g, 7, scale=0.05)
        for k in range(K):
            ops.synth_fill(slab[k * stride:], 1000 + k, scale=0.01, base=g, n=n)
        xs = [slab[k * stride:k * stride + n] for k in range(K)]
        w = [1.0 / K] * K
        m = torch.zeros(n, device="cuda")
        v = torch.full((n,), 1e-6, device="cuda")
        extra = {"none": 2, "avgm": 4}.get(opt, 6)
        alg = 4 * (K + extra) * n

        def mk(shape):
            def f():
                ops.tune(*shape)
                if opt == "none":
                    ops.reduce_opt(xs, w, g)
                else:
                    ops.reduce_opt(xs, w, g, m, v, opt, **hp)
            return f
        ts = (time_blocks if a.blocks else time_rounds)([mk(s) for s in shapes], a.reps, a.batch)
        print(f"== n={n} K={K} opt={opt} client bytes {K*n*4/1e9:.2f} GB", flush=True)
        for s, t in zip(shapes, ts):
            med, best = t[len(t) // 2], t[0]
            print(f"  shape={s}: med {med:.4f} ms best {best:.4f}  client {K*n*4/med/1e6:7.0f} GB/s  "
                  f"alg {alg/med/1e6:7.0f} GB/s  frac {alg/med/1e6/8000:.3f}", flush=True)
        del slab, xs, g, m, v
        torch.cuda.empty_cache()
    ops.tune()


if __name__ == "__main__":
    main()
