"""Development tool: time the reduce kernel over launch shapes on one GPU (not part of the product).

usage: python tools/sweep.py [--n N] [--k K] [--opt none|adam] [--reps R]
"""
import argparse
import itertools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from appfl_amd import ops


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=11689512)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--opt", default="none")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pad", type=int, default=0, help="extra floats between client rows")
    ap.add_argument("--shapes", default="")
    a = ap.parse_args()
    n, K = a.n, a.k
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
    hp = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
    extra = {"none": 2, "avgm": 4}.get(a.opt, 6)
    alg = 4 * (K + extra) * n
    # copy baseline: what this box's HBM gives a plain float4 copy
    src = torch.empty(256 * 1024 * 1024, device="cuda"); dst = torch.empty_like(src)
    med, best = timeit(lambda: dst.copy_(src), 10)
    print(f"copy 1GiB->1GiB: med {med:.3f} ms  {2*src.numel()*4/med/1e6:.0f} GB/s (r+w)")
    del src, dst
    if a.shapes:
        shapes = [tuple(int(t) for t in s.split(",")) for s in a.shapes.split(";")]
    else:
        shapes = [(vpt, u, nt, mb) for vpt, u, nt, mb in itertools.product((1, 2, 4), (4, 8, 16), (1, 0), (2048, 1024, 4096, 0x7fffffff))]
    res = []
    for vpt, u, nt, mb in shapes:
        ops.tune(vpt, u, nt, mb)
        if a.opt == "none":
            fn = lambda: ops.reduce_opt(xs, w, g)
        else:
            fn = lambda: ops.reduce_opt(xs, w, g, m, v, a.opt, **hp)
        med, best = timeit(fn, a.reps)
        res.append((med, vpt, u, nt, mb, best))
        print(f"vpt={vpt} unroll={u:2d} nt={nt} maxblk={mb:10d}: med {med:.4f} ms best {best:.4f} ms  "
              f"client {K*n*4/med/1e6:7.0f} GB/s  alg {alg/med/1e6:7.0f} GB/s  frac {alg/med/1e6/8000:.3f}", flush=True)
    res.sort()
    print("best:", res[:5])


if __name__ == "__main__":
    main()
