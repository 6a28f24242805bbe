"""Development tool: the persistent kernel with the late optimizer epilogue (fa_tune stash = -3)
against the default routing, same process, every shape timed in a row; also checks that both give
the same bits.   python tools/late_probe.py "n,K,opt;..." [phase_tiles,...]"""
import sys

import torch

sys.path.insert(0, ".")
from appfl_amd import ops  # noqa: E402

HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
EXTRA = {"none": 2, "avgm": 4}


def timed(fn, reps=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    cfgs = sys.argv[1] if len(sys.argv) > 1 else ";".join(
        f"25557032,{K},adam" for K in (16, 24, 32, 48, 64))
    phases = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    for cfg in cfgs.split(";"):
        n, K, opt = cfg.split(",")
        n, K = int(n), int(K)
        slab = torch.empty((K, n), device="cuda")
        g0 = torch.empty(n, device="cuda")
        ops.synth_fill(g0, 7, scale=0.05)
        for k in range(K):
            ops.synth_fill(slab[k], 1000 + k, scale=0.01, base=g0)
        xs, w = list(slab), [(k + 1) / (K * (K + 1) / 2) for k in range(K)]
        alg = 4 * (K + EXTRA.get(opt, 6)) * n

        def state():
            return g0.clone(), torch.zeros(n, device="cuda"), torch.full((n,), 1e-6, device="cuda")

        def run(g, m, v):
            ops.reduce_opt(xs, w, g, m if opt not in ("none", "avgm") else None,
                           v if opt != "none" else None, opt, **HP)

        # same bits
        ops.tune()
        a = state()
        run(*a), run(*a)
        ops.tune(stash=-3)
        b = state()
        run(*b), run(*b)
        same = all(torch.equal(x, y) for x, y in zip(a, b))
        ops.tune()
        st = state()
        name = ops.reduce_kernel_name(K, n, opt)
        t_def = timed(lambda: run(*st))
        line = f"n={n} K={K} {opt}: default ({name}) {t_def:.4f} ms frac {alg / t_def / 8e9:.3f};"
        for ph in phases:
            ops.tune(stash=-3, nt=1 | (ph << 8))
            t = timed(lambda: run(*st))
            line += f" late[{ph or 10} tiles] {t:.4f} ms frac {alg / t / 8e9:.3f} ({(t / t_def - 1) * 100:+.1f} %);"
        ops.tune()
        print(line + f" same bits: {same}", flush=True)
        del slab, xs, g0, a, b, st
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
