"""Development tool: plain FedAvg through the default routing, the pipelined persistent kernel
(fa_tune stash = -1) and the wide-batch persistent kernel (stash = -2), same process, kernel time
per shape; checks that all three give the same bits.   python tools/park_probe.py "n,K;..." [phase_tiles,...]"""
import sys

import torch

sys.path.insert(0, ".")
from appfl_amd import ops  # noqa: E402
from tools.late_probe import timed  # noqa: E402


def main():
    phases = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else []
    for cfg in sys.argv[1].split(";"):
        n, K = (int(x) for x in cfg.split(","))
        slab = torch.empty((K, n), device="cuda")
        g0 = torch.empty(n, device="cuda")
        ops.synth_fill(g0, 7, scale=0.05)
        for k in range(K):
            ops.synth_fill(slab[k], 1000 + k, scale=0.01, base=g0)
        xs, w = list(slab), [(k + 1) / (K * (K + 1) / 2) for k in range(K)]
        alg = 4 * (K + 2) * n
        outs, line = [], f"n={n} K={K}:"
        g = g0.clone()                       # ONE result buffer for every variant (same addresses)
        for label, stash in (("default", None), ("pipelined", -1), ("wide", -2)):
            ops.tune() if stash is None else ops.tune(stash=stash)
            name = ops.reduce_kernel_name(K, n, "none")
            g.copy_(g0)
            ops.reduce_opt(xs, w, g, None, None, "none")
            ops.reduce_opt(xs, w, g, None, None, "none")
            outs.append(g.cpu())
            t = timed(lambda: ops.reduce_opt(xs, w, g, None, None, "none"))
            line += f" {label} ({name}) {t:.4f} ms frac {alg / t / 8e9:.3f};"
        for ph in phases:
            ops.tune(stash=-2, nt=1 | (ph << 8))
            g.copy_(g0)
            ops.reduce_opt(xs, w, g, None, None, "none")
            ops.reduce_opt(xs, w, g, None, None, "none")
            outs.append(g.cpu())
            t = timed(lambda: ops.reduce_opt(xs, w, g, None, None, "none"))
            line += f" wide[{ph} tiles] {t:.4f} ms;"
        ops.tune()
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(line + f" same bits: {same}", flush=True)
        del slab, xs, g0, outs, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
