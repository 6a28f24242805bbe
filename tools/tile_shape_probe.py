"""Development tool: the tile-per-block reduce kernel under other launch shapes (fa_tune vpt x
unroll, persistent kernels off) against the default routing, same process, kernel time.
python tools/tile_shape_probe.py "n,K,opt;..." """
import sys

import torch

sys.path.insert(0, ".")
from appfl_amd import ops  # noqa: E402
from tools.late_probe import HP, EXTRA, timed  # noqa: E402


def main():
    for cfg in sys.argv[1].split(";"):
        n, K, opt = cfg.split(",")
        n, K = int(n), int(K)
        slab = torch.empty((K, n), device="cuda")
        g = torch.empty(n, device="cuda")
        ops.synth_fill(g, 7, scale=0.05)
        for k in range(K):
            ops.synth_fill(slab[k], 1000 + k, scale=0.01, base=g)
        xs, w = list(slab), [(k + 1) / (K * (K + 1) / 2) for k in range(K)]
        m, v = torch.zeros(n, device="cuda"), torch.full((n,), 1e-6, device="cuda")
        alg = 4 * (K + EXTRA.get(opt, 6)) * n

        def run():
            ops.reduce_opt(xs, w, g, m if opt not in ("none", "avgm") else None,
                           v if opt != "none" else None, opt, **HP)

        ops.tune()
        line = f"n={n} K={K} {opt}: default ({ops.reduce_kernel_name(K, n, opt)}) {timed(run):.4f} ms;"
        for vpt in (1, 2, 4):
            for unroll in (4, 8, 16):
                ops.tune(vpt, unroll, 1, 0, 0)
                t = timed(run)
                line += f" {vpt}x{unroll} {t:.4f} ({alg / t / 8e9:.3f});"
        ops.tune()
        print(line, flush=True)
        del slab, xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
