"""Development tool: the fused FedAdam round against the same round in passes (weighted step
into a buffer, then the optimizer epilogue and the apply as separate launches), kernel time by
events.  python tools/two_pass_probe.py "n,K;n,K;..." """
import sys

import torch

sys.path.insert(0, ".")
from appfl_amd import ops  # noqa: E402

HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)


def timed(fn, reps=12):
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
    cfgs = sys.argv[1] if len(sys.argv) > 1 else "25557032,16;25557032,24;25557032,32;25557032,40"
    for cfg in cfgs.split(";"):
        n, K = (int(t) for t in cfg.split(","))
        slab = torch.empty(K * n, device="cuda")
        g = torch.empty(n, device="cuda")
        ops.synth_fill(g, 7, scale=0.05)
        for k in range(K):
            ops.synth_fill(slab[k * n:], 1000 + k, scale=0.01, base=g, n=n)
        xs = [slab[k * n:(k + 1) * n] for k in range(K)]
        w = [1.0 / K] * K
        m, v = torch.zeros(n, device="cuda"), torch.full((n,), 1e-6, device="cuda")
        step = torch.empty(n, device="cuda")
        alg = 4 * (K + 6) * n

        def fused():
            ops.reduce_opt(xs, w, g, m, v, "adam", **HP)

        def step_only():
            ops.reduce_step(xs, w, g, step)

        def epilogue():
            ops.server_opt_step(step, m, v, "adam", **HP)
            ops.apply_step(g, step)

        tf, ts, te = timed(fused), timed(step_only), timed(epilogue)
        print(f"n={n} K={K}: fused {tf:.4f} ms (frac {alg / tf / 8e9:.3f}); step only {ts:.4f} + "
              f"epilogue in two launches {te:.4f} = {ts + te:.4f} ms (frac {alg / (ts + te) / 8e9:.3f}); "
              f"with a 7-stream epilogue at 6 TB/s: {ts + 28 * n / 6e9:.4f} ms "
              f"(frac {alg / (ts + 28 * n / 6e9) / 8e9:.3f})", flush=True)
        del slab, xs, g, m, v, step
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
