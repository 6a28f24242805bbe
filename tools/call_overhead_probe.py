"""Development tool: what one round of a SMALL model costs per call (batches of 20 calls between
two events, as tools/sweep.py --batch 20 measures): the aggregator's fused launch with its kept
arguments (ops.RoundPlan), the same call marshalled afresh (ops.round_sync), ops.reduce_opt over a
list of tensors, and a library call that does nothing.

    python tools/call_overhead_probe.py [n,K ...]      (default: the MNIST-CNN size, 2 and 10 clients)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import appfl_amd.aggregator as A  # noqa: E402
from appfl_amd import _native as N, ops, workloads as W  # noqa: E402
from appfl_amd.layout import STEP  # noqa: E402


def per_call(fn, batch=20, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gpu, host = [], []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        t0 = time.perf_counter()
        for _ in range(batch):
            fn()
        t1 = time.perf_counter()
        b.record()
        b.synchronize()
        gpu.append(a.elapsed_time(b) * 1e3 / batch)
        host.append((t1 - t0) * 1e6 / batch)
    gpu.sort(), host.sort()
    return gpu[len(gpu) // 2], host[len(host) // 2]


def main():
    shapes = [tuple(int(t) for t in s.split(",")) for s in sys.argv[1:]] or [(0, 2), (0, 10), (0, 32)]
    for n, K in shapes:
        for cls in ("FedAvgAggregator", "FedAdamAggregator"):
            if n == 0:
                rows = W.mnist_cnn()
                state = {name: torch.empty(shape, dtype=dtype) for name, shape, dtype, _ in rows}
            else:
                state = {"w": torch.empty(n)}
            agg = getattr(A, cls)(None, dict(W.EXAMPLE_CONFIGS[cls]), None)
            agg._init_global_state(state)
            lay = agg._layout
            n_step = lay.count[STEP]
            g = agg._flats["f32"]
            ops.synth_fill(g, 7, scale=0.05, n=lay.n_f32)
            agg._slab.reserve(K)
            agg._slab.slab.zero_()
            for k in range(K):
                ops.synth_fill(agg._slab.row_view(k, 0, lay.n_f32, torch.float32), 1000 + k, scale=0.01,
                               base=g, n=lay.n_f32)
            w = [1.0 / K] * K
            agg._ensure_moments()
            hyper = agg._hyper()
            ptrs = agg._rows(K, 0)
            tens = [agg._slab.row_view(k, 0, n_step, torch.float32) for k in range(K)]
            lib = N.load()
            tiny = torch.zeros(4, device="cuda")
            agg._launch_round(K, w, True)
            plan = next(iter(agg._round_plans.values()))
            res = {
                "aggregator launch (kept arguments)": per_call(lambda: agg._launch_round(K, w, True)),
                "the kept plan's launch alone": per_call(plan.launch),
                "ops.round_sync (marshalled per call)": per_call(
                    lambda: ops.round_sync(ptrs, w, g, agg._m, agg._v, agg._OPT, n=n_step, **hyper)),
                "ops.reduce_opt (tensor list)": per_call(
                    lambda: ops.reduce_opt(tens, w, g, agg._m, agg._v, agg._OPT, n=n_step, **hyper)),
                "ops.fill of 4 elements (a bare HIP launch)": per_call(lambda: ops.fill(tiny, 1.0)),
                "library call that does nothing": per_call(lambda: lib.fa_kchunk()),
            }
            print(f"== {cls} n_step={n_step} K={K}")
            for what, (gpu, host) in res.items():
                print(f"   {what:40s} {gpu:7.2f} us per call between events, {host:6.2f} us of host time", flush=True)
            del agg


if __name__ == "__main__":
    main()
