"""Device-resident aggregation throughput of the MI355X aggregator hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one pass of the hot path over one batch of synthetic client updates that are already
resident in HBM: for the FedAvg family one round (the fused weighted-step / server-optimizer /
apply kernel over the K staged client rows plus the small plain-mean kernels), for FedAsync /
FedBuff one arrival.  `value` = client-update bytes consumed per second over all ranks.

N = 1: the headline workload (ResNet-18 FedAvg, K = 100) is timed as `value`; the other
benchmark configurations are measured in the same process and listed under "workloads" (kernel
time from events on the launch stream, roofline fraction), the north-star target configuration
under "north_star_target", the PCIe-inclusive aggregate() under "end_to_end", and the whole
1.3 B-parameter K = 32 round on this one GPU under "strong_scaling_reference".

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of the 1.3 B-parameter
K = 32 FedAvg round (or of --workload's flat vector): the parameter axis is dealt to the ranks in
blocks (appfl_amd.sharded.ShardMap), every rank aggregates its blocks of every client update and
RCCL all-gathers hand every rank the whole new model.  `value` times the product path
(`round_sliced_gather`: kernels and gathers pipelined over the block rows); "no_collective",
"allgather" (round, then gather) and "allgather_pipelined" report the three forms side by side;
"workloads" holds the same three forms for the ResNet-50 K = 64 FedAdam / FedYogi rounds
(BASELINE configuration 3, which is quoted on 1/2/4/8 GPUs) dealt to the same ranks.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

METRIC = "device-resident FedAvg aggregation GB/s (K clients x params), % HBM peak"
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (datasheet)
DEFAULT_WORKLOAD = "resnet18_fedavg_k100"
SHARDED_WORKLOAD = "flat1300m_fedavg_k32"
NORTH_STAR_WORKLOAD = "resnet50_fedadam_k64"
SHARDED_EXTRA_WORKLOADS = ["resnet50_fedadam_k64", "resnet50_fedyogi_k64"]
EXTRA_WORKLOADS = ["resnet50_fedadam_k64", "resnet50_fedyogi_k64", "resnet50_fedavgm_k64",
                   "resnet18_fedavg_k100_bf16", "resnet50_fedadam_k64_bf16", "resnet50_fedadam_k64_f16",
                   "vitb16_fedasync_bf16",
                   "vitb16_fedcompass_group_k8_bf16", "flat162m_fedavg_k32", "vitb16_fedasync", "vitb16_fedbuff_k16",
                   "vitb16_fedbuff_first", "vitb16_fedbuff_middle", "vitb16_fedbuff_flush",
                   "vitb16_fedcompass_group_k8", "vitb16_fedqueue_replace_k16"]
# workloads also timed with COLD state vectors (see SyncBench): the headline, the north-star
# target and the long flat vector
COLD_WORKLOADS = {"resnet18_fedavg_k100", "resnet50_fedadam_k64", "flat162m_fedavg_k32"}
INFINITY_CACHE_BYTES = 256 << 20
_OPT_OF = {"FedAvgAggregator": "none", "FedAvgMAggregator": "avgm", "FedAdamAggregator": "adam",
           "FedYogiAggregator": "yogi", "FedAdagradAggregator": "adagrad"}
STATE_VECTORS = {"none": 2, "avgm": 4, "adam": 6, "yogi": 6, "adagrad": 6}  # g/m/v reads + writes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None,
                    help=f"default: {DEFAULT_WORKLOAD} on one GPU, {SHARDED_WORKLOAD} sharded on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true",
                    help="one GPU: time the chosen workload only (no extra workloads, no end-to-end)")
    ap.add_argument("--sharded", action="store_true",
                    help="run the block-dealt multi-GPU form also on one rank (rehearsal)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-big", action="store_true",
                    help="one GPU: skip the 166 GB strong-scaling reference run")
    ap.add_argument("--cold", action="store_true",
                    help="with --only: time the workload with its state vectors rotated over independent "
                         "sets (cold state) instead of re-used ones -- for profiler runs of the cold figure")
    ap.add_argument("--tune-stash", type=int, default=None,
                    help="benchmark knob for A/B profiles: fa_tune's stash value (0: tile-per-block kernel "
                         "only, -1 / -2 / -3: force one of the persistent kernels)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="CPU time budget of each cpu_baseline leg")
    return ap.parse_args()


def template_state(rows):
    return {name: torch.empty(shape, dtype=dtype) for name, shape, dtype, _ in rows}


class SyncBench:
    """FedAvg-family workload: K synthetic client rows in the device slab, one round per step.

    In a benchmark loop the state vectors (g, and m / v under the optimizers) are the same buffers
    every step, and vectors of up to 256 MB can stay in the Infinity Cache from one step to the
    next, which a real round -- gigabytes of client uploads in between -- would not see.
    `cold_sets` > 1 rotates that many independent sets of state vectors (together at least twice
    the cache), so that no step reads state an earlier step left in the cache."""

    def __init__(self, wl, rank, cold_sets=0):
        import appfl_amd.aggregator as A
        from appfl_amd import ops, workloads as W
        from appfl_amd.layout import MEAN_I32, MEAN_I64, STEP
        self.ops = ops
        rows = wl["table"]()
        self.K = wl["K"]
        self.agg = getattr(A, wl["cls"])(None, dict(W.EXAMPLE_CONFIGS[wl["cls"]]), None)
        self.agg._init_global_state(template_state(rows))
        lay = self.agg._layout
        self.n_step = lay.count[STEP]
        half = lay.half_dtype      # a 16-bit model: 2-byte elements in the rows and in the state
        item = 2 if half is not None else 4
        self.payload = item * lay.n_f32 + 8 * lay.count[MEAN_I64] + 4 * lay.count[MEAN_I32]
        g = self.agg._flats["f32"]
        slab = self.agg._slab
        slab.reserve(self.K)
        slab.slab.zero_()
        if half is None:
            ops.synth_fill(g, 7 + 1000003 * rank, scale=0.05, n=lay.n_f32)
            for k in range(self.K):
                ops.synth_fill(slab.row_view(k, 0, lay.n_f32, torch.float32), 1000 + k + 1000003 * rank,
                               scale=0.01, base=g, n=lay.n_f32)
        else:
            # the same synthetic values, generated in fp32 and rounded to the 16-bit type
            g32, x32 = (torch.empty(lay.n_f32, dtype=torch.float32, device=g.device) for _ in range(2))
            ops.synth_fill(g32, 7 + 1000003 * rank, scale=0.05, n=lay.n_f32)
            g[:lay.n_f32].copy_(g32)
            g32.copy_(g[:lay.n_f32])
            for k in range(self.K):
                ops.synth_fill(x32, 1000 + k + 1000003 * rank, scale=0.01, base=g32, n=lay.n_f32)
                slab.row_view(k, 0, lay.n_f32, half).copy_(x32)
            del g32, x32
        self.weights = [1.0 / self.K] * self.K
        self.opt = self.agg._OPT
        if half is None:
            self.kernel = "%s<%s>" % (ops.reduce_kernel_name(self.K, self.n_step, self.opt), self.opt)
        else:
            from appfl_amd import half_ops
            self.kernel = "%s<%s,%s>" % (half_ops.HalfOps(half).reduce_kernel_name(self.K, self.n_step, self.opt),
                                         self.opt, str(half).split(".")[-1])
        self.alg_bytes = item * (self.K + STATE_VECTORS[self.opt]) * self.n_step
        self.client_bytes = self.K * self.payload
        self.result_numel = lay.n_f32
        self.sets, self.turn = [], 0
        if cold_sets > 1:
            self.agg._ensure_moments()
            first = (self.agg._flats["f32"], self.agg._m, self.agg._v)
            self.sets = [first] + [tuple(None if t is None else t.clone() for t in first)
                                   for _ in range(cold_sets - 1)]

    @staticmethod
    def cold_sets_for(wl):
        """Sets of state vectors whose total is at least twice the Infinity Cache (3 at least)."""
        from appfl_amd import workloads as W
        n = W.numel([r for r in wl["table"]() if r[2] == torch.float32])
        per_set = 4 * n * (STATE_VECTORS[_OPT_OF[wl["cls"]]] // 2)
        return max(3, min(16, -(-2 * INFINITY_CACHE_BYTES // per_set)))

    def step(self, events=None):
        if self.sets:
            self.turn = (self.turn + 1) % len(self.sets)
            self.agg._flats["f32"], self.agg._m, self.agg._v = self.sets[self.turn]
        self.agg._launch_round(self.K, self.weights, True, events)

    def result(self):
        return self.agg._flats["f32"][:self.result_numel]


class AsyncBench:
    """FedAsync / FedBuff workload: one staged client update, one arrival per step."""

    def __init__(self, wl, rank):
        import appfl_amd.aggregator as A
        from appfl_amd import ops, workloads as W
        from appfl_amd.layout import STEP
        self.ops = ops
        rows = wl["table"]()
        self.agg = getattr(A, wl["cls"])(None, dict(W.EXAMPLE_CONFIGS[wl["cls"]]), None)
        self.agg._init_global_state(template_state(rows))
        lay = self.agg._layout
        self.buff = hasattr(self.agg, "buff_size")
        if self.buff:
            self.agg._step_flats = lay.alloc_state(self.agg.device)
        self.n_step = lay.count[STEP]
        half = lay.half_dtype
        item = 2 if half is not None else 4
        g = self.agg._flats["f32"]
        self.agg._slab.slab.zero_()
        g32 = g if half is None else torch.empty(lay.n_f32, dtype=torch.float32, device=g.device)
        ops.synth_fill(g32, 7 + 1000003 * rank, scale=0.05, n=lay.n_f32)
        if half is None:
            ops.synth_fill(self.agg._slab.row_view(0, 0, lay.n_f32, torch.float32), 1000 + 1000003 * rank,
                           scale=0.01, base=g, n=lay.n_f32)
        else:                      # the same values rounded to the 16-bit type
            g[:lay.n_f32].copy_(g32)
            g32.copy_(g[:lay.n_f32])
            x32 = torch.empty_like(g32)
            ops.synth_fill(x32, 1000 + 1000003 * rank, scale=0.01, base=g32, n=lay.n_f32)
            self.agg._slab.row_view(0, 0, lay.n_f32, half).copy_(x32)
            del g32, x32
        self.K = wl["K"]
        self.i = 0
        self.kind = wl.get("kind")          # FedBuff: time only first / middle / flushing arrivals
        self.kernel = ("buff_accum_kernel" if self.buff else "async_mix_kernel") if half is None else \
            "arrival_half_kernel<%s,%s>" % ("buff" if self.buff else "async", str(half).split(".")[-1])
        # x, g read + g write; FedBuff: x, g, step read + step write -- the first arrival of a
        # buffer skips the step read (12) and the flushing one adds the g write (20), so 16 on
        # average over a buffer
        per = {"first": 12, "middle": 16, "flush": 20}.get(self.kind, 16) if self.buff else 12
        self.alg_bytes = per * item // 4 * self.n_step
        if self.kind:
            self.kernel += "<%s>" % self.kind
        self.client_bytes = item * lay.n_f32
        self.result_numel = lay.n_f32

    def step(self, events=None):
        # scripted staleness 0..15, polynomial a = 0.5, alpha = 0.9, 16 clients
        u = self.i % 16
        alpha_t = 0.9 * (u + 1) ** (-0.5) * (1.0 / 16)
        if events:
            events[0].record()
        if self.buff:
            pos = self.i % self.K
            if self.kind:
                pos = {"first": 0, "middle": 1, "flush": self.K - 1}[self.kind]
            self.agg._launch_arrival(alpha_t, pos == 0, pos == self.K - 1)
        else:
            self.agg._launch_arrival(alpha_t)
        if events:
            events[1].record()
        self.i += 1

    def result(self):
        return self.agg._flats["f32"][:self.result_numel]


class MixBench:
    """FedCompass / FedQueue combine (fa_weighted_mix) over K rows of the model's parameters
    resident in HBM: `convex` g = (1 - sum w) g + sum w_k x_k (a FedCompass group arrival,
    fedcompass_aggregator.py:152-338) or `replace` g = sum w_k x_k (FedQueue,
    fedqueue_aggregator.py:75-131).  One call per step."""

    def __init__(self, wl, rank):
        from appfl_amd import half_ops, ops, workloads as W
        rows = wl["table"]()
        floats = (torch.float32, torch.bfloat16, torch.float16)
        dtype = next(r[2] for r in rows if r[2] in floats)
        half = dtype if dtype != torch.float32 else None
        item = 2 if half is not None else 4
        self.ops = ops if half is None else half_ops.HalfOps(half)
        n = self.n_step = W.numel([r for r in rows if r[2] in floats], params_only=True)
        K = self.K = wl["K"]
        self.mode = wl["mode"]
        stride = (n + 63) // 64 * 64
        dev = torch.device("cuda", torch.cuda.current_device())
        g32 = torch.empty(n, dtype=torch.float32, device=dev)
        ops.synth_fill(g32, 7 + 1000003 * rank, scale=0.05)
        self.g = g32.to(dtype)
        g32.copy_(self.g)              # a 16-bit model: the same values, rounded
        self.slab = torch.empty((K, stride), dtype=dtype, device=dev)
        x32 = torch.empty_like(g32)
        for k in range(K):
            ops.synth_fill(x32, 1000 + k, scale=0.01, base=g32, n=n)
            self.slab[k, :n].copy_(x32)
        del g32, x32
        self.rows = [self.slab[k] for k in range(K)]
        self.weights = [0.5 / K] * K if self.mode == "convex" else [1.0 / K] * K
        self.kernel = "mix_kernel<%s>" % self.mode if half is None else \
            "mix_half_kernel<%s,%s>" % (self.mode, str(half).split(".")[-1])
        # K client reads + g written (replace) or read and written (convex)
        self.alg_bytes = item * (K + (2 if self.mode == "convex" else 1)) * n
        self.client_bytes = item * K * n
        self.result_numel = n

    def step(self, events=None):
        if events:
            events[0].record()
        self.ops.weighted_mix(self.rows, self.weights, self.g, self.mode, coeff=0.5, n=self.n_step)
        if events:
            events[1].record()

    def result(self):
        return self.g


class AdmmBench:
    """IIADMM / ICEADMM workload: K clients' primal, dual and previous-primal rows of the model's
    parameters in HBM, one server update (fa_admm_ii / fa_admm_ice) per step.  The timed call
    includes the small launches that total the residual norms."""

    def __init__(self, wl, rank):
        from appfl_amd import _native as N, ops, workloads as W
        self.ops = ops
        rows = wl["table"]()
        n = self.n_step = W.numel([r for r in rows if r[2] == torch.float32], params_only=True)
        K = self.K = wl["K"]
        self.alg = "ii" if wl["cls"] == "IIADMMAggregator" else "ice"
        stride = (n + 63) // 64 * 64
        dev = torch.device("cuda", torch.cuda.current_device())
        self.g = torch.empty(n, dtype=torch.float32, device=dev)
        ops.synth_fill(self.g, 7 + 1000003 * rank, scale=0.05)
        self.slabs = torch.empty((3, K, stride), dtype=torch.float32, device=dev)
        for k in range(K):
            ops.synth_fill(self.slabs[0, k], 1000 + k, scale=0.01, base=self.g, n=n)     # primal
            ops.synth_fill(self.slabs[1, k], 3000 + k, scale=0.5, n=n)                   # dual
            ops.synth_fill(self.slabs[2, k], 2000 + k, scale=0.01, base=self.g, n=n)     # previous primal
        self.rows = [[self.slabs[j, k] for k in range(K)] for j in range(3)]
        self.penalty = [500.0 / (1 + k % 4) for k in range(K)]
        self.norms = torch.zeros(2, dtype=torch.float64, device=dev)
        self.res_ws = torch.empty(N.FA_ADMM_RES_WS, dtype=torch.float64, device=dev)
        self.carry = (torch.empty(ops.admm_carry_floats(n), dtype=torch.float32, device=dev)
                      if K > N.FA_ADMM_KCHUNK else None)
        self.kernel = "admm_kernel<%s>" % self.alg
        # per client and parameter: primal, dual, previous primal read (+ dual written back by
        # IIADMM); g read and written once
        self.alg_bytes = 4 * ((4 if self.alg == "ii" else 3) * K + 2) * n
        self.client_bytes = 4 * (1 if self.alg == "ii" else 2) * K * n      # what the clients sent
        self.result_numel = n

    def step(self, events=None):
        if events:
            events[0].record()
        self.ops.admm_round(self.alg, self.rows[0], self.rows[1], self.rows[2], self.penalty, self.g,
                            self.norms, self.res_ws, num_clients=self.K, carry_ws=self.carry, n=self.n_step)
        if events:
            events[1].record()

    def result(self):
        return self.g


def bench_class(wl):
    if wl["cls"] in ("FedAsyncAggregator", "FedBuffAggregator"):
        return AsyncBench
    if wl["cls"] in ("IIADMMAggregator", "ICEADMMAggregator"):
        return AdmmBench
    if wl["cls"] in ("FedCompassAggregator", "FedQueueAggregator"):
        return MixBench
    return SyncBench


def cpu_baseline(wl, budget_s):
    """Times the oracle on host cores over a bounded sample of the same workload (rank 0, N = 1).
    Returns (reference-like per-tensor torch loops, fused C/OpenMP restatement)."""
    import numpy as np
    from appfl_amd import workloads as W
    from oracle import c_oracle as OC
    from oracle import torch_naive as TN

    rows = wl["table"]()
    cls = wl["cls"]
    cfg = W.EXAMPLE_CONFIGS[cls]
    opt = {"FedAvgAggregator": "none", "FedAvgMAggregator": "avgm", "FedAdamAggregator": "adam",
           "FedYogiAggregator": "yogi", "FedAdagradAggregator": "adagrad"}.get(cls)
    total = W.numel(rows)
    is_async = opt is None
    # sample: whole state dicts, fewer clients (cost is linear in the client count)
    Kc = 1 if is_async else max(2, min(wl["K"], int(3.0e9 // (4 * total)) or 2, 16))
    threads = torch.get_num_threads()

    def client_state(seed, base):
        out, off = {}, 0
        for name, shape, dtype, _ in rows:
            n = 1
            for s in shape:
                n *= s
            if dtype == torch.float32:
                b = None if base is None else base[name].numpy().ravel()
                a = OC.synth_fill(n, seed, first=off, scale=0.05 if base is None else 0.01, base=b)
                out[name] = torch.from_numpy(a.reshape(shape))
            else:
                out[name] = torch.full(shape, seed % 7, dtype=dtype)
            off += n
        return out

    base = client_state(7, None)
    clients = {k: client_state(1000 + k, base) for k in range(Kc)}
    sample_bytes = Kc * 4 * total

    def timed(fn, max_reps=7):
        fn()                                    # warm-up
        ts, t_end = [], time.perf_counter() + budget_s
        while len(ts) < max_reps and (time.perf_counter() < t_end or len(ts) < 2):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    if is_async:
        g = {k: v.clone() for k, v in base.items()}
        x = clients[0]

        def naive():
            TN.naive_async_mix(g, {k: v.clone() for k, v in x.items()}, 0.05)
        # the clone of the arriving model stands for its deserialisation; subtract its cost
        t_clone = timed(lambda: {k: v.clone() for k, v in x.items()})
        t_naive = max(timed(naive) - t_clone, 1e-9)
    else:
        agg = TN.NaiveFedAvg(opt, cfg)
        agg.global_state = {k: v.clone() for k, v in base.items()}
        w = {k: 1.0 / Kc for k in range(Kc)}
        t_naive = timed(lambda: agg.aggregate(clients, w))
    naive_out = {"value": round(sample_bytes / t_naive / 1e9, 3), "unit": "GB/s", "cores": threads,
                 "kind": "port",
                 "sample": f"{Kc} of {wl['K']} client updates x full {total}-element state dict, "
                           f"per-tensor torch CPU loops as in the reference, median of repeats"}

    # fused C / OpenMP restatement on flat fp32 vectors of the same size
    n = W.numel([r for r in rows if r[2] == torch.float32])
    if n == 0:      # a 16-bit workload: the C restatement is fp32 only
        return naive_out, {"skipped": "the fused C restatement covers fp32 vectors only"}
    gf = np.concatenate([base[r[0]].numpy().ravel() for r in rows if r[2] == torch.float32])
    xs = [np.concatenate([clients[k][r[0]].numpy().ravel() for r in rows if r[2] == torch.float32])
          for k in range(Kc)]
    if is_async:
        t_c = timed(lambda: OC.async_mix(xs[0], 0.05, 0, 1, gf))
    else:
        m, v = np.zeros(n, np.float32), np.full(n, np.float32(1e-6))
        code = {"none": 0, "avgm": 1, "adam": 2, "yogi": 3, "adagrad": 4}[opt]
        wl_ = [1.0 / Kc] * Kc
        t_c = timed(lambda: OC.reduce_opt(xs, wl_, gf, m, v, code, lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3))
    fused_out = {"value": round(Kc * 4 * n / t_c / 1e9, 3), "unit": "GB/s",
                 "cores": int(os.environ.get("OMP_NUM_THREADS", threads)), "kind": "port",
                 "sample": f"{Kc} client updates x {n} fp32, fused C/OpenMP restatement"}
    return naive_out, fused_out


def end_to_end(wl, rounds=4, arrivals=7):
    """aggregate() of the drop-in class as the scheduler calls it: K CPU state dicts in, one CPU
    state dict out (pack + pinned H2D of every client, kernels, D2H).  PCIe-inclusive; reported
    beside `value`, never as `value`."""
    import appfl_amd.aggregator as A
    from appfl_amd import ops, workloads as W

    rows, K = wl["table"](), wl["K"]
    floats = (torch.float32, torch.bfloat16, torch.float16)     # a 16-bit model: the same values, rounded
    nf = W.numel([r for r in rows if r[2] in floats])
    base = torch.empty(nf, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, 7, scale=0.05)
    buf = torch.empty_like(base)

    def state(seed):
        if seed is None:
            flat = base.cpu()
        else:
            ops.synth_fill(buf, seed, scale=0.01, base=base)
            flat = buf.cpu()
        out, off = {}, 0
        for name, shape, dtype, _ in rows:
            n = 1
            for d in shape:
                n *= d
            if dtype in floats:
                out[name] = flat[off:off + n].view(shape).to(dtype, copy=True)
                off += n
            else:
                out[name] = torch.full(shape, 3, dtype=dtype)
        return out

    clients = {k: state(1000 + k) for k in range(K)}
    nbytes = sum(t.numel() * t.element_size() for t in clients[0].values())
    del base, buf
    # ServerAgent copies the number of clients into the aggregator's configuration (agent/server.py:652-662)
    agg = getattr(A, wl["cls"])(None, dict(W.EXAMPLE_CONFIGS[wl["cls"]], num_clients=K), None)
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agg.aggregate(clients)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts[1:])[len(ts[1:]) // 2]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agg._stage(clients)
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t0
    # clients staged as they arrive (aggregator.stage): what is left after the LAST arrival
    ids = list(clients)
    tl = []
    for _ in range(arrivals):
        for c in ids[:-1]:
            agg.stage(c, clients[c])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agg.stage(ids[-1], clients[ids[-1]])
        agg.aggregate(clients)
        tl.append(time.perf_counter() - t0)
    return {"value": round(K * nbytes / t / 1e9, 2), "unit": "GB/s", "ms_per_round": round(t * 1e3, 2),
            "after_last_arrival_ms": round(sorted(tl)[len(tl) // 2] * 1e3, 2),
            "first_round_ms": round(ts[0] * 1e3, 2), "stage_ms": round(t_stage * 1e3, 2),
            "stage_GBps": round(K * nbytes / t_stage / 1e9, 2),
            "what": f"aggregate() of {K} CPU state dicts ({nbytes} B each) -> CPU state dict, "
                    "median of rounds 2.."}


def end_to_end_wire(wl, rounds=3):
    """The round as a communicator sees it, from wire bytes to wire bytes: K `torch.save`
    payloads (comm/mpi/serializer.py:46-50, comm/grpc/utils.py:20-24) -> new global model as
    `torch.save`-compatible bytes.  `device_wire_path`: `stage_bytes` per arrival (tensors read in
    place from the payload and packed straight into the pinned row), `aggregate`,
    `wire.model_to_bytes`.  `torch_io_path`: what the unmodified server does around the same
    aggregator -- `torch.load(map_location="cpu")` per arrival (agent/server.py:505-513),
    `aggregate`, `torch.save` (comm/mpi/serializer.py:46-50).  Host-inclusive; never `value`."""
    import io

    import appfl_amd.aggregator as A
    from appfl_amd import ops, wire, workloads as W

    rows, K = wl["table"](), wl["K"]
    nf = W.numel([r for r in rows if r[2] == torch.float32])
    base = torch.empty(nf, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, 7, scale=0.05)
    buf = torch.empty_like(base)
    payloads = []
    for k in range(K):
        ops.synth_fill(buf, 1000 + k, scale=0.01, base=base)
        flat, state, off = buf.cpu(), {}, 0
        for name, shape, dtype, _ in rows:
            n = 1
            for d in shape:
                n *= d
            if dtype == torch.float32:
                state[name] = flat[off:off + n].view(shape)
                off += n
            else:
                state[name] = torch.full(shape, 3, dtype=dtype)
        b = io.BytesIO()
        torch.save(state, b)
        payloads.append(b.getvalue())
    del base, buf
    nbytes = sum(len(p) for p in payloads)

    def device_path(agg):
        t0 = time.perf_counter()
        models = {}
        for k, p in enumerate(payloads[:-1]):
            models[k] = agg.stage_bytes(k, p)
        t1 = time.perf_counter()
        models[K - 1] = agg.stage_bytes(K - 1, payloads[-1])
        out = agg.aggregate(models)
        reply = wire.model_to_bytes(out)
        t2 = time.perf_counter()
        return t2 - t0, t2 - t1, len(reply)

    def torch_path(agg):
        t0 = time.perf_counter()
        models = {}
        for k, p in enumerate(payloads[:-1]):
            models[k] = torch.load(io.BytesIO(p), map_location="cpu")
        t1 = time.perf_counter()
        models[K - 1] = torch.load(io.BytesIO(payloads[-1]), map_location="cpu")
        out = agg.aggregate(models)
        b = io.BytesIO()
        torch.save(out, b)
        t2 = time.perf_counter()
        return t2 - t0, t2 - t1, b.tell()

    res = {}
    for label, fn in (("device_wire_path", device_path), ("torch_io_path", torch_path)):
        agg = getattr(A, wl["cls"])(None, dict(W.EXAMPLE_CONFIGS[wl["cls"]], num_clients=K), None)
        runs = [fn(agg) for _ in range(rounds)]
        total, last, reply = sorted(runs[1:])[len(runs[1:]) // 2]
        res[label] = {"ms_per_round": round(total * 1e3, 2), "after_last_arrival_ms": round(last * 1e3, 2),
                      "value": round(nbytes / total / 1e9, 2), "unit": "GB/s", "reply_bytes": reply}
        del agg
        torch.cuda.empty_cache()
    res["what"] = (f"{K} torch.save payloads ({nbytes} B in all) -> reply bytes; arrivals back to back on one "
                   "thread, median of rounds 2..")
    return res


class ShardedBench:
    """The flat vector of a FedAvg-family workload dealt to the ranks in blocks; K synthetic
    client rows per rank, generated on the device so that the ranks' blocks are pieces of ONE
    consistent set of client vectors."""

    def __init__(self, wl, world, rank):
        from appfl_amd import ops, workloads as W
        from appfl_amd.sharded import ShardedFlatAggregator
        self.ops = ops
        rows = wl["table"]()
        self.n_total = W.numel([r for r in rows if r[2] == torch.float32 and r[3]])
        self.K = wl["K"]
        self.opt = _OPT_OF[wl["cls"]]
        cfg = W.EXAMPLE_CONFIGS[wl["cls"]]
        hp = dict(lr=cfg.get("server_learning_rate", 0.0), tau=cfg.get("server_adapt_param", 0.0),
                  beta1=cfg.get("server_momentum_param_1", 0.0),
                  beta2=cfg.get("server_momentum_param_2", 0.0))
        self.agg = ShardedFlatAggregator(self.n_total, self.opt, **hp)
        smap = self.map = self.agg.map
        g = self.agg.g
        self.slab = torch.zeros((self.K, smap.L), dtype=torch.float32, device=g.device)
        for lo, hi, loc in smap.windows():
            ops.synth_fill(g[loc:], 7, first=lo, scale=0.05, n=hi - lo)
            for k in range(self.K):
                ops.synth_fill(self.slab[k, loc:], 1000 + k, first=lo, scale=0.01, base=g[loc:],
                               n=hi - lo)
        self.rows = [self.slab[k] for k in range(self.K)]
        self.weights = [1.0 / self.K] * self.K
        self.out = self.agg.new_gather_buffer()
        self.kernel = "%s<%s>" % (ops.reduce_kernel_name(self.K, max(smap.n_local, 1), self.opt), self.opt)
        self.alg_bytes = 4 * (self.K + STATE_VECTORS[self.opt]) * smap.n_local
        self.client_bytes = 4 * self.K * smap.n_local          # this rank's share of the client bytes

    def step_local(self, events=None):
        if events:
            events[0].record()
        self.agg.round_sliced(self.rows, self.weights)
        if events:
            events[1].record()

    def step_gather(self, events=None):
        self.agg.round_sliced(self.rows, self.weights)
        self.agg.all_gather_global(self.out)

    def step_pipelined(self, events=None):
        self.agg.round_sliced_gather(self.rows, self.weights, self.out)


class WholeUpdateBench:
    """The same round in the WHOLE-UPDATE mode: rank r holds the whole vectors of the clients
    k = r (mod R) only; partial steps, RCCL reduce-scatter, optimizer and apply on the own blocks,
    all-gather (ShardedFlatAggregator.round_whole).  Consumes the same client bytes per step."""

    def __init__(self, wl, world, rank):
        from appfl_amd import ops, workloads as W
        from appfl_amd.sharded import ShardedFlatAggregator
        rows = wl["table"]()
        n = self.n_total = W.numel([r for r in rows if r[2] == torch.float32 and r[3]])
        K = self.K = wl["K"]
        opt = _OPT_OF[wl["cls"]]
        cfg = W.EXAMPLE_CONFIGS[wl["cls"]]
        hp = dict(lr=cfg.get("server_learning_rate", 0.0), tau=cfg.get("server_adapt_param", 0.0),
                  beta1=cfg.get("server_momentum_param_1", 0.0),
                  beta2=cfg.get("server_momentum_param_2", 0.0))
        self.agg = ShardedFlatAggregator(n, opt, **hp)
        dev = self.agg.g.device
        base = torch.empty(n, dtype=torch.float32, device=dev)
        ops.synth_fill(base, 7, scale=0.05)
        self.agg.load_global(base)
        mine = [k for k in range(K) if k % world == rank]
        self.slab = torch.empty((max(len(mine), 1), n), dtype=torch.float32, device=dev)
        for i, k in enumerate(mine):
            ops.synth_fill(self.slab[i], 1000 + k, scale=0.01, base=base)
        del base
        self.rows = [self.slab[i] for i in range(len(mine))]
        self.weights = [1.0 / K] * len(mine)
        self.client_bytes = 4 * len(mine) * n

    def step(self, events=None):
        self.agg.round_whole(self.rows, self.weights)


def time_steps(step, steps, warmup, barrier, use_dist, with_events=True):
    """`warmup` untimed steps, then `steps` steps between barriers; returns (seconds -- the max
    over ranks --, sorted per-step kernel ms from events on the launch stream)."""
    for _ in range(warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(steps)] if with_events else [None] * steps
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(evs[i])
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs) if with_events else []
    return elapsed, kernel_ms


def roofline_of(b, kernel_ms, workload):
    avg = sum(kernel_ms) / len(kernel_ms)
    achieved = b.alg_bytes / (avg * 1e-3) / 1e9
    traffic, source = None, None
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get(workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        if traffic is not None:
            source = ("profiles/pmc_traffic.json: rocprofv3 --pmc passes of this workload recorded "
                      "earlier (2*FETCH_SIZE + WRITE_SIZE per launch), not measured in this run")
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
            "kernel": b.kernel, "kernel_avg_ms": round(avg, 5), "kernel_min_ms": round(kernel_ms[0], 5),
            "algorithmic_bytes_per_launch": int(b.alg_bytes)}


def measure_cold(wl, name, steps, warmup, barrier):
    """The same workload with its state vectors rotated over independent sets (cold state: no
    step re-reads what an earlier one left in the Infinity Cache)."""
    sets = SyncBench.cold_sets_for(wl)
    b = SyncBench(wl, 0, cold_sets=sets)
    elapsed, kernel_ms = time_steps(b.step, steps, max(warmup, sets), barrier, False)
    r = roofline_of(b, kernel_ms, name + "_cold")
    out = {"frac_cold": r["frac"], "kernel_avg_ms_cold": r["kernel_avg_ms"],
           "kernel_min_ms_cold": r["kernel_min_ms"], "achieved_cold": r["achieved"],
           "traffic_cold": r["traffic"],
           "value_cold": round(b.client_bytes * steps / elapsed / 1e9, 2),
           "ms_per_step_cold": round(elapsed / steps * 1e3, 4), "state_sets": sets}
    del b
    torch.cuda.empty_cache()
    return out


def measure_workload(name, steps, warmup, barrier):
    """One extra workload on this GPU: same timing as the headline, its objects freed after."""
    from appfl_amd import workloads as W
    wl = W.WORKLOADS[name]
    b = bench_class(wl)(wl, 0)
    elapsed, kernel_ms = time_steps(b.step, steps, warmup, barrier, False)
    r = roofline_of(b, kernel_ms, name)
    out = {"workload": name, "aggregator": wl["cls"], "clients": wl["K"], "steps": steps,
           "value": round(b.client_bytes * steps / elapsed / 1e9, 2), "unit": "GB/s",
           "ms_per_step": round(elapsed / steps * 1e3, 4), "kernel": r["kernel"],
           "kernel_avg_ms": r["kernel_avg_ms"], "kernel_min_ms": r["kernel_min_ms"],
           "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
           "achieved": r["achieved"], "frac": r["frac"], "traffic": r["traffic"]}
    del b
    torch.cuda.empty_cache()
    if name in COLD_WORKLOADS:
        out.update(measure_cold(wl, name, steps, warmup, barrier))
    return out


def measure_sharded(name, wl, world, rank, steps, warmup, barrier, use_dist):
    """One FedAvg-family workload dealt to the ranks: the three timed forms (every rank runs this,
    the gathers are collectives), the local kernel's roofline entry and the shard geometry."""
    b = ShardedBench(wl, world, rank)
    # the product path first (kernels and gathers pipelined over the block rows)
    elapsed, _ = time_steps(b.step_pipelined, steps, warmup, barrier, use_dist, False)
    el_seq, _ = time_steps(b.step_gather, steps, 2, barrier, use_dist, False)
    el_loc, kernel_ms = time_steps(b.step_local, steps, 2, barrier, use_dist)
    total_client = torch.tensor([b.client_bytes], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(total_client)
    total_client = float(total_client.item())

    def form(el):
        return {"value": round(total_client * steps / el / 1e9, 2), "unit": "GB/s",
                "ms_per_step": round(el / steps * 1e3, 4)}
    forms = {"no_collective": form(el_loc), "allgather": form(el_seq), "allgather_pipelined": form(elapsed)}
    geometry = (b.n_total, b.map.n_local, b.map.pieces, b.map.per, b.map.P)
    roof = roofline_of(b, kernel_ms, name)
    del b
    torch.cuda.empty_cache()
    # the whole-update mode (reduce-scatter + all-gather) on the same clients, where it fits
    try:
        wb = WholeUpdateBench(wl, world, rank)
        el_whole, _ = time_steps(wb.step, steps, 2, barrier, use_dist, False)
        forms["reduce_scatter_whole_updates"] = form(el_whole)
        del wb
    except (RuntimeError, MemoryError) as e:
        forms["reduce_scatter_whole_updates"] = {"skipped": str(e)[:200]}
    torch.cuda.empty_cache()
    out = {"forms": forms,
           "roofline": roof,
           "config": {"params_total": geometry[0], "params_per_gpu": geometry[1],
                      "client_bytes_per_step_total": int(total_client),
                      "block_rows": geometry[2], "block_elements": geometry[3],
                      "gathered_bytes_per_gpu": 4 * geometry[4]}}
    return out


def device_identity(index):
    """host | GPU uuid | PCI address of device `index` of this process."""
    import socket
    p = torch.cuda.get_device_properties(index)
    pci = "%04x:%02x:%02x" % tuple(int(getattr(p, k, 0) or 0)
                                   for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    return "%s|%s|%s" % (socket.gethostname(), getattr(p, "uuid", ""), pci)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # proof that the job really is N ranks on N distinct GPUs: the process group's own size and
    # every rank's device identity, gathered over the group
    ranks, devices = 1, [device_identity(local_rank)]
    if use_dist:
        ranks = dist.get_world_size()
        devices = [None] * ranks
        dist.all_gather_object(devices, device_identity(local_rank))
        if ranks != a.gpus or len(set(devices)) != ranks:
            raise SystemExit(f"--gpus {a.gpus}: the process group has {ranks} ranks on "
                             f"{len(set(devices))} distinct devices: {devices}")

    from appfl_amd import workloads as W
    sharded = world > 1 or a.sharded
    name = a.workload or (SHARDED_WORKLOAD if sharded else DEFAULT_WORKLOAD)
    if name not in W.WORKLOADS:
        raise SystemExit(f"unknown workload {name}; choose from {sorted(W.WORKLOADS)}")
    wl = W.WORKLOADS[name]
    is_async = bench_class(wl) is not SyncBench
    if sharded and is_async:
        raise SystemExit("the sharded benchmark takes a FedAvg-family workload")

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if sharded:
        m = measure_sharded(name, wl, world, rank, a.steps, a.warmup, barrier, use_dist)
        extra = []
        if not a.only:
            # BASELINE configuration 3 (ResNet-50, K = 64, FedAdam / FedYogi on 1/2/4/8 GPUs), sharded
            # the same way; every rank takes part (the gathers are collectives)
            for other in SHARDED_EXTRA_WORKLOADS:
                if other != name:
                    e = measure_sharded(other, W.WORKLOADS[other], world, rank, a.steps, 2, barrier, use_dist)
                    extra.append({"workload": other, "aggregator": W.WORKLOADS[other]["cls"],
                                  "clients": W.WORKLOADS[other]["K"], **e["forms"]["allgather_pipelined"],
                                  "params_total": e["config"]["params_total"],
                                  "params_per_gpu": e["config"]["params_per_gpu"],
                                  "block_rows": e["config"]["block_rows"],
                                  "no_collective": e["forms"]["no_collective"],
                                  "allgather": e["forms"]["allgather"],
                                  "allgather_pipelined": e["forms"]["allgather_pipelined"],
                                  "kernel": e["roofline"]["kernel"],
                                  "kernel_avg_ms": e["roofline"]["kernel_avg_ms"],
                                  "algorithmic_bytes_per_launch": e["roofline"]["algorithmic_bytes_per_launch"],
                                  "frac": e["roofline"]["frac"]})
        if rank == 0:
            out = {
                "metric": METRIC, **m["forms"]["allgather_pipelined"], "n_gpus": world, "ranks": ranks,
                "devices": devices, "steps": a.steps,
                "warmup": a.warmup, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": name, "aggregator": wl["cls"], "clients": wl["K"], **m["config"],
                           "parallelism": f"parameter axis dealt in blocks to {world} GPUs; fused "
                                          "kernel per block row, RCCL all_gather_into_tensor per row, "
                                          "pipelined (value); N=1 counterpart of this workload: "
                                          "strong_scaling_reference of the one-GPU line"},
                **m["forms"],
                "roofline": m["roofline"],
            }
            if extra:
                out["workloads"] = extra
            print(json.dumps(out), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    if a.tune_stash is not None:
        from appfl_amd import ops
        ops.tune(stash=a.tune_stash)
    cold_only = a.cold and a.only and bench_class(wl) is SyncBench
    if cold_only:
        sets = SyncBench.cold_sets_for(wl)
        b = SyncBench(wl, rank, cold_sets=sets)
        elapsed, kernel_ms = time_steps(b.step, a.steps, max(a.warmup, sets), barrier, use_dist)
    else:
        b = bench_class(wl)(wl, rank)
        elapsed, kernel_ms = time_steps(b.step, a.steps, a.warmup, barrier, use_dist)
    if rank == 0:
        out = {
            "metric": METRIC,
            "value": round(world * b.client_bytes * a.steps / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "ranks": ranks,
            "devices": devices,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": name, "aggregator": wl["cls"], "clients": wl["K"],
                       "params_per_gpu": b.result_numel, "client_bytes_per_step_per_gpu": b.client_bytes,
                       "parallelism": "one GPU",
                       "state": f"cold: rotated over {len(b.sets)} independent sets" if cold_only
                                else "re-used every step"},
            "roofline": roofline_of(b, kernel_ms, name),
        }
        del b
        torch.cuda.empty_cache()
        if not cold_only and (name in COLD_WORKLOADS or (bench_class(wl) is SyncBench and not a.only)) \
                and not (a.only and a.tune_stash is not None):
            cold = measure_cold(wl, name, a.steps, a.warmup, barrier)
            out["roofline"].update({k: cold[k] for k in ("frac_cold", "kernel_avg_ms_cold",
                                                           "kernel_min_ms_cold", "achieved_cold",
                                                           "traffic_cold", "state_sets")})
            out["value_cold"], out["ms_per_step_cold"] = cold["value_cold"], cold["ms_per_step_cold"]
        if not a.only:
            extra = []
            for other in EXTRA_WORKLOADS:
                if other != name:
                    extra.append(measure_workload(other, a.steps, a.warmup, barrier))
            out["workloads"] = extra
            target = next((w for w in extra if w["workload"] == NORTH_STAR_WORKLOAD), None)
            if target is not None:
                out["north_star_target"] = {"workload": NORTH_STAR_WORKLOAD, "frac": target["frac"],
                                            "frac_cold": target.get("frac_cold"),
                                            "kernel_avg_ms": target["kernel_avg_ms"],
                                            "kernel_avg_ms_cold": target.get("kernel_avg_ms_cold"),
                                            "value": target["value"], "unit": "GB/s"}
            if not a.no_big:
                try:
                    out["strong_scaling_reference"] = measure_workload(
                        SHARDED_WORKLOAD, max(3, a.steps // 5), 2, barrier)
                except (RuntimeError, MemoryError) as e:      # not enough free HBM on this box
                    out["strong_scaling_reference"] = {"workload": SHARDED_WORKLOAD,
                                                       "skipped": str(e)[:200]}
                    torch.cuda.empty_cache()
            if not a.no_end_to_end and not is_async:
                out["end_to_end"] = end_to_end(wl)
                out["end_to_end_wire"] = end_to_end_wire(wl)
                if name != NORTH_STAR_WORKLOAD:
                    # the same PCIe-inclusive measurement on the north-star target configuration
                    e2e = end_to_end(W.WORKLOADS[NORTH_STAR_WORKLOAD], rounds=3, arrivals=3)
                    e2e["workload"] = NORTH_STAR_WORKLOAD
                    out["end_to_end_north_star"] = e2e
                    # ... and on the same model held in bfloat16 (half the bytes on PCIe and in HBM)
                    e2e = end_to_end(W.WORKLOADS[NORTH_STAR_WORKLOAD + "_bf16"], rounds=3, arrivals=3)
                    e2e["workload"] = NORTH_STAR_WORKLOAD + "_bf16"
                    out["end_to_end_north_star_bf16"] = e2e
        if not a.no_cpu_baseline and bench_class(wl) not in (AdmmBench, MixBench):
            naive, fused = cpu_baseline(wl, a.cpu_seconds)
            out["cpu_baseline"] = naive
            out["cpu_baseline_fused"] = fused
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
