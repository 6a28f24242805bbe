"""Device-resident aggregation throughput of the MI355X aggregator hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one pass of the hot path over one batch of synthetic client updates that are already
resident in HBM: for the FedAvg family one round (the fused weighted-step / server-optimizer /
apply kernel over the K staged client rows plus the small plain-mean kernels), for FedAsync /
FedBuff one arrival.  `value` = client-update bytes consumed per second over all ranks.

N > 1 (launched by torch.distributed.run, one rank per GPU): the parameter axis is sharded, every
rank owns a shard of the workload's size and aggregates it with no data-path collective (weak
scaling); `--allgather` additionally times the round with an RCCL all-gather of the updated
shards and reports it under "allgather".

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
STATE_VECTORS = {"none": 2, "avgm": 4, "adam": 6, "yogi": 6, "adagrad": 6}  # g/m/v reads + writes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allgather", action="store_true",
                    help="also time the round followed by an all-gather of the updated shards")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="CPU time budget of each cpu_baseline leg")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time aggregate() from CPU state dicts to a CPU state dict (PCIe included)")
    return ap.parse_args()


def template_state(rows):
    return {name: torch.empty(shape, dtype=dtype) for name, shape, dtype, _ in rows}


class SyncBench:
    """FedAvg-family workload: K synthetic client rows in the device slab, one round per step."""

    def __init__(self, wl, rank):
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
        self.payload = 4 * lay.n_f32 + 8 * lay.count[MEAN_I64] + 4 * lay.count[MEAN_I32]
        g = self.agg._flats["f32"]
        ops.synth_fill(g, 7 + 1000003 * rank, scale=0.05, n=lay.n_f32)
        slab = self.agg._slab
        slab.reserve(self.K)
        slab.slab.zero_()
        for k in range(self.K):
            ops.synth_fill(slab.row_view(k, 0, lay.n_f32, torch.float32), 1000 + k + 1000003 * rank,
                           scale=0.01, base=g, n=lay.n_f32)
        self.weights = [1.0 / self.K] * self.K
        self.opt = self.agg._OPT
        self.kernel = "%s<%s>" % (ops.reduce_kernel_name(self.K, self.n_step, self.opt), self.opt)
        self.alg_bytes = 4 * (self.K + STATE_VECTORS[self.opt]) * self.n_step
        self.client_bytes = self.K * self.payload
        self.result_numel = lay.n_f32

    def step(self, events=None):
        self.agg._launch_round(self.K, self.weights, True, events)

    def step_piece(self, lo, hi):
        """The fused round restricted to elements [lo, hi) of the step segment (lo % 4 == 0)."""
        slab, lay, g = self.agg._slab, self.agg._layout, self.agg._flats["f32"]
        hi = min(hi, self.n_step)
        if hi <= lo:
            return
        rows = [slab.row_ptr(k) + 4 * lo for k in range(self.K)]
        self.agg._ensure_moments()
        m, v = self.agg._m, self.agg._v
        self.ops.reduce_opt(rows, self.weights, g[lo:], None if m is None else m[lo:],
                            None if v is None else v[lo:], self.opt, n=hi - lo, **self.agg._hyper())

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
        g = self.agg._flats["f32"]
        ops.synth_fill(g, 7 + 1000003 * rank, scale=0.05, n=lay.n_f32)
        self.agg._slab.slab.zero_()
        ops.synth_fill(self.agg._slab.row_view(0, 0, lay.n_f32, torch.float32), 1000 + 1000003 * rank,
                       scale=0.01, base=g, n=lay.n_f32)
        self.K = wl["K"]
        self.i = 0
        self.kernel = "buff_accum_kernel" if self.buff else "async_mix_kernel"
        # x, g read + g write; FedBuff: x, g, step read + step write -- the first arrival of a
        # buffer skips the step read and the flushing one adds the g write, so 16 on average
        self.alg_bytes = 16 * self.n_step if self.buff else 12 * self.n_step
        self.client_bytes = 4 * lay.n_f32
        self.result_numel = lay.n_f32

    def step(self, events=None):
        # scripted staleness 0..15, polynomial a = 0.5, alpha = 0.9, 16 clients
        u = self.i % 16
        alpha_t = 0.9 * (u + 1) ** (-0.5) * (1.0 / 16)
        if events:
            events[0].record()
        if self.buff:
            pos = self.i % self.K
            self.agg._launch_arrival(alpha_t, pos == 0, pos == self.K - 1)
        else:
            self.agg._launch_arrival(alpha_t)
        if events:
            events[1].record()
        self.i += 1

    def result(self):
        return self.agg._flats["f32"][:self.result_numel]


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


def end_to_end(wl, rounds=4):
    """aggregate() of the drop-in class as the scheduler calls it: K CPU state dicts in, one CPU
    state dict out (pack + pinned H2D of every client, kernels, D2H).  PCIe-inclusive; reported
    beside `value`, never as `value`."""
    import appfl_amd.aggregator as A
    from appfl_amd import ops, workloads as W

    rows, K = wl["table"](), wl["K"]
    nf = W.numel([r for r in rows if r[2] == torch.float32])
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
            if dtype == torch.float32:
                out[name] = flat[off:off + n].view(shape).clone()
                off += n
            else:
                out[name] = torch.full(shape, 3, dtype=dtype)
        return out

    clients = {k: state(1000 + k) for k in range(K)}
    nbytes = sum(t.numel() * t.element_size() for t in clients[0].values())
    del base, buf
    agg = getattr(A, wl["cls"])(None, dict(W.EXAMPLE_CONFIGS[wl["cls"]]), None)
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
    for _ in range(3):
        for c in ids[:-1]:
            agg.stage(c, clients[c])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agg.stage(ids[-1], clients[ids[-1]])
        agg.aggregate(clients)
        tl.append(time.perf_counter() - t0)
    return {"value": round(K * nbytes / t / 1e9, 2), "unit": "GB/s", "ms_per_round": round(t * 1e3, 2),
            "after_last_arrival_ms": round(sorted(tl)[1] * 1e3, 2),
            "first_round_ms": round(ts[0] * 1e3, 2), "stage_ms": round(t_stage * 1e3, 2),
            "stage_GBps": round(K * nbytes / t_stage / 1e9, 2),
            "what": f"aggregate() of {K} CPU state dicts ({nbytes} B each) -> CPU state dict, "
                    "median of rounds 2.."}


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

    from appfl_amd import workloads as W
    if a.workload not in W.WORKLOADS:
        raise SystemExit(f"unknown workload {a.workload}; choose from {sorted(W.WORKLOADS)}")
    wl = W.WORKLOADS[a.workload]
    is_async = wl["cls"] in ("FedAsyncAggregator", "FedBuffAggregator")
    b = (AsyncBench if is_async else SyncBench)(wl, rank)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        b.step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        b.step(evs[i])
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)

    extra = {}
    if a.allgather and use_dist:
        shard = b.result()
        full = torch.empty(world * shard.numel(), dtype=shard.dtype, device=shard.device)
        for _ in range(2):
            b.step()
            dist.all_gather_into_tensor(full, shard)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            b.step()
            dist.all_gather_into_tensor(full, shard)
        barrier()
        el2 = time.perf_counter() - t0
        t = torch.tensor([el2], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el2 = float(t.item())
        extra["allgather"] = {"value": round(world * b.client_bytes * a.steps / el2 / 1e9, 2),
                              "unit": "GB/s", "ms_per_step": round(el2 / a.steps * 1e3, 4),
                              "gathered_bytes_per_rank": world * shard.numel() * 4}
        if hasattr(b, "step_piece"):
            # the same, pipelined: the gather of piece p overlaps the aggregation of piece p+1
            pieces = 4
            n = shard.numel()
            per = ((n + pieces - 1) // pieces + 3) // 4 * 4
            full = torch.empty(world * n, dtype=shard.dtype, device=shard.device)

            def pipelined():
                works = []
                for lo in range(0, n, per):
                    hi = min(n, lo + per)
                    b.step_piece(lo, hi)
                    dst = [full[r * n + lo:r * n + hi] for r in range(world)]
                    works.append(dist.all_gather(dst, shard[lo:hi], async_op=True))
                for w in works:
                    w.wait()

            for _ in range(2):
                pipelined()
            barrier()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                pipelined()
            barrier()
            el3 = time.perf_counter() - t0
            t = torch.tensor([el3], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el3 = float(t.item())
            extra["allgather_pipelined"] = {
                "value": round(world * b.client_bytes * a.steps / el3 / 1e9, 2), "unit": "GB/s",
                "ms_per_step": round(el3 / a.steps * 1e3, 4), "pieces": pieces}

    if rank == 0:
        achieved = b.alg_bytes / (kernel_avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(a.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": METRIC,
            "value": round(world * b.client_bytes * a.steps / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": a.workload, "aggregator": wl["cls"], "clients": wl["K"],
                       "params_per_gpu": b.result_numel, "client_bytes_per_step_per_gpu": b.client_bytes,
                       "parallelism": f"param-shard x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": b.kernel, "kernel_avg_ms": round(kernel_avg_ms, 5),
                         "kernel_min_ms": round(kernel_ms[0], 5),
                         "algorithmic_bytes_per_launch": int(b.alg_bytes)},
        }
        out.update(extra)
        if world == 1 and a.end_to_end and not is_async:
            out["end_to_end"] = end_to_end(wl)
        if world == 1 and not a.no_cpu_baseline:
            del b
            torch.cuda.empty_cache()
            naive, fused = cpu_baseline(wl, a.cpu_seconds)
            out["cpu_baseline"] = naive
            out["cpu_baseline_fused"] = fused
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
