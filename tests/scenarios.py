"""Scripted aggregation scenarios shared by the golden-vector generator, the oracle tests and the
GPU parity tests.  One scenario = one aggregator instance driven through a fixed sequence of
calls on inputs from the counter-based PRNG, so the reference (tests/golden/gen_golden.py), the
CPU oracle and the HIP aggregators all see identical bits.

`run_scenario(factory, sc)` returns an ordered list of (label, {name: np.ndarray}) records.
`factory(class_name, model, cfg_dict)` builds the aggregator under test.
"""

from __future__ import annotations

import itertools
from typing import Callable, Dict, List, Tuple

import numpy as np
import torch

from oracle import synth

# name -> (shape, dtype)
TOY = {
    "conv.weight": ((4, 3, 3, 3), "f32"),
    "bn.weight": ((4,), "f32"),
    "bn.bias": ((4,), "f32"),
    "bn.running_mean": ((4,), "f32"),
    "bn.running_var": ((4,), "f32"),
    "bn.num_batches_tracked": ((), "i64"),
    "fc.weight": ((10, 64), "f32"),
    "fc.bias": ((10,), "f32"),
}
# odd sizes (tails, unaligned entry offsets), an int32 entry and a multi-element int64 entry
RAGGED = {
    "a": ((5,), "f32"),
    "steps": ((3,), "i64"),
    "b": ((7, 3), "f32"),
    "cnt32": ((5,), "i32"),
    "c": ((1,), "f32"),
    "w": ((100003,), "f32"),
}
BIG = {"w": ((1000003,), "f32"), "n": ((), "i64")}
# float64 parameters and buffers next to fp32 ones (state_dict order of Mixed64Model)
MIXED64 = {
    "a.weight": ((6, 4), "f32"),
    "a.bias": ((6,), "f32"),
    "bn.weight": ((5,), "f64"),
    "bn.bias": ((5,), "f64"),
    "bn.running_mean": ((5,), "f64"),
    "bn.running_var": ((5,), "f64"),
    "bn.num_batches_tracked": ((), "i64"),
    "b.weight": ((37, 5), "f64"),
    "b.bias": ((37,), "f64"),
}
# one large parameter with an odd element count and its bias (ADMM scenarios: 128-bit body + tail)
LINEAR = {"l.weight": ((257, 1001), "f32"), "l.bias": ((257,), "f32")}
SPECS = {"toy": TOY, "ragged": RAGGED, "big": BIG, "mixed64": MIXED64, "linear": LINEAR}
# the same entries with every fp32 tensor held in bfloat16 / float16 instead
HALF_KINDS = {"bf16": torch.bfloat16, "f16": torch.float16}
for _kind in HALF_KINDS:
    for _base in ("toy", "ragged"):
        SPECS[f"{_base}_{_kind}"] = {n: (shp, _kind if k == "f32" else k) for n, (shp, k) in SPECS[_base].items()}

_NP = {"f32": np.float32, "f64": np.float64, "i64": np.int64, "i32": np.int32}


def fill64(n: int, seed: int, scale: float = 1.0, base=None) -> np.ndarray:
    """float64 values that are not representable in fp32: two fp32 draws, the second scaled
    down by 2**-25."""
    a = synth.fill(n, seed, scale=scale).astype(np.float64) + \
        synth.fill(n, seed + 7919, scale=scale).astype(np.float64) * 2.0 ** -25
    return a if base is None else base + a


class ToyModel(torch.nn.Module):
    """conv -> bn -> fc with the TOY entry names; parameters come from the PRNG, not torch's."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 4, 3, bias=False)
        self.bn = torch.nn.BatchNorm2d(4)
        self.fc = torch.nn.Linear(64, 10)
        with torch.no_grad():
            for i, (name, t) in enumerate(self.state_dict().items()):
                if t.dtype == torch.int64:
                    t.fill_(3)
                else:
                    t.copy_(torch.from_numpy(
                        synth.fill(t.numel(), 500 + i, scale=0.05).reshape(tuple(t.shape))))
            self.bn.running_var.abs_().add_(1.0)


class Mixed64Model(torch.nn.Module):
    """fp32 linear -> float64 batch-norm -> float64 linear, with the MIXED64 entry names."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 6)
        self.bn = torch.nn.BatchNorm1d(5).double()
        self.b = torch.nn.Linear(5, 37).double()
        with torch.no_grad():
            for i, (name, t) in enumerate(self.state_dict().items()):
                if t.dtype == torch.int64:
                    t.fill_(3)
                elif t.dtype == torch.float64:
                    t.copy_(torch.from_numpy(fill64(t.numel(), 600 + i, scale=0.05).reshape(tuple(t.shape))))
                else:
                    t.copy_(torch.from_numpy(
                        synth.fill(t.numel(), 600 + i, scale=0.05).reshape(tuple(t.shape))))
            self.bn.running_var.abs_().add_(1.0)


class LinearModel(torch.nn.Module):
    """One linear layer with the LINEAR entry names."""

    def __init__(self):
        super().__init__()
        self.l = torch.nn.Linear(1001, 257)
        with torch.no_grad():
            for i, (name, t) in enumerate(self.state_dict().items()):
                t.copy_(torch.from_numpy(synth.fill(t.numel(), 700 + i, scale=0.05).reshape(tuple(t.shape))))


MODELS = {"toy": ToyModel, "mixed64": Mixed64Model, "linear": LinearModel,
          "toy_bf16": lambda: ToyModel().to(torch.bfloat16), "toy_f16": lambda: ToyModel().to(torch.float16)}


def _name_seed(name: str) -> int:
    return sum((i + 1) * ord(c) for i, c in enumerate(name)) % 100000


def make_state(spec: Dict, seed: int, base_seed: int = 7, scale: float = 0.02,
               gradient: bool = False, huge: bool = True) -> Dict[str, torch.Tensor]:
    """One client's state dict: common base (seed 7) + small client-specific noise (`huge`: some
    clients also carry one value of 3e30)."""
    out = {}
    for name, (shape, kind) in spec.items():
        n = int(np.prod(shape)) if shape else 1
        ns = _name_seed(name)
        if kind == "f64":
            if gradient:
                a = fill64(n, seed * 131 + ns, scale=scale)
            else:
                a = fill64(n, seed * 131 + ns, scale=scale, base=fill64(n, base_seed * 131 + ns, scale=0.05))
            if n >= 5:
                a[1] = 0.0 if gradient else fill64(n, base_seed * 131 + ns, scale=0.05)[1]
                a[2] = -0.0
                a[4] = 1e-310                       # a float64 denormal
        elif kind in ("f32", "bf16", "f16"):
            if gradient:
                a = synth.fill(n, seed * 131 + ns, scale=scale)
            else:
                base = synth.fill(n, base_seed * 131 + ns, scale=0.05)
                a = synth.fill(n, seed * 131 + ns, scale=scale, base=base)
            if n >= 8:
                # edge values: exact base (zero difference), a signed zero, a huge value, a denormal
                a[1] = 0.0 if gradient else synth.fill(n, base_seed * 131 + ns, scale=0.05)[1]
                a[2] = -0.0
                if huge and seed % 3 == 0:
                    a[3] = 3.0e30
                a[4] = 1e-41
        else:
            a = synth.ints(seed * 131 + ns, n, -9, 10, _NP[kind])
        out[name] = torch.from_numpy(a.reshape(shape))
        if kind in HALF_KINDS:          # rounded to the 16-bit type (3e30 is inf in float16)
            out[name] = out[name].to(HALF_KINDS[kind])
    return out


def _snap(d) -> Dict[str, np.ndarray]:
    """Records hold numpy arrays; bfloat16 / float16 tensors (numpy has no bfloat16) are held as
    their 16-bit patterns (uint16), every NaN as the type's one quiet NaN."""
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor) and v.dtype in (torch.bfloat16, torch.float16):
            v = v.detach().cpu().clone()
            v[v.isnan()] = float("nan")
            a = v.contiguous().view(torch.int16).numpy().view(np.uint16)
        elif hasattr(v, "bits") and hasattr(v, "kind"):      # oracle.halfarray.HalfArray
            a = v.bits()
        else:
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        out[k] = np.array(a, copy=True)
    return out


HP = {
    "FedAvgAggregator": {},
    "FedAvgMAggregator": {"server_momentum_param_1": 0.9},
    "FedAdamAggregator": {"server_learning_rate": 0.1, "server_adapt_param": 1e-3,
                          "server_momentum_param_1": 0.9, "server_momentum_param_2": 0.99},
    "FedYogiAggregator": {"server_learning_rate": 0.01, "server_adapt_param": 1e-3,
                          "server_momentum_param_1": 0.9, "server_momentum_param_2": 0.99},
    "FedAdagradAggregator": {"server_learning_rate": 0.1, "server_adapt_param": 1e-3,
                             "server_momentum_param_1": 0.9},
}
SYNC_CLASSES = list(HP)


def sync_scenarios() -> List[Dict]:
    out = []
    for cls, with_model, mode, optmem in itertools.product(
            SYNC_CLASSES, (False, True), ("equal", "sample_size"), (True, False)):
        cfg = dict(HP[cls], client_weights_mode=mode, optimize_memory=optmem)
        out.append({
            "id": f"{cls[:-10]}-{'model' if with_model else 'nomodel'}-{mode}-"
                  f"{'opt' if optmem else 'orig'}",
            "kind": "sync", "cls": cls, "spec": "toy", "model": with_model, "cfg": cfg,
            # K per round; client ids rotate so that sample sizes and membership change
            "rounds": [2, 4, 7],
            "sample_sizes": {c: 3 + 8 * c + (c % 3) for c in range(6)} if mode == "sample_size" else {},
        })
    for cls in SYNC_CLASSES:
        out.append({"id": f"{cls[:-10]}-ragged", "kind": "sync", "cls": cls, "spec": "ragged",
                    "model": False, "cfg": dict(HP[cls], client_weights_mode="sample_size"),
                    "rounds": [3, 5, 1], "sample_sizes": {c: 10 + 7 * c for c in range(5)}})
    for cls in ("FedAvgAggregator", "FedAvgMAggregator"):
        out.append({"id": f"{cls[:-10]}-big", "kind": "sync", "cls": cls, "spec": "big",
                    "model": False, "cfg": dict(HP[cls], client_weights_mode="equal"),
                    "rounds": [3, 2], "sample_sizes": {}, "hash_only": True})
    # Yogi sign(0): tau = 2^-10 so v0 = 2^-20; two clients at g + 2^-10 give step = 2^-10 exactly
    out.append({"id": "FedYogi-sign0", "kind": "sync", "cls": "FedYogiAggregator", "spec": "toy",
                "model": False, "cfg": dict(HP["FedYogiAggregator"], server_adapt_param=2.0 ** -10,
                                            client_weights_mode="equal"),
                "rounds": [2, 2], "sample_sizes": {}, "pow2_offsets": True})
    return out


STALENESS = {
    "constant": {"staleness_fn": "constant"},
    "polynomial": {"staleness_fn": "polynomial", "staleness_fn_kwargs": {"a": 0.5}},
    "hinge": {"staleness_fn": "hinge", "staleness_fn_kwargs": {"a": 10, "b": 2}},
}
# (client, download-before-upload?) arrival script: staleness between 0 and 5
ASYNC_SCRIPT = [(0, True), (1, True), (2, False), (0, False), (0, True), (1, False), (2, True),
                (1, True)]


def async_scenarios() -> List[Dict]:
    out = []
    for sname, scfg in STALENESS.items():
        for gb, with_model, optmem in itertools.product((False, True), (False, True), (True, False)):
            cfg = dict(scfg, alpha=0.9, num_clients=3, gradient_based=gb, optimize_memory=optmem)
            out.append({"id": f"FedAsync-{sname}-{'grad' if gb else 'model'}-"
                              f"{'model' if with_model else 'nomodel'}-{'opt' if optmem else 'orig'}",
                        "kind": "async", "cls": "FedAsyncAggregator", "spec": "toy",
                        "model": with_model, "cfg": cfg, "script": ASYNC_SCRIPT, "sample_sizes": {}})
    out.append({"id": "FedAsync-samplesize-ragged", "kind": "async", "cls": "FedAsyncAggregator",
                "spec": "ragged", "model": False,
                "cfg": dict(STALENESS["polynomial"], alpha=0.9, num_clients=3,
                            client_weights_mode="sample_size"),
                "script": ASYNC_SCRIPT, "sample_sizes": {0: 10, 1: 25, 2: 7}})
    for gb, with_model in itertools.product((False, True), (False, True)):
        cfg = dict(STALENESS["polynomial"], alpha=0.9, num_clients=3, K=3, gradient_based=gb)
        out.append({"id": f"FedBuff-{'grad' if gb else 'model'}-{'model' if with_model else 'nomodel'}",
                    "kind": "async", "cls": "FedBuffAggregator", "spec": "toy", "model": with_model,
                    "cfg": cfg, "script": ASYNC_SCRIPT[:7], "sample_sizes": {}})
    out.append({"id": "FedBuff-ragged-K2", "kind": "async", "cls": "FedBuffAggregator",
                "spec": "ragged", "model": False,
                "cfg": dict(STALENESS["hinge"], alpha=0.9, num_clients=3, K=2,
                            client_weights_mode="sample_size"),
                "script": ASYNC_SCRIPT, "sample_sizes": {0: 10, 1: 25, 2: 7}})
    return out


# The reference evaluates sqrt(v) with MKL's vector-math sqrt (torch.sqrt on an MKL build),
# which is within 1 ulp but NOT correctly rounded (measured: 0.65 % of values one ulp low), and
# whose bits depend on the host CPU's approximation instructions.  The oracle and the HIP
# kernels use the IEEE correctly rounded sqrt, so aggregators with a sqrt in their step are
# compared with the reference under a tolerance; everything else is compared bit for bit.
SQRT_CLASSES = ("FedAdamAggregator", "FedYogiAggregator", "FedAdagradAggregator")
REFERENCE_RTOL = 1e-6


def reference_rtol(sc: Dict):
    """None = bit-exact against the reference's vectors; else max |diff| / max |tensor|.
    (bfloat16 / float16 scenarios are bit-exact throughout: a last-place difference of the fp32
    sqrt disappears when the result is rounded to 8 or 11 significant bits.)"""
    if sc.get("half"):
        return None
    return REFERENCE_RTOL if sc["cls"] in SQRT_CLASSES else None


def chunked_scenarios() -> List[Dict]:
    """Streamed aggregation (fedavg_aggregator.py:406-497): every round arrives as chunks of
    name-sorted keys; FedAdam is included because the reference applies no server optimizer on
    this path."""
    out = []
    for cls, with_model in (("FedAvgAggregator", False), ("FedAvgAggregator", True),
                            ("FedAdamAggregator", False)):
        out.append({"id": f"{cls[:-10]}-chunked-{'model' if with_model else 'nomodel'}",
                    "kind": "chunked", "cls": cls, "spec": "toy", "model": with_model,
                    "cfg": dict(HP[cls], client_weights_mode="sample_size"),
                    "rounds": [2, 3], "chunk_bytes": 600,
                    "sample_sizes": {c: 4 + 5 * c for c in range(6)}})
    # whole-model rounds after streamed ones continue from the state the chunks built, in the
    # chunks' (name-sorted) order
    for cls, with_model in (("FedAvgAggregator", False), ("FedAvgMAggregator", True)):
        out.append({"id": f"{cls[:-10]}-chunked-then-whole-{'model' if with_model else 'nomodel'}",
                    "kind": "chunked", "cls": cls, "spec": "toy", "model": with_model,
                    "cfg": dict(HP[cls], client_weights_mode="equal"),
                    "rounds": [2], "chunk_bytes": 600, "whole_after": [3, 2], "sample_sizes": {}})
    return out


def chunk_keys_by_size(spec: Dict, max_bytes: int) -> List[List[str]]:
    """misc/memory_utils.py:317-379: names sorted, greedy fill, oversized tensors alone."""
    item = {"f32": 4, "f64": 8, "i64": 8, "i32": 4, "bf16": 2, "f16": 2}
    chunks, cur, size = [], [], 0
    for key in sorted(spec):
        shape, kind = spec[key]
        nbytes = (int(np.prod(shape)) if shape else 1) * item[kind]
        if nbytes > max_bytes:
            if cur:
                chunks.append(cur)
                cur, size = [], 0
            chunks.append([key])
            continue
        if size + nbytes > max_bytes and cur:
            chunks.append(cur)
            cur, size = [], 0
        cur.append(key)
        size += nbytes
    if cur:
        chunks.append(cur)
    return chunks


# (kind of call, clients, staleness per client): FedCompass mixes single arrivals and groups
COMPASS_SCRIPT = [("single", [0], [0]), ("group", [1, 2], [1, 3]), ("single", [2], [2]),
                  ("group", [0, 1, 2], [0, 4, 1]), ("single", [1], [5])]
QUEUE_SCRIPT = [([0, 1], [0, 1]), ([2], [3]), ([1, 2, 0], [2, 0, 4])]


def mix_scenarios() -> List[Dict]:
    """FedCompass (fedcompass_aggregator.py) and FedQueue (fedqueue_aggregator.py)."""
    out = []
    for gb, with_model, optmem in itertools.product((False, True), (False, True), (True, False)):
        cfg = dict(STALENESS["polynomial"], alpha=0.9, num_clients=3, gradient_based=gb,
                   optimize_memory=optmem)
        out.append({"id": f"FedCompass-{'grad' if gb else 'model'}-"
                          f"{'model' if with_model else 'nomodel'}-{'opt' if optmem else 'orig'}",
                    "kind": "compass", "cls": "FedCompassAggregator", "spec": "toy",
                    "model": with_model, "cfg": cfg, "sample_sizes": {}})
    out.append({"id": "FedCompass-ragged-hinge", "kind": "compass", "cls": "FedCompassAggregator",
                "spec": "ragged", "model": False,
                "cfg": dict(STALENESS["hinge"], alpha=0.5, num_clients=4), "sample_sizes": {}})
    for gb, with_model, sized in itertools.product((False, True), (False, True), (True, False)):
        cfg = {"staleness_fn": "harmonic", "staleness_fn_kwargs": {"beta": 0.5}, "gradient_based": gb}
        out.append({"id": f"FedQueue-{'grad' if gb else 'model'}-{'model' if with_model else 'nomodel'}-"
                          f"{'sized' if sized else 'equal'}",
                    "kind": "queue", "cls": "FedQueueAggregator", "spec": "toy", "model": with_model,
                    "cfg": cfg, "sample_sizes": {0: 12, 1: 30, 2: 7} if sized else {}})
    out.append({"id": "FedQueue-ragged-constant", "kind": "queue", "cls": "FedQueueAggregator",
                "spec": "ragged", "model": False, "cfg": {}, "sample_sizes": {0: 5, 1: 9, 2: 2}})
    return out


def secure_scenarios() -> List[Dict]:
    """Flat secure-aggregation payloads (fedavg_aggregator.py:171-225): every client sends the
    flattened fp32 update plus its example count; the masks themselves are outside this path."""
    out = []
    for mode in ("equal", "sample_size"):
        out.append({"id": f"FedAvg-secureflat-{mode}", "kind": "secure", "cls": "FedAvgAggregator",
                    "spec": "toy", "model": False,
                    "cfg": {"use_secure_agg": True, "client_weights_mode": mode},
                    "rounds": [3, 2, 4], "sample_sizes": {}})
    return out


def getparams_scenarios() -> List[Dict]:
    """Chunked `get_parameters` replies (fedavg_aggregator.py:84-134 with the chunk rule of
    misc/memory_utils.py:317-379): before and after the first round, with and without a model,
    first-request form (`_request_chunked`) and per-chunk form (`_chunk_id`), and a chunk size
    above the model's size (plain reply)."""
    out = []
    for cls, with_model, spec in (("FedAvgAggregator", True, "toy"), ("FedAvgAggregator", False, "toy"),
                                  ("FedAdamAggregator", True, "toy"), ("FedAvgMAggregator", False, "ragged")):
        out.append({"id": f"{cls[:-10]}-getparams-{'model' if with_model else 'nomodel'}-{spec}",
                    "kind": "getparams", "cls": cls, "spec": spec, "model": with_model,
                    "cfg": dict(HP[cls], client_weights_mode="equal"),
                    "chunk_bytes": [600, 10 ** 9], "rounds": [3, 2], "sample_sizes": {}})
    return out


def sched_scenarios() -> List[Dict]:
    """The aggregator driven by a scheduler, as the server agent drives it: per-client
    `schedule` calls with Futures and per-client metadata kwargs (sync_scheduler.py:25-107), the
    initial-model barrier (base_scheduler.py:38-79), a streamed round through the scheduler's
    chunk bookkeeping, and arrivals through async_scheduler.py:20-42.  The golden records come
    from the reference's own schedulers (and its `get_appfl_aggregator` loader)."""
    out = []
    for cls, with_model in (("FedAvgAggregator", True), ("FedAvgAggregator", False),
                            ("FedAdamAggregator", True), ("FedYogiAggregator", False)):
        out.append({"id": f"{cls[:-10]}-sched-sync-{'model' if with_model else 'nomodel'}",
                    "kind": "sched_sync", "cls": cls, "spec": "toy", "model": with_model,
                    "cfg": dict(HP[cls], client_weights_mode="sample_size"), "num_clients": 3,
                    "rounds": 3, "streamed_round": cls == "FedAvgAggregator", "chunk_bytes": 600,
                    "sample_sizes": {"c0": 12, "c1": 30, "c2": 7}})
    for cls, with_model in (("FedAsyncAggregator", True), ("FedAsyncAggregator", False),
                            ("FedBuffAggregator", False)):
        cfg = dict(STALENESS["polynomial"], alpha=0.9, num_clients=3)
        if cls == "FedBuffAggregator":
            cfg["K"] = 2
        out.append({"id": f"{cls[:-10]}-sched-async-{'model' if with_model else 'nomodel'}",
                    "kind": "sched_async", "cls": cls, "spec": "toy", "model": with_model, "cfg": cfg,
                    "script": ASYNC_SCRIPT[:6], "sample_sizes": {}})
    return out


def float64_scenarios() -> List[Dict]:
    """State dicts with float64 parameters and buffers beside fp32 ones: the reference's float
    path takes any float dtype (misc/memory_utils.py:112-131) and PyTorch then keeps the Python
    scalars (weights, alpha_t, hyper-parameters) in double."""
    out = []
    for cls, with_model in itertools.product(SYNC_CLASSES, (False, True)):
        out.append({"id": f"{cls[:-10]}-f64-{'model' if with_model else 'nomodel'}", "kind": "sync",
                    "cls": cls, "spec": "mixed64", "model": with_model,
                    "cfg": dict(HP[cls], client_weights_mode="sample_size"), "rounds": [2, 3, 2],
                    "sample_sizes": {c: 3 + 8 * c + (c % 3) for c in range(6)}})
    for cls, with_model, gb in (("FedAsyncAggregator", False, False), ("FedAsyncAggregator", True, True),
                                ("FedBuffAggregator", True, False), ("FedBuffAggregator", False, True)):
        cfg = dict(STALENESS["polynomial"], alpha=0.9, num_clients=3, gradient_based=gb)
        if cls == "FedBuffAggregator":
            cfg["K"] = 2
        out.append({"id": f"{cls[:-10]}-f64-{'model' if with_model else 'nomodel'}-{'grad' if gb else 'model'}",
                    "kind": "async", "cls": cls, "spec": "mixed64", "model": with_model, "cfg": cfg,
                    "script": ASYNC_SCRIPT[:6], "sample_sizes": {}})
    for gb, with_model in ((False, False), (True, True)):
        out.append({"id": f"FedCompass-f64-{'grad' if gb else 'model'}-{'model' if with_model else 'nomodel'}",
                    "kind": "compass", "cls": "FedCompassAggregator", "spec": "mixed64", "model": with_model,
                    "cfg": dict(STALENESS["polynomial"], alpha=0.9, num_clients=3, gradient_based=gb),
                    "sample_sizes": {}})
        out.append({"id": f"FedQueue-f64-{'grad' if gb else 'model'}-{'model' if with_model else 'nomodel'}",
                    "kind": "queue", "cls": "FedQueueAggregator", "spec": "mixed64", "model": with_model,
                    "cfg": {"staleness_fn": "harmonic", "staleness_fn_kwargs": {"beta": 0.5},
                            "gradient_based": gb},
                    "sample_sizes": {0: 12, 1: 30, 2: 7}})
    out.append({"id": "FedAvg-f64-chunked", "kind": "chunked", "cls": "FedAvgAggregator", "spec": "mixed64",
                "model": False, "cfg": dict(HP["FedAvgAggregator"], client_weights_mode="equal"),
                "rounds": [2, 3], "chunk_bytes": 400, "whole_after": [2], "sample_sizes": {}})
    return out


# ADMM rounds: (client, penalty) in arrival order; penalty None = the client sent no model this
# round (the aggregators then keep its previous primal / dual / penalty).  Round 2 is a subset in
# another order, round 4 changes every penalty.
ADMM_SCRIPT = [
    [(0, 500.0), (1, 250.0), (2, 1000.0)],
    [(0, 500.0), (1, 125.0), (2, 1000.0)],
    [(2, 2000.0), (0, 500.0)],
    [(0, 500.0), (1, None), (2, 1000.0)],
    [(1, 62.5), (2, 1.0 / 3.0), (0, 7.7)],
]
# records of the two residual norms (Python floats from torch.sum over fp32 tensors, whose
# summation order is torch's own): compared under this relative tolerance, everything else in an
# ADMM scenario bit for bit
RESIDUAL_LABEL = ".residuals"
RESIDUAL_RTOL = 1e-5


def admm_scenarios() -> List[Dict]:
    """IIADMM / ICEADMM server updates (iiadmm_aggregator.py:50-165, iceadmm_aggregator.py:50-169):
    clients send {"primal": state dict, "penalty": float} (ICEADMM also "dual")."""
    out = []
    for cls in ("IIADMMAggregator", "ICEADMMAggregator"):
        for spec, optmem in (("toy", True), ("toy", False), ("linear", True)):
            out.append({"id": f"{cls[:-10]}-{spec}-{'opt' if optmem else 'orig'}", "kind": "admm",
                        "cls": cls, "spec": spec, "model": True, "cfg": {"optimize_memory": optmem},
                        "script": ADMM_SCRIPT, "sample_sizes": {}, "huge": not optmem})
    return out


def half_scenarios() -> List[Dict]:
    """bfloat16 / float16 state dicts (integer entries stay int64 / int32): the reference's float
    path takes any float dtype (misc/memory_utils.py:112-131) and PyTorch then rounds every single
    tensor operation to the 16-bit type.  Golden records in tests/golden/golden_half.npz."""
    out = []
    sizes = {c: 3 + 8 * c + (c % 3) for c in range(6)}
    for kind in HALF_KINDS:
        for cls, with_model in itertools.product(SYNC_CLASSES, (False, True)):
            out.append({"id": f"{cls[:-10]}-{kind}-{'model' if with_model else 'nomodel'}", "kind": "sync",
                        "cls": cls, "spec": f"toy_{kind}", "model": with_model,
                        "cfg": dict(HP[cls], client_weights_mode="sample_size"), "rounds": [2, 4, 3],
                        "sample_sizes": sizes})
        for cls in ("FedAvgAggregator", "FedAdamAggregator", "FedYogiAggregator"):
            out.append({"id": f"{cls[:-10]}-{kind}-ragged", "kind": "sync", "cls": cls,
                        "spec": f"ragged_{kind}", "model": False,
                        "cfg": dict(HP[cls], client_weights_mode="equal"), "rounds": [3, 5, 1],
                        "sample_sizes": {}})
        for gb, with_model, optmem in itertools.product((False, True), (False, True), (True, False)):
            cfg = dict(STALENESS["polynomial"], alpha=0.9, num_clients=3, gradient_based=gb,
                       optimize_memory=optmem)
            out.append({"id": f"FedAsync-{kind}-{'grad' if gb else 'model'}-"
                              f"{'model' if with_model else 'nomodel'}-{'opt' if optmem else 'orig'}",
                        "kind": "async", "cls": "FedAsyncAggregator", "spec": f"toy_{kind}",
                        "model": with_model, "cfg": cfg, "script": ASYNC_SCRIPT[:6], "sample_sizes": {}})
        for gb, with_model in ((False, False), (True, True), (False, True)):
            cfg = dict(STALENESS["hinge"], alpha=0.9, num_clients=3, K=2, gradient_based=gb)
            out.append({"id": f"FedBuff-{kind}-{'grad' if gb else 'model'}-{'model' if with_model else 'nomodel'}",
                        "kind": "async", "cls": "FedBuffAggregator",
                        "spec": f"toy_{kind}" if with_model else f"ragged_{kind}", "model": with_model,
                        "cfg": cfg, "script": ASYNC_SCRIPT[:6], "sample_sizes": {}})
        for gb, with_model in ((False, False), (True, True), (False, True)):
            out.append({"id": f"FedCompass-{kind}-{'grad' if gb else 'model'}-{'model' if with_model else 'nomodel'}",
                        "kind": "compass", "cls": "FedCompassAggregator",
                        "spec": f"toy_{kind}" if with_model else f"ragged_{kind}", "model": with_model,
                        "cfg": dict(STALENESS["polynomial"], alpha=0.9, num_clients=3, gradient_based=gb),
                        "sample_sizes": {}})
            out.append({"id": f"FedQueue-{kind}-{'grad' if gb else 'model'}-{'model' if with_model else 'nomodel'}",
                        "kind": "queue", "cls": "FedQueueAggregator",
                        "spec": f"toy_{kind}" if with_model else f"ragged_{kind}", "model": with_model,
                        "cfg": {"staleness_fn": "harmonic", "staleness_fn_kwargs": {"beta": 0.5},
                                "gradient_based": gb},
                        "sample_sizes": {0: 12, 1: 30, 2: 7} if gb else {}})
        out.append({"id": f"FedAvg-{kind}-chunked", "kind": "chunked", "cls": "FedAvgAggregator",
                    "spec": f"toy_{kind}", "model": False,
                    "cfg": dict(HP["FedAvgAggregator"], client_weights_mode="equal"),
                    "rounds": [2, 3], "chunk_bytes": 300, "whole_after": [2], "sample_sizes": {}})
        out.append({"id": f"FedAvgM-{kind}-getparams", "kind": "getparams", "cls": "FedAvgMAggregator",
                    "spec": f"toy_{kind}", "model": True,
                    "cfg": dict(HP["FedAvgMAggregator"], client_weights_mode="equal"),
                    "chunk_bytes": [300, 10 ** 9], "rounds": [3, 2], "sample_sizes": {}})
    for sc in out:
        sc["half"] = True
    return out


def base_scenarios() -> List[Dict]:
    """Everything whose golden records live in tests/golden/golden.npz."""
    return (sync_scenarios() + async_scenarios() + chunked_scenarios() + mix_scenarios()
            + secure_scenarios() + getparams_scenarios() + sched_scenarios() + float64_scenarios()
            + admm_scenarios())


def all_scenarios() -> List[Dict]:
    return base_scenarios() + half_scenarios()


def _meta_record(meta: Dict) -> Dict[str, np.ndarray]:
    """A chunk reply's metadata as arrays (so it is hashed and stored like tensors)."""
    return {"_chunk_idx": np.array(meta["_chunk_idx"], np.int64),
            "_total_chunks": np.array(meta["_total_chunks"], np.int64),
            "_chunk_keys": np.frombuffer("\n".join(meta["_chunk_keys"]).encode(), dtype=np.uint8).copy()}


def _reply_record(reply) -> Dict[str, np.ndarray]:
    """get_parameters reply: a state dict, or (chunk dict, metadata)."""
    if isinstance(reply, tuple):
        d = _snap(reply[0])
        assert list(reply[0]) == list(reply[1]["_chunk_keys"])
        d.update({f"meta.{k}": v for k, v in _meta_record(reply[1]).items()})
        return d
    d = _snap(reply)
    d["meta.plain"] = np.array(1, np.int64)
    return d


def _error_record(e: Exception) -> Dict[str, np.ndarray]:
    return {"message": np.frombuffer(f"{type(e).__name__}: {e}".encode(), dtype=np.uint8).copy()}


def default_schedulers(kind: str, agg, num_clients: int = 0):
    """The package's own scheduling stand-ins, adapted to `schedule(client_id, model, **kw)`."""
    from appfl_amd import harness as H

    class _Sched:
        def __init__(self, h):
            self.h = h
            self.get_parameters = h.get_parameters
            self.get_num_global_epochs = h.get_num_global_epochs

        def schedule(self, client_id, local_model, **kwargs):
            return self.h.global_update(client_id, local_model, **kwargs)

    if kind == "sync":
        return _Sched(H.SerialSyncHarness(agg, num_clients))
    return _Sched(H.SerialAsyncHarness(agg))


def _state_snapshots(agg) -> List[Tuple[str, Dict]]:
    out = []
    for attr in ("m_vector", "v_vector"):
        d = getattr(agg, attr, None)
        if d:
            out.append((attr, _snap(d)))
    return out


def _run_getparams(agg, sc, spec) -> List[Tuple[str, Dict]]:
    rec: List[Tuple[str, Dict]] = []

    def ask(label, **kw):
        try:
            rec.append((label, _reply_record(agg.get_parameters(**kw))))
        except (ValueError, IndexError) as e:
            rec.append((label + ".error", _error_record(e)))

    def sweep(tag):
        for cb in sc["chunk_bytes"]:
            agg.set_model_chunk_size(cb)
            ask(f"{tag}.cb{cb}.first", _request_chunked=True)
            for i in range(3):
                ask(f"{tag}.cb{cb}.chunk{i}", _chunk_id=i)
            ask(f"{tag}.cb{cb}.plain")
            ask(f"{tag}.cb{cb}.first_and_id", _request_chunked=True, _chunk_id=1)

    ask("nochunksize.first", _request_chunked=True)      # no chunk size set yet: plain reply
    sweep("pre")
    for r, K in enumerate(sc["rounds"]):
        local = {c: make_state(spec, seed=8000 + 53 * r + c) for c in range(K)}
        rec.append((f"round{r}", _snap(agg.aggregate(local))))
        sweep(f"post{r}")
    return rec


def _run_sched_sync(agg, sc, spec, model, schedulers) -> List[Tuple[str, Dict]]:
    rec: List[Tuple[str, Dict]] = []
    cids = list(sc["sample_sizes"])
    sched = schedulers("sync", agg, sc["num_clients"])
    # initial model: every client asks; the Futures resolve together on the last request
    futs = []
    for i, cid in enumerate(cids):
        try:
            f = sched.get_parameters(init_model=True)
            futs.append(f)
            rec.append((f"init.{cid}", {"done": np.array([int(x.done()) for x in futs], np.int64)}))
        except ValueError as e:
            rec.append((f"init.{cid}.error", _error_record(e)))
    if futs and all(f.done() for f in futs):
        rec.append(("init.model", _snap(futs[0].result())))
        rec.append(("init.shared", {"same_object": np.array(int(all(f.result() is futs[0].result()
                                                                  for f in futs)), np.int64)}))
    for r in range(sc["rounds"]):
        states = {cid: make_state(spec, seed=9000 + 41 * r + j) for j, cid in enumerate(cids)}
        if sc.get("streamed_round") and r == 1:
            chunks = chunk_keys_by_size(spec, sc["chunk_bytes"])
            for ci, keys in enumerate(chunks):
                futs = [sched.schedule(cid, {k: states[cid][k].clone() for k in keys}, _chunk_idx=ci,
                                       _total_chunks=len(chunks), _chunk_keys=list(keys)) for cid in cids]
                rec.append((f"round{r}.chunk{ci}", _snap(futs[0].result())))
                rec.append((f"round{r}.chunk{ci}.epochs",
                            {"epochs": np.array(sched.get_num_global_epochs(), np.int64)}))
            continue
        futs = []
        for j, cid in enumerate(cids):
            futs.append(sched.schedule(cid, states[cid], val_accuracy=0.1 * j + r, round=r))
            rec.append((f"round{r}.{cid}", {"done": np.array([int(x.done()) for x in futs], np.int64),
                                            "epochs": np.array(sched.get_num_global_epochs(), np.int64)}))
        rec.append((f"round{r}", _snap(futs[0].result())))
        rec.append((f"round{r}.shared", {"same_object": np.array(
            int(all(f.result() is futs[0].result() for f in futs)), np.int64)}))
        rec.append((f"round{r}.get_parameters", _snap(sched.get_parameters(init_model=False))))
    return rec


def _run_sched_async(agg, sc, spec, schedulers) -> List[Tuple[str, Dict]]:
    rec: List[Tuple[str, Dict]] = []
    sched = schedulers("async", agg)
    for i, (cid, download) in enumerate(sc["script"]):
        if download:
            try:
                rec.append((f"arrival{i}.download",
                            _snap(sched.get_parameters(init_model=False, client_id=cid))))
            except ValueError as e:
                rec.append((f"arrival{i}.download.error", _error_record(e)))
        st = make_state(spec, seed=9500 + 31 * i + cid)
        rec.append((f"arrival{i}", _snap(sched.schedule(cid, st, local_steps=5))))
        rec.append((f"arrival{i}.epochs", {"epochs": np.array(sched.get_num_global_epochs(), np.int64)}))
    return rec


def run_scenario(factory: Callable, sc: Dict, schedulers: Callable = default_schedulers
                 ) -> List[Tuple[str, Dict[str, np.ndarray]]]:
    """`schedulers(kind, aggregator, num_clients)` builds the scheduler that drives the
    aggregator in the scheduler-driven scenarios (default: the package's stand-ins; the golden
    generator passes the reference's own)."""
    spec = SPECS[sc["spec"]]
    model = MODELS.get(sc["spec"], ToyModel)() if sc["model"] else None
    agg = factory(sc["cls"], model, dict(sc["cfg"]))
    for cid, n in sc["sample_sizes"].items():
        agg.set_client_sample_size(cid, n)
    if sc["kind"] == "getparams":
        return _run_getparams(agg, sc, spec)
    if sc["kind"] == "sched_sync":
        return _run_sched_sync(agg, sc, spec, model, schedulers)
    if sc["kind"] == "sched_async":
        return _run_sched_async(agg, sc, spec, schedulers)
    rec: List[Tuple[str, Dict]] = []
    if model is not None:
        rec.append(("initial", _snap(agg.get_parameters())))
    if sc["kind"] == "admm":
        named = [n for n, _ in model.named_parameters()]
        for r, clients in enumerate(sc["script"]):
            local = {}
            for cid, penalty in clients:
                if penalty is None:
                    local[cid] = None
                    continue
                huge = sc.get("huge", False)     # its square overflows fp32: both norms become inf
                payload = {"primal": make_state(spec, seed=7000 + 61 * r + cid, huge=huge),
                           "penalty": penalty}
                if sc["cls"] == "ICEADMMAggregator":
                    dual = make_state(spec, seed=7500 + 61 * r + cid, gradient=True, scale=0.5, huge=huge)
                    payload["dual"] = {n: dual[n] for n in named}
                local[cid] = payload
            rec.append((f"round{r}", _snap(agg.aggregate(local))))
            rec.append((f"round{r}{RESIDUAL_LABEL}", {"prim_res": np.array(agg.prim_res, np.float64),
                                                     "dual_res": np.array(agg.dual_res, np.float64)}))
            if sc["cls"] == "IIADMMAggregator":
                rec.append((f"round{r}.dual_states",
                            _snap({f"{cid}/{n}": agg.dual_states[cid][n] for cid in sorted(agg.dual_states)
                                   for n in named})))
        rec.append(("final_get_parameters", _snap(agg.get_parameters())))
    elif sc["kind"] == "sync":
        for r, K in enumerate(sc["rounds"]):
            cids = [(r + j) % max(6, K) for j in range(K)]
            local = {}
            for cid in cids:
                st = make_state(spec, seed=1000 + 97 * r + cid)
                if sc.get("pow2_offsets"):
                    # every client sits exactly 2^-10 above the current global model
                    for name in st:
                        if st[name].dtype != torch.float32:
                            continue
                        if rec and name in rec[-1][1]:
                            ref = torch.from_numpy(rec[-1][1][name])
                        else:
                            ref = torch.from_numpy(synth.fill(
                                st[name].numel(), 7 * 131 + _name_seed(name),
                                scale=0.05).reshape(tuple(st[name].shape)))
                        st[name] = (ref + 2.0 ** -10).to(torch.float32)
                local[cid] = st
            out = agg.aggregate(local, round_tag={c: r for c in cids})
            for lbl, d in _state_snapshots(agg):
                rec.append((f"round{r}.{lbl}", d))
            rec.append((f"round{r}", _snap(out)))
        rec.append(("final_get_parameters", _snap(agg.get_parameters())))
    elif sc["kind"] == "secure":
        names = [n for n, (_, kind) in spec.items() if kind == "f32"]
        for r, K in enumerate(sc["rounds"]):
            local = {}
            for c in range(K):
                st = make_state(spec, seed=6000 + 31 * r + c, gradient=True, scale=0.01)
                flat = torch.cat([st[n].reshape(-1) for n in names])
                local[c] = {"type": "masked_update_flat", "flat": flat,
                            "shapes": [(n, tuple(st[n].shape)) for n in names],
                            "num_examples": 10 + 7 * c + r}
            rec.append((f"round{r}", _snap(agg.aggregate(local))))
        rec.append(("final_get_parameters", _snap(agg.get_parameters())))
    elif sc["kind"] == "compass":
        gb = sc["cfg"].get("gradient_based", False)
        for i, (how, cids, stale) in enumerate(COMPASS_SCRIPT):
            states = {c: make_state(spec, seed=4000 + 31 * i + c, gradient=gb,
                                    scale=0.01 if gb else 0.02) for c in cids}
            if how == "single":
                out = agg.aggregate(client_id=cids[0], local_model=states[cids[0]], staleness=stale[0],
                                    local_steps=7)
            else:
                out = agg.aggregate(local_models=states, staleness=dict(zip(cids, stale)),
                                    local_steps={c: 7 for c in cids})
            rec.append((f"call{i}.{how}", _snap(out)))
        rec.append(("final_get_parameters", _snap(agg.get_parameters())))
    elif sc["kind"] == "queue":
        gb = sc["cfg"].get("gradient_based", False)
        for i, (cids, stale) in enumerate(QUEUE_SCRIPT):
            states = {c: make_state(spec, seed=5000 + 31 * i + c, gradient=gb,
                                    scale=0.01 if gb else 0.02) for c in cids}
            out = agg.aggregate(states, dict(zip(cids, stale)), {c: 3 for c in cids})
            rec.append((f"call{i}", _snap(out)))
        rec.append(("final_get_parameters", _snap(agg.get_parameters())))
    elif sc["kind"] == "chunked":
        chunks = chunk_keys_by_size(spec, sc["chunk_bytes"])
        for r, K in enumerate(sc["rounds"]):
            cids = [(r + j) % 6 for j in range(K)]
            states = {cid: make_state(spec, seed=3000 + 97 * r + cid) for cid in cids}
            for ci, keys in enumerate(chunks):
                local = {cid: {k: states[cid][k].clone() for k in keys} for cid in cids}
                out = agg.aggregate(local, _chunk_idx={c: ci for c in cids},
                                    _total_chunks={c: len(chunks) for c in cids},
                                    _chunk_keys={c: list(keys) for c in cids})
                rec.append((f"round{r}.chunk{ci}", _snap(out)))
        for r, K in enumerate(sc.get("whole_after", [])):
            local = {c: make_state(spec, seed=3500 + 97 * r + c) for c in range(K)}
            rec.append((f"whole{r}", _snap(agg.aggregate(local))))
        rec.append(("final_get_parameters", _snap(agg.get_parameters())))
    else:
        gb = sc["cfg"].get("gradient_based", False)
        for i, (cid, download) in enumerate(sc["script"]):
            if download:
                # without a model the first download raises (and still records the client's step)
                try:
                    rec.append((f"arrival{i}.download", _snap(agg.get_parameters(client_id=cid))))
                except ValueError as e:
                    rec.append((f"arrival{i}.download.ValueError",
                                {"message": np.frombuffer(str(e).encode(), dtype=np.uint8).copy()}))
            st = make_state(spec, seed=2000 + 31 * i + cid, gradient=gb, scale=0.01 if gb else 0.02)
            out = agg.aggregate(cid, st, local_steps=5)
            rec.append((f"arrival{i}", _snap(out)))
        rec.append(("final_get_parameters", _snap(agg.get_parameters())))
    return rec
