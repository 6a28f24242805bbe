"""State-dict level FedAvg-family aggregation with the parameter axis sharded over the GPUs of
one node (one process per GPU, SPMD).

Every rank calls `aggregate(local_models)` with the SAME client state dicts (e.g. an MPI server
that broadcasts the received bytes, or ranks reading one shared-memory copy).  Rank r packs and
uploads only slice r of every client's weighted entries, runs the fused kernel on its slice
(`ShardedFlatAggregator.round_sliced`, no collective in the arithmetic), and one RCCL
all-gather of the updated slices gives every rank the whole new model, which it returns like
`FedAvgAggregator.aggregate` would.  Integer / non-parameter entries (a few KB) are averaged
redundantly on every rank.  Results are bit-identical to the single-GPU aggregators.

No counterpart in the reference, whose server is a single CPU process
(comm/mpi/mpi_server_communicator.py:56-90); the semantics of a round are those of
algorithm/aggregator/fedavg_aggregator.py:150-404 and the optimizer subclasses.
"""

from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist

from .. import _native as N
from ..layout import MEAN_F32, MEAN_I32, MEAN_I64, STEP, Layout
from ..sharded import ShardedFlatAggregator
from .base_aggregator import BaseAggregator, DeviceStateMixin, cfg_get, cfg_require

_OPTS = {"FedAvg": "none", "FedAvgM": "avgm", "FedAdam": "adam", "FedYogi": "yogi",
         "FedAdagrad": "adagrad"}


class ShardedFedAvgAggregator(BaseAggregator, DeviceStateMixin):
    """`algorithm` selects the server optimizer: FedAvg | FedAvgM | FedAdam | FedYogi |
    FedAdagrad (same configuration fields as the single-GPU classes)."""

    def __init__(self, model: Optional[torch.nn.Module] = None, aggregator_configs: Any = None,
                 logger: Optional[Any] = None, algorithm: str = "FedAvg", group=None,
                 backend=None, device=None):
        self.model = model
        self.logger = logger
        self.aggregator_configs = aggregator_configs if aggregator_configs is not None else {}
        self.client_weights_mode = cfg_get(self.aggregator_configs, "client_weights_mode", "equal")
        self.named_parameters = ({n for n, _ in model.named_parameters()} if model is not None else None)
        self.opt = _OPTS[algorithm]
        self.group = group
        if backend is None:
            from .. import ops as backend
        self.ops = backend
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self._layout: Optional[Layout] = None
        self._flat: Optional[ShardedFlatAggregator] = None
        self._small: Optional[Dict[str, torch.Tensor]] = None    # replicated mean_f32 / i64 / i32 state
        self._init_lock()

    def _hyper(self) -> Dict[str, float]:
        c = self.aggregator_configs
        if self.opt == "none":
            return {}
        if self.opt == "avgm":
            return {"beta1": cfg_require(c, "server_momentum_param_1")}
        return {"lr": cfg_require(c, "server_learning_rate"), "tau": cfg_require(c, "server_adapt_param"),
                "beta1": cfg_require(c, "server_momentum_param_1"),
                "beta2": cfg_require(c, "server_momentum_param_2") if self.opt != "adagrad" else 0.0}

    def _client_weights(self, local_models) -> List[float]:
        sizes = getattr(self, "client_sample_size", None) or {}
        total = sum(sizes[c] for c in local_models if c in sizes)
        return [sizes[c] / total if (self.client_weights_mode == "sample_size" and c in sizes and total > 0)
                else 1.0 / len(local_models) for c in local_models]

    # -- host-side packing of this rank's slice ------------------------------------------------
    def _pack_slice(self, state: Dict[str, torch.Tensor], dst: torch.Tensor, fill: torch.Tensor) -> None:
        """dst[0 : hi-lo] <- elements [lo, hi) of the client's weighted entries (fp32, flat
        order); an entry the client lacks is filled from `fill` (this rank's slice of g), which
        makes its difference zero (fedavg_aggregator.py:360-361)."""
        lo, hi = self._flat.lo, self._flat.hi
        src, off, nb, keep = [], [], [], []
        for e in self._layout.entries:
            if e.cls != STEP:
                continue
            a, b = max(e.offset, lo), min(e.offset + e.numel, hi)
            if a >= b:
                continue
            t = state.get(e.name)
            if t is None:
                dst[a - lo:b - lo].copy_(fill[a - lo:b - lo])
                continue
            t = t.detach().to(torch.float32).contiguous().cpu()
            keep.append(t)
            src.append(t.data_ptr() + 4 * (a - e.offset))
            off.append(4 * (a - lo))
            nb.append(4 * (b - a))
        T = len(src)
        if T:
            rc = N.load().fa_pack_host(C.c_void_p(dst.data_ptr()), (C.c_void_p * T)(*src),
                                       (C.c_size_t * T)(*off), (C.c_size_t * T)(*nb), T, 8)
            N.check(rc, "fa_pack_host")

    def _init_state(self, first_model) -> None:
        if self.model is not None:
            ms = self.model.state_dict()
            init = {n: ms[n] for n in first_model if n in ms}
        else:
            init = dict(first_model)
        self._layout = lay = Layout([(n, tuple(t.shape), t.dtype) for n, t in init.items()],
                                    self.named_parameters)
        hp = self._hyper()
        self._flat = ShardedFlatAggregator(max(lay.count[STEP], 1), self.opt, device=self.device,
                                           group=self.group, backend=self.ops, **hp)
        step_full = torch.zeros(max(lay.count[STEP], 1), dtype=torch.float32)
        small = {"f32": torch.zeros(max(lay.count[MEAN_F32], 1), dtype=torch.float32),
                 "i64": torch.zeros(max(lay.count[MEAN_I64], 1), dtype=torch.int64),
                 "i32": torch.zeros(max(lay.count[MEAN_I32], 1), dtype=torch.int32)}
        for e in lay.entries:
            t = init[e.name].detach().reshape(-1).cpu()
            if e.cls == STEP:
                step_full[e.offset:e.offset + e.numel] = t
            else:
                small[Layout.flat_key(e.cls)][e.offset:e.offset + e.numel] = t
        self._flat.load_global(step_full.to(self.device))
        self._small = {k: v.to(self.device) for k, v in small.items()}

    def get_parameters(self, **kwargs) -> Dict:
        with self._lock:
            if self._layout is None:
                if self.model is None:
                    raise ValueError("Model is not provided to the aggregator.")
                with torch.no_grad():
                    return {k: v.clone().detach() for k, v in self.model.state_dict().items()}
            return self._assemble()

    def _assemble(self, full: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        lay = self._layout
        if full is None:
            full = self._flat.all_gather_global()          # RCCL all-gather of the slices
        host = {"f32": torch.cat([full[:lay.count[STEP]].cpu(),
                                  self._small["f32"][:lay.count[MEAN_F32]].cpu()]),
                "i64": self._small["i64"][:lay.count[MEAN_I64]].cpu(),
                "i32": self._small["i32"][:lay.count[MEAN_I32]].cpu()}
        return lay.views(host)

    def aggregate(self, local_models: Dict, **kwargs) -> Dict:
        with self._lock:
            first = next(iter(local_models.values()))
            if self._layout is None:
                self._init_state(first)
            lay, fl = self._layout, self._flat
            K, n_local = len(local_models), fl.hi - fl.lo
            pin = self.device.type == "cuda"
            rows, full = [], None
            if lay.count[STEP]:
                if n_local:
                    g_host = fl.g[:n_local].cpu()
                    for model in local_models.values():
                        buf = torch.empty(n_local, dtype=torch.float32, pin_memory=pin)
                        self._pack_slice(model, buf, g_host)
                        rows.append(buf.to(self.device, non_blocking=pin))
                ws = None
                if K > 128 and n_local:
                    ws = torch.empty(n_local, dtype=torch.float32, device=self.device)
                # every rank takes part in the gathers, also one whose slice is empty
                full = fl.round_sliced_gather(rows, self._client_weights(local_models), step_ws=ws)
            # replicated plain means of the small entries (fedavg_aggregator.py:285-296)
            for cls, key in ((MEAN_F32, "f32"), (MEAN_I64, "i64"), (MEAN_I32, "i32")):
                cnt = lay.count[cls]
                if not cnt:
                    continue
                xs = []
                for model in local_models.values():
                    parts = [model[e.name].detach().reshape(-1).to(e.dtype) for e in lay.entries if e.cls == cls]
                    xs.append(torch.cat(parts).to(self.device))
                self.ops.mean(xs, self._small[key], n=cnt)
            out = self._assemble(full)
            self._sync_model(self.model, out)
            return out
