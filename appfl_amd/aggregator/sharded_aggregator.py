"""State-dict level FedAvg-family aggregation with the parameter axis sharded over the GPUs of
one node (one process per GPU, SPMD).

Every rank calls `aggregate(local_models)` (and `stage(client_id, model)`, if the scheduler
stages on arrival) with the SAME client state dicts, e.g. an MPI server that broadcasts the
received bytes, or ranks reading one shared-memory copy.  Rank r is a `FedAvgAggregator` whose
rows hold only ITS blocks of the weighted entries (`ShardMap`, `WindowedLayout`): the native
packer copies just those windows of every client tensor into the pinned slot, the slab row is
1/R as long, the fused kernel runs on the local blocks with no collective in the arithmetic, and
the RCCL all-gathers of the block rows -- pipelined behind the kernels -- leave the whole new
model on every rank, which downloads it like the single-GPU class.  Integer / non-parameter
entries (a few KB) are averaged redundantly on every rank, and so are float64 entries (rare and
small: they stay whole, with moments of their own, on every rank).  Results are bit-identical to
the single-GPU aggregators.

Collectives happen in `aggregate` only.  `get_parameters` is local: it serves the model every
rank holds after the last gather (or the initial state), so one rank may answer client requests
at any time (base_scheduler.py:67,76 call it outside the scheduler's lock) without the others.

No counterpart in the reference, whose server is a single CPU process
(comm/mpi/mpi_server_communicator.py:56-90); the semantics of a round are those of
algorithm/aggregator/fedavg_aggregator.py:150-404 and the optimizer subclasses.
"""

from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from ..layout import MEAN_F32, MEAN_I32, MEAN_I64, STEP, STEP_F64, Layout, WindowedLayout
from ..sharded import ShardMap, round_gather_pipelined
from ..staging import ClientSlab
from .base_aggregator import cfg_require
from .fedavg_aggregator import FedAvgAggregator, _native

_OPTS = {"FedAvg": "none", "FedAvgM": "avgm", "FedAdam": "adam", "FedYogi": "yogi",
         "FedAdagrad": "adagrad"}


class ShardedFedAvgAggregator(FedAvgAggregator):
    """`algorithm` selects the server optimizer: FedAvg | FedAvgM | FedAdam | FedYogi |
    FedAdagrad (same configuration fields as the single-GPU classes).  `backend` / `device` exist
    for the tests only: they rehearse the multi-rank logic on CPU tensors with gloo and a
    stand-in kernel module; the product leaves both unset (HIP kernels, current GPU)."""

    def __init__(self, model: Optional[torch.nn.Module] = None, aggregator_configs: Any = None,
                 logger: Optional[Any] = None, algorithm: str = "FedAvg", group=None,
                 backend=None, device=None, pieces: Optional[int] = None):
        super().__init__(model, aggregator_configs, logger, device=device)
        self._OPT = _OPTS[algorithm]
        if backend is not None:
            self._ops = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._pieces = pieces
        self._full_layout: Optional[Layout] = None
        self._map: Optional[ShardMap] = None
        self._gathered: Optional[torch.Tensor] = None   # padded full step segment, on every rank

    # -- hyper-parameters and moments of the selected optimizer ---------------------------------
    def _hyper(self) -> Dict[str, float]:
        c = self.aggregator_configs
        if self._OPT == "none":
            return {}
        if self._OPT == "avgm":
            return {"beta1": cfg_require(c, "server_momentum_param_1")}
        return {"lr": cfg_require(c, "server_learning_rate"), "tau": cfg_require(c, "server_adapt_param"),
                "beta1": cfg_require(c, "server_momentum_param_1"),
                "beta2": cfg_require(c, "server_momentum_param_2") if self._OPT != "adagrad" else 0.0}

    def _ensure_moments(self) -> None:
        L = self._map.L
        z = dict(dtype=torch.float32, device=self.device)
        if self._OPT != "none" and self._v is None:
            self._v = torch.zeros(L, **z)
            n64 = self._layout.count[STEP_F64]
            if n64:
                self._v64 = torch.zeros(n64, dtype=torch.float64, device=self.device)
            if self._OPT != "avgm":       # fedadam_aggregator.py:41-50: m = 0, v = tau**2
                tau = cfg_require(self.aggregator_configs, "server_adapt_param")
                self._m = torch.zeros(L, **z)
                self._ops.fill(self._v, tau ** 2)
                if n64:
                    self._m64 = torch.zeros(n64, dtype=torch.float64, device=self.device)
                    self._ops.fill(self._v64, tau ** 2)

    # -- state ------------------------------------------------------------------------------------
    def _names_layout(self) -> Layout:
        return self._full_layout

    def _init_global_state(self, first_model: Dict[str, torch.Tensor]):
        """fedavg_aggregator.py:228-271 on this rank's blocks; the gathered copy of the whole
        weighted segment is built locally (every rank sees the same initial values)."""
        if self.model is not None:
            model_state = self.model.state_dict()
            init = {n: model_state[n] for n in first_model if n in model_state}
        else:
            init = dict(first_model)
        full = Layout([(n, tuple(t.shape), t.dtype) for n, t in init.items()], self.named_parameters)
        smap = ShardMap(full.count[STEP], self.world, self.rank, self._pieces)
        lay = WindowedLayout(full, smap.windows(), smap.L)
        flats = lay.alloc_state(self.device)
        gathered = torch.zeros(smap.P, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for e in lay.entries:
                src = init[e.name].detach().reshape(-1)[e.src_offset:e.src_offset + e.numel]
                off = lay.flat_offset(e)
                flats[Layout.flat_key(e.cls)][off:off + e.numel].copy_(src)
            for e in full.entries:
                if e.cls == STEP:
                    gathered[e.offset:e.offset + e.numel].copy_(init[e.name].detach().reshape(-1))
        self._full_layout, self._map, self._layout = full, smap, lay
        self._flats, self._gathered = flats, gathered
        self._slab = ClientSlab(lay, self.device)

    @property
    def global_state(self) -> Optional[Dict[str, torch.Tensor]]:
        """{name: device tensor} snapshot of the whole global state as this rank holds it after
        the last gather (the gathered weighted segment and the replicated small entries)."""
        if self._full_layout is None:
            return None
        return self._full_layout.views(self._device_state())

    def _device_state(self) -> Dict[str, torch.Tensor]:
        full, L = self._full_layout, self._map.L
        n_step = full.count[STEP]
        f32 = self._gathered[:n_step]
        if full.count[MEAN_F32]:
            f32 = torch.cat([f32, self._flats["f32"][L:L + full.count[MEAN_F32]]])
        return {"f32": f32, "f64": self._flats["f64"], "i64": self._flats["i64"],
                "i32": self._flats["i32"]}

    def _moment_views(self, flat, flat64=None):
        """The moments live sharded (`_m` / `_v` hold this rank's blocks); no per-name views."""
        return {}

    def _download(self, names=None) -> Dict[str, torch.Tensor]:
        """Fresh CPU copy of the whole global state -- local, no collective."""
        full, L = self._full_layout, self._map.L
        n_step, n_mean = full.count[STEP], full.count[MEAN_F32]
        host = {"f32": self._dl.fetch_parts([self._gathered[:n_step],
                                             self._flats["f32"][L:L + n_mean]]),
                "f64": self._dl.fetch(self._flats["f64"], full.n_f64),
                "i64": self._dl.fetch(self._flats["i64"], full.count[MEAN_I64]),
                "i32": self._dl.fetch(self._flats["i32"], full.count[MEAN_I32])}
        out = full.views(host)
        if names is not None:
            return {n: out[n] for n in names if n in out}
        return out

    # -- the round ----------------------------------------------------------------------------------
    _fold_means = False   # the step runs block by block between collectives

    def _launch_step(self, K: int, weights, fused: bool = True, events=None) -> None:
        """Fused kernel on every local block, each followed by the all-gather of its block row.
        Every rank takes part in every gather, also one whose blocks are all empty."""
        smap = self._map
        self._ensure_moments()
        ws = None
        if K > self._ops.kchunk() and smap.n_local:
            if self._step_flat is None:
                self._step_flat = torch.empty(smap.L, dtype=torch.float32, device=self.device)
            ws = self._step_flat
        round_gather_pipelined(self._ops, smap, self._rows(K, 0) if smap.n_local else [], weights,
                               self._flats["f32"], self._m, self._v, self._OPT, self._hyper(),
                               self._gathered, self.group, ws)
        n64 = self._layout.count[STEP_F64]
        if n64:      # whole on every rank
            ws64 = None
            if K > self._ops.kchunk():
                if self._step_flat64 is None:
                    self._step_flat64 = torch.empty(n64, dtype=torch.float64, device=self.device)
                ws64 = self._step_flat64
            self._ops.reduce_opt(self._rows(K, self._layout.seg_byte[STEP_F64]), weights,
                                 self._flats["f64"], self._m64, self._v64, self._OPT, step_ws=ws64,
                                 n=n64, **self._hyper())

    @_native
    def compute_steps(self, local_models):
        raise NotImplementedError("the sharded aggregator runs the fused round only; subclass the "
                                  "single-GPU classes to override compute_steps")

    def aggregate(self, local_models: Dict, **kwargs) -> Dict:
        if "_chunk_idx" in kwargs:
            raise NotImplementedError("streamed (chunked) aggregation is not sharded; use the "
                                      "single-GPU FedAvgAggregator for it")
        if any(isinstance(m, dict) and m.get("type") == "masked_update_flat" for m in local_models.values()):
            raise NotImplementedError("secure-aggregation payloads are not sharded; use the "
                                      "single-GPU FedAvgAggregator for them")
        return super().aggregate(local_models, **kwargs)
