"""State-dict level FedAvg-family aggregation with the parameter axis sharded over the GPUs of
one node (one process per GPU, SPMD).

Two ways the clients' state dicts reach the ranks (`mode`):

* ``"sliced"`` (default) -- every rank calls `aggregate(local_models)` (and `stage(client_id,
  model)`, if the scheduler stages on arrival) with the SAME client state dicts, e.g. an MPI
  server that broadcasts the received bytes, or ranks reading one shared-memory copy.  Rank r is
  a `FedAvgAggregator` whose rows hold only ITS blocks of the weighted entries (`ShardMap`,
  `WindowedLayout`): the native packer copies just those windows of every client tensor into the
  pinned slot, the slab row is 1/R as long, the fused kernel runs on the local blocks with no
  collective in the arithmetic, and the RCCL all-gathers of the block rows -- pipelined behind
  the kernels -- leave the whole new model on every rank, which downloads it like the single-GPU
  class.  Bit-identical to the single-GPU aggregators.
* ``"whole"`` -- every rank receives the WHOLE state dicts of SOME clients (client k of the round
  goes to rank k mod R: `aggregate` is called on rank r with the dict of its clients only, in
  their order; a rank may hold none).  Each rank folds its clients into a full-length partial
  step against its copy of the model, an RCCL reduce-scatter (sum) leaves every rank the step of
  its blocks, the server optimizer and the apply run there and an all-gather hands every rank the
  new model.  Host traffic is 1/R of the sliced mode's; the cross-rank sum reorders the fp32
  additions of the weighted step, so weighted fp32 entries agree with the single-GPU result to
  1e-6 of the tensor's scale, not bit for bit.  The small entries (plain means, integers,
  float64) are exchanged whole (an all-gather of the tail of every client row) and computed on
  every rank in the round's client order: those stay bit-identical.

Integer / non-parameter entries (a few KB) are averaged redundantly on every rank, and so are
float64 entries (rare and small: they stay whole, with moments of their own, on every rank).

bfloat16 / float16 models run in the sliced mode: blocks, rows, moments and the gathered model
are 16-bit vectors (half the bytes on the links too) and the arithmetic is the 16-bit kernels'
(appfl_amd.half_ops), bit-identical to the single-GPU aggregators.  The whole mode refuses
them: its cross-rank sum would round a 16-bit partial step once per rank.

Collectives happen in `aggregate`, `compute_steps` and `gather_moments` only -- every rank must
call those, in the same order.  `get_parameters` is local: it serves the model every rank holds
after the last gather (or the initial state), so one rank may answer client requests at any time
(base_scheduler.py:67,76 call it outside the scheduler's lock) without the others.

No counterpart in the reference, whose server is a single CPU process
(comm/mpi/mpi_server_communicator.py:56-90); the semantics of a round are those of
algorithm/aggregator/fedavg_aggregator.py:150-404 and the optimizer subclasses.
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..layout import (MEAN_F32, MEAN_F64, MEAN_I32, MEAN_I64, STEP, STEP_F64, Layout,
                      WindowedLayout)
from ..sharded import (WHOLE_MODE_16BIT, ShardMap, gather_rows, round_gather_pipelined,
                       round_whole_update)
from ..staging import ClientSlab
from .base_aggregator import cfg_require
from .fedavg_aggregator import FedAvgAggregator, _native

_OPTS = {"FedAvg": "none", "FedAvgM": "avgm", "FedAdam": "adam", "FedYogi": "yogi",
         "FedAdagrad": "adagrad"}
_DTYPE_NAMES = {torch.float32: "f32", torch.float64: "f64", torch.int64: "i64", torch.int32: "i32",
                torch.bfloat16: "bf16", torch.float16: "f16"}
_DTYPES = {v: k for k, v in _DTYPE_NAMES.items()}


class ShardedFedAvgAggregator(FedAvgAggregator):
    """`algorithm` selects the server optimizer: FedAvg | FedAvgM | FedAdam | FedYogi |
    FedAdagrad (same configuration fields as the single-GPU classes); `mode`: "sliced" | "whole"
    (module docstring).  `backend` / `device` exist for the tests only: they rehearse the
    multi-rank logic on CPU tensors with gloo and a stand-in kernel module; the product leaves
    both unset (HIP kernels, current GPU)."""

    def __init__(self, model: Optional[torch.nn.Module] = None, aggregator_configs: Any = None,
                 logger: Optional[Any] = None, algorithm: str = "FedAvg", group=None,
                 backend=None, device=None, pieces: Optional[int] = None, mode: str = "sliced"):
        super().__init__(model, aggregator_configs, logger, device=device)
        if mode not in ("sliced", "whole"):
            raise ValueError(f"mode must be 'sliced' or 'whole', got {mode!r}")
        self._OPT = _OPTS[algorithm]
        self.mode = mode
        if backend is not None:
            self._ops = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._pieces = pieces
        self._full_layout: Optional[Layout] = None
        self._map: Optional[ShardMap] = None
        self._gathered: Optional[torch.Tensor] = None   # padded full step segment, on every rank
        self._moments_full: Optional[Tuple[dict, dict]] = None   # last gather_moments(), until the next round
        self._step_full: Optional[torch.Tensor] = None  # compute_steps: the gathered step
        self._partial: Optional[torch.Tensor] = None    # whole mode: this rank's full-length partial step
        self._tail: Optional[torch.Tensor] = None       # whole mode: every client's small entries, gathered
        self._pending: List[tuple] = []                 # whole mode: clients staged before the first round
        self._last_tails: Optional[tuple] = None        # whole mode: (row addresses, K) of the last tail gather

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
        z = dict(dtype=self._full_layout.float_dtype, device=self.device)
        if self._OPT != "none" and self._v is None:
            self._v = torch.zeros(L, **z)
            n64 = self._full_layout.count[STEP_F64]
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
    def _sharded(self) -> bool:
        """False while the instance works as a replicated single-GPU aggregator: before the
        first round, and for secure-aggregation payloads (a layout of their own)."""
        return self._full_layout is not None

    def _names_layout(self) -> Layout:
        return self._full_layout if self._sharded() else self._layout

    def _build_state(self, init: Dict[str, torch.Tensor]) -> None:
        """The global state from {name: initial tensor}: this rank's blocks of the weighted
        segment, the small entries whole, and the gathered copy of the weighted segment (built
        locally: every rank sees the same initial values)."""
        full = Layout([(n, tuple(t.shape), t.dtype) for n, t in init.items()], self.named_parameters)
        if full.half_dtype is not None and self.mode == "whole":
            raise NotImplementedError(WHOLE_MODE_16BIT)      # on every rank: the layout is the same
        smap = ShardMap(full.count[STEP], self.world, self.rank, self._pieces, full.itemsize(STEP))
        lay = WindowedLayout(full, smap.windows(), smap.L)
        flats = lay.alloc_state(self.device)
        gathered = torch.zeros(smap.P, dtype=full.float_dtype, device=self.device)
        with torch.no_grad():
            for e in lay.entries:
                src = init[e.name].detach().reshape(-1)[e.src_offset:e.src_offset + e.numel]
                off = lay.flat_offset(e)
                flats[Layout.flat_key(e.cls)][off:off + e.numel].copy_(src)
            for e in full.entries:
                if e.cls == STEP:
                    gathered[e.offset:e.offset + e.numel].copy_(init[e.name].detach().reshape(-1))
        self._full_layout, self._map = full, smap
        self._flats, self._gathered = flats, gathered
        # rows: this rank's windows of every client (sliced) or whole client updates (whole)
        self._layout = lay if self.mode == "sliced" else full
        self._slab = ClientSlab(self._layout, self.device)
        self._bind_ops()

    def _init_global_state(self, first_model: Dict[str, torch.Tensor]):
        """fedavg_aggregator.py:228-271 on this rank's blocks."""
        if self.model is not None:
            model_state = self.model.state_dict()
            init = {n: model_state[n] for n in first_model if n in model_state}
        else:
            init = dict(first_model)
        self._build_state(init)

    def _adopt_streamed_state(self) -> None:
        init, self._extra_state = self._extra_state, {}
        self._build_state(init)

    @property
    def global_state(self) -> Optional[Dict[str, torch.Tensor]]:
        """{name: device tensor} snapshot of the whole global state as this rank holds it after
        the last gather (the gathered weighted segment and the replicated small entries)."""
        if not self._sharded():
            return FedAvgAggregator.global_state.fget(self)
        out = self._full_layout.views(self._device_state())
        out.update(self._extra_state)
        return out

    def _device_state(self) -> Dict[str, torch.Tensor]:
        full, L = self._full_layout, self._map.L
        n_step = full.count[STEP]
        f32 = self._gathered[:n_step]
        if full.count[MEAN_F32]:
            f32 = torch.cat([f32, self._flats["f32"][L:L + full.count[MEAN_F32]]])
        return {"f32": f32, "f64": self._flats["f64"], "i64": self._flats["i64"],
                "i32": self._flats["i32"]}

    def _state_value(self, key) -> Optional[torch.Tensor]:
        if not self._sharded():
            return super()._state_value(key)
        full, L = self._full_layout, self._map.L
        e = full.by_name.get(key)
        if e is None:
            return self._extra_state.get(key)
        if e.cls == STEP:
            return self._gathered[e.offset:e.offset + e.numel].view(e.shape)
        if e.cls == MEAN_F32:
            return self._flats["f32"][L + e.offset:L + e.offset + e.numel].view(e.shape)
        off = full.flat_offset(e)
        return self._flats[Layout.flat_key(e.cls)][off:off + e.numel].view(e.shape)

    def _state_written(self, key) -> None:
        """A weighted entry was overwritten in the gathered copy (a streamed round): refresh this
        rank's blocks of it."""
        if not self._sharded():
            return
        e = self._full_layout.by_name.get(key)
        if e is None or e.cls != STEP:
            return
        for lo, hi, loc in self._map.windows():
            a, b = max(lo, e.offset), min(hi, e.offset + e.numel)
            if a < b:
                self._flats["f32"][loc + a - lo:loc + b - lo].copy_(self._gathered[a:b])

    # -- moments ----------------------------------------------------------------------------------
    def _name_views(self, flat, flat64) -> Dict[str, torch.Tensor]:
        """{name: view} of a full-length (gathered) fp32 vector and the whole float64 one."""
        src = {STEP: flat, STEP_F64: flat64}
        return {e.name: src[e.cls][e.offset:e.offset + e.numel].view(e.shape)
                for e in self._full_layout.entries if src.get(e.cls) is not None}

    def gather_moments(self) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        """COLLECTIVE (every rank calls it): all-gathers the sharded moments and returns the
        reference's `(m_vector, v_vector)` dicts of device tensors (fedadam_aggregator.py:41-50;
        an empty dict for a moment the optimizer does not keep).  `m_vector` / `v_vector` serve
        the same dicts until the next round."""
        if not self._sharded() or self._v is None:
            return {}, {}
        out = []
        for flat, flat64 in ((self._m, self._m64), (self._v, self._v64)):
            if flat is None:
                out.append({})
                continue
            full = torch.zeros(self._map.P, dtype=flat.dtype, device=self.device)
            gather_rows(self._map, flat, full, self.group)
            out.append(self._name_views(full, flat64))
        self._moments_full = (out[0], out[1])
        return self._moments_full

    def _moments(self, which: int) -> Dict[str, torch.Tensor]:
        """`m_vector` (0) / `v_vector` (1).  The moments of the weighted fp32 entries live sharded:
        a single rank serves them from its own storage, several ranks from the last
        `gather_moments()` as long as no round has run since."""
        if not self._sharded():
            return FedAvgAggregator._moment_views(self, *((self._m, self._m64), (self._v, self._v64))[which])
        flat, flat64 = ((self._m, self._m64), (self._v, self._v64))[which]
        if flat is None and flat64 is None:
            return {}
        if self.world == 1:
            return self._name_views(flat, flat64)     # one rank's blocks are the vector itself
        if self._moments_full is None:
            raise RuntimeError("the moments are sharded over the ranks: call gather_moments() on "
                               "EVERY rank after the round, then read m_vector / v_vector")
        return self._moments_full[which]

    @property
    def m_vector(self) -> Dict[str, torch.Tensor]:
        return self._moments(0)

    @property
    def v_vector(self) -> Dict[str, torch.Tensor]:
        return self._moments(1)

    # -- download ---------------------------------------------------------------------------------
    def _download(self, names=None) -> Dict[str, torch.Tensor]:
        """Fresh CPU copy of the whole global state -- local, no collective."""
        if not self._sharded():
            return super()._download(names)
        full, L = self._full_layout, self._map.L
        n_step, n_mean = full.count[STEP], full.count[MEAN_F32]
        host = {"f32": self._dl.fetch_parts([self._gathered[:n_step],
                                             self._flats["f32"][L:L + n_mean]]),
                "f64": self._dl.fetch(self._flats["f64"], full.n_f64),
                "i64": self._dl.fetch(self._flats["i64"], full.count[MEAN_I64]),
                "i32": self._dl.fetch(self._flats["i32"], full.count[MEAN_I32])}
        out = full.views(host)
        out.update({k: v.cpu() for k, v in self._extra_state.items()})
        if names is not None:
            return {n: out[n] for n in names if n in out}
        return out

    # -- the round ----------------------------------------------------------------------------------
    _fold_means = False   # the step runs block by block between collectives

    def _launch_step(self, K: int, weights, fused: bool = True, events=None) -> None:
        """Sliced mode.  Fused: the kernel on every local block, each followed by the all-gather
        of its block row (every rank takes part in every gather, also one whose blocks are all
        empty).  Un-fused (a `compute_steps` override filled `self.step` with full-length
        tensors): every rank applies the step to its copy of the weighted segment and to its
        blocks -- redundant, no collective."""
        smap, full = self._map, self._full_layout
        self._moments_full = None
        if not fused:
            for e in full.entries:
                if e.name in self.step and e.cls in (STEP, STEP_F64):
                    st = self.step[e.name].contiguous().view(-1)
                    self._ops.apply_step(self._state_value(e.name).view(-1), st)
                    self._state_written(e.name)
            return
        self._ensure_moments()
        ws = None
        if K > self._ops.kchunk() and smap.n_local:
            if self._step_flat is None:
                self._step_flat = torch.empty(smap.L, dtype=full.float_dtype, device=self.device)
            ws = self._step_flat
        round_gather_pipelined(self._ops, smap, self._rows(K, 0) if smap.n_local else [], weights,
                               self._flats["f32"], self._m, self._v, self._OPT, self._hyper(),
                               self._gathered, self.group, ws)
        n64 = full.count[STEP_F64]
        if n64:      # whole on every rank
            self._step_f64(self._rows(K, self._layout.seg_byte[STEP_F64]), weights, K)

    def _step_f64(self, rows, weights, K: int) -> None:
        n64 = self._full_layout.count[STEP_F64]
        ws64 = None
        if K > self._ops.kchunk():
            if self._step_flat64 is None:
                self._step_flat64 = torch.empty(n64, dtype=torch.float64, device=self.device)
            ws64 = self._step_flat64
        self._ops.reduce_opt(rows, weights, self._flats["f64"], self._m64, self._v64, self._OPT,
                             step_ws=ws64, n=n64, **self._hyper())

    @_native
    def compute_steps(self, local_models):
        """COLLECTIVE.  fedavg_aggregator.py:322-368 plus the optimizer epilogue of the selected
        algorithm (e.g. fedadam_aggregator.py:60-99): `self.step` = {name: full-length device
        tensor} of the step the round would apply, the moments updated -- the state the
        reference's `compute_steps` leaves.  Sliced mode: the step of the local blocks, then an
        all-gather; whole mode: partial steps, reduce-scatter, epilogue, all-gather."""
        with self._lock:
            if self.mode == "whole":
                return self._whole_round(local_models, apply=False)
            if self._layout is None:
                self._init_global_state(next(iter(local_models.values())))
            self._stage(local_models)
            self._moments_full = None
            smap, full = self._map, self._full_layout
            K, weights = len(local_models), self._client_weights(local_models)
            self._ensure_moments()
            if self._step_flat is None:
                self._step_flat = torch.empty(smap.L, dtype=full.float_dtype, device=self.device)
            if self._step_full is None:
                self._step_full = torch.zeros(smap.P, dtype=full.float_dtype, device=self.device)
            if smap.n_local:
                self._ops.reduce_step(self._rows(K, 0), weights, self._flats["f32"], self._step_flat,
                                      n=smap.n_local)
                if self._OPT != "none":
                    self._ops.server_opt_step(self._step_flat, self._m, self._v, self._OPT,
                                              n=smap.n_local, **self._hyper())
            gather_rows(smap, self._step_flat, self._step_full, self.group)
            self._steps_f64(self._rows(K, self._layout.seg_byte[STEP_F64]), weights)
            self.step = self._name_views(self._step_full, self._step_flat64
                                         if full.count[STEP_F64] else None)

    def _steps_f64(self, rows, weights) -> None:
        n64 = self._full_layout.count[STEP_F64]
        if not n64:
            return
        if self._step_flat64 is None:
            self._step_flat64 = torch.empty(n64, dtype=torch.float64, device=self.device)
        self._ops.reduce_step(rows, weights, self._flats["f64"], self._step_flat64, n=n64)
        if self._OPT != "none":
            self._ops.server_opt_step(self._step_flat64, self._m64, self._v64, self._OPT, n=n64,
                                      **self._hyper())

    def _fill_source(self):
        """Whole mode: a client lacking a weighted entry gets the global values from the gathered
        copy (its row holds the whole update, in the full layout's offsets)."""
        if self.mode == "whole" and self._sharded():
            return {"f32": self._gathered, "f64": self._flats["f64"]}
        return self._flats

    def stage(self, client_id, local_model) -> None:
        """Upload on arrival.  Whole mode before the first round: the state does not exist yet and
        creating it is a collective, so the client is kept and uploaded by `aggregate`."""
        if self.mode == "whole" and self._full_layout is None:
            self._pending.append((client_id, local_model))
            return
        super().stage(client_id, local_model)

    def aggregate(self, local_models: Dict, **kwargs) -> Dict:
        masked = bool(local_models) and all(isinstance(m, dict) and m.get("type") == "masked_update_flat"
                                            for m in local_models.values())
        if self.mode == "whole":
            if "_chunk_idx" in kwargs or masked:
                raise NotImplementedError(
                    "streamed rounds and secure-aggregation payloads need every rank to see every "
                    "client: use mode='sliced' (they then run replicated on every rank)")
            with self._lock:
                if getattr(type(self).compute_steps, "_appfl_amd_native", False):
                    return self._whole_round(local_models, apply=True)
                # a subclass overrides compute_steps (and may call back into ours): its step is
                # applied on every rank's copy, then the small entries take their means
                self.compute_steps(local_models)
                self._launch_step(0, None, fused=False)
                if self._last_tails is not None:
                    self._launch_means_rows(*self._last_tails)
                out = self._download()
                self.step = {}
                self._sync_model(self.model, out)
                return out
        # streamed rounds (_chunk_idx) and secure-aggregation payloads are small-path work: every
        # rank sees every client, so every rank runs the single-GPU path on its own GPU
        # (replicated, no collective) and keeps the sharded copies of the state in step
        return super().aggregate(local_models, **kwargs)

    # -- whole-update mode ----------------------------------------------------------------------------
    def _tensor_bcast(self, t: torch.Tensor) -> None:
        if self.world > 1:
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)

    def _init_whole(self, local_models: Dict) -> None:
        """First round of the whole mode: the first client of the round sits on rank 0, whose
        initial state (fedavg_aggregator.py:228-271) is broadcast -- the entry list as an object,
        the values as the four flat vectors."""
        spec = [None]
        src_flats = None
        if self.rank == 0:
            if not local_models:
                raise ValueError("whole mode: rank 0 holds the first client of every round")
            first = next(iter(local_models.values()))
            if self.model is not None:
                model_state = self.model.state_dict()
                init = {n: model_state[n] for n in first if n in model_state}
            else:
                init = dict(first)
            for n, t in init.items():
                if t.dtype not in _DTYPE_NAMES:
                    raise NotImplementedError(
                        f"{n}: {t.dtype} entries are not supported by the sharded (multi-GPU) aggregator "
                        "(fp32, bf16, fp16, fp64, int64 and int32 are); there is no CPU fallback")
            spec[0] = [(n, tuple(t.shape), _DTYPE_NAMES[t.dtype]) for n, t in init.items()]
        if self.world > 1:
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast_object_list(spec, src=src, group=self.group)
        names = spec[0]
        full = Layout([(n, s, _DTYPES[d]) for n, s, d in names], self.named_parameters)
        src_flats = full.alloc_state(self.device)
        if self.rank == 0:
            with torch.no_grad():
                for n, t in init.items():
                    full.view(src_flats, n).copy_(t.detach())
        for key in ("f32", "f64", "i64", "i32"):
            self._tensor_bcast(src_flats[key])
        self._build_state(full.views(src_flats))

    def _round_geometry(self, local_models: Dict):
        """(client counts per rank, this rank's clients' weights, every client's weight in the
        round's order).  The round's order deals the ranks' clients round-robin: client j of rank
        r is client j*R + r of the round.  Weights: fedavg_aggregator.py:343-358 over the whole
        round (sample sizes are exchanged as integers, so the quotients are the single-process
        ones)."""
        sizes = getattr(self, "client_sample_size", None) or {}
        K_local = len(local_models)
        known = [int(sizes[c]) if c in sizes else -1 for c in local_models]
        counts = [K_local]
        if self.world > 1:
            t = torch.tensor([K_local], dtype=torch.int64, device=self.device)
            allc = torch.empty(self.world, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(allc, t, group=self.group)
            counts = [int(c) for c in allc.cpu()]
        maxc = max(counts) if counts else 0
        mine = torch.full((max(maxc, 1),), -1, dtype=torch.int64, device=self.device)
        if K_local:
            mine[:K_local] = torch.tensor(known, dtype=torch.int64)
        table = mine.view(1, -1)
        if self.world > 1:
            table = torch.empty((self.world, max(maxc, 1)), dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(table.view(-1), mine, group=self.group)
        table = table.cpu()
        order = [(j, r) for j in range(maxc) for r in range(self.world) if j < counts[r]]
        K = len(order)
        total = sum(int(table[r, j]) for j, r in order if int(table[r, j]) >= 0)
        w_all = []
        for j, r in order:
            n_k = int(table[r, j])
            if self.client_weights_mode == "sample_size" and n_k >= 0 and total > 0:
                w_all.append(n_k / total)
            else:
                w_all.append(1.0 / K)
        w_mine = [w for w, (j, r) in zip(w_all, order) if r == self.rank]
        return counts, order, w_mine, w_all

    def _whole_round(self, local_models: Dict, apply: bool):
        if self._full_layout is None:
            self._init_whole(local_models)
            for cid, model in self._pending:
                FedAvgAggregator.stage(self, cid, model)
            self._pending = []
        full, smap = self._full_layout, self._map
        self._moments_full = None
        K_local = len(local_models)
        self._stage(local_models)              # uploads ordered before the compute stream
        counts, order, w_mine, w_all = self._round_geometry(local_models)
        K = len(order)
        if K == 0:
            raise ValueError("a round needs at least one client on some rank")
        self._ensure_moments()
        if self._partial is None:
            self._partial = torch.zeros(smap.P, dtype=torch.float32, device=self.device)
            self._step_flat = torch.zeros(smap.L, dtype=torch.float32, device=self.device)
        if not apply and self._step_full is None:
            self._step_full = torch.zeros(smap.P, dtype=torch.float32, device=self.device)
        rows = self._rows(K_local, 0)
        if full.count[STEP]:
            round_whole_update(self._ops, smap, rows, w_mine, self._flats["f32"], self._m, self._v,
                               self._OPT, self._hyper(), self._gathered, self._partial,
                               self._step_flat, self.group, apply=apply, step_full=self._step_full)
        # the small entries: every client's tail of the row on every rank, in the round's order
        tail_rows = self._gather_tails(self._row_indices(K_local), counts, order)
        self._last_tails = None if tail_rows is None else (tail_rows, K)
        if tail_rows is not None:
            if apply:
                self._launch_means_rows(tail_rows, K)
                if full.count[STEP_F64]:
                    self._step_f64([p + full.seg_byte[STEP_F64] for p in tail_rows], w_all, K)
            else:
                self._steps_f64([p + full.seg_byte[STEP_F64] for p in tail_rows], w_all)
        if not apply:
            self.step = self._name_views(self._step_full, self._step_flat64
                                         if full.count[STEP_F64] else None)
            return None
        out = self._download()
        self.step = {}
        self._sync_model(self.model, out)
        return out

    def _gather_tails(self, rows, counts, order) -> Optional[List[int]]:
        """`rows`: slab row indices of this rank's clients, in their order.  All-gather of the part of every client row behind the weighted fp32 segment (plain-mean
        fp32, float64, integer entries: KBs per client).  Returns, in the round's client order,
        row addresses such that the usual segment offsets of the full layout apply; None when
        the layout has no such entries."""
        full = self._full_layout
        small = (full.count[MEAN_F32] + full.n_f64 + full.count[MEAN_I64] + full.count[MEAN_I32])
        if not small:
            return None
        t0 = 4 * full.count[STEP] // 256 * 256       # rows and their tails keep their alignment
        tail = full.row_bytes - t0
        maxc = max(counts)
        send = torch.zeros((maxc, tail), dtype=torch.uint8, device=self.device)
        for i, k in enumerate(rows):
            send[i].copy_(self._slab.slab[k, t0:])
        recv = send.view(1, maxc, tail)
        if self.world > 1:
            recv = torch.empty((self.world, maxc, tail), dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        self._tail = recv          # alive until the kernels that read it have been enqueued and the next round replaces it
        return [recv.data_ptr() + (r * maxc + j) * tail - t0 for j, r in order]

    def _launch_means_rows(self, rows: List[int], K: int) -> None:
        """`_launch_means` over explicit row addresses (fedavg_aggregator.py:285-296)."""
        full, L = self._full_layout, self._map.L
        if full.count[MEAN_F32]:
            self._ops.mean([p + full.seg_byte[MEAN_F32] for p in rows], self._flats["f32"][L:],
                           n=full.count[MEAN_F32])
        if full.count[MEAN_F64]:
            self._ops.mean([p + full.seg_byte[MEAN_F64] for p in rows],
                           self._flats["f64"][full.count[STEP_F64]:], n=full.count[MEAN_F64])
        if full.count[MEAN_I64]:
            self._ops.mean([p + full.seg_byte[MEAN_I64] for p in rows], self._flats["i64"],
                           n=full.count[MEAN_I64])
        if full.count[MEAN_I32]:
            self._ops.mean([p + full.seg_byte[MEAN_I32] for p in rows], self._flats["i32"],
                           n=full.count[MEAN_I32])
