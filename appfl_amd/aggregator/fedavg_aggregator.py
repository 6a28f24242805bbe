"""FedAvg-family aggregators with the global state, the server-optimizer moments and the K
staged client updates resident in HBM.

Drop-in for the reference's classes of the same names (constructor, `aggregate`,
`get_parameters`, `set_client_sample_size`, `set_model_chunk_size`, `compute_steps`):

* algorithm/aggregator/fedavg_aggregator.py:17-404
* algorithm/aggregator/fedavgm_aggregator.py:9-75
* algorithm/aggregator/fedadam_aggregator.py:9-133
* algorithm/aggregator/fedyogi_aggregator.py:9-132
* algorithm/aggregator/fedadagrad_aggregator.py:9-118

Inputs are the scheduler's CPU (or device) state dicts, outputs are fresh CPU tensors.  One
`aggregate` call = pack + upload the K clients, ONE fused kernel over the weighted entries
(weighted step, optimizer, apply), small mean kernels for the other entries, one download.

Deliberate differences from the reference, none of which changes a returned value:
`global_state`, `step`, `m_vector`, `v_vector` hold DEVICE tensors (views of flat vectors);
client tensors are never mutated; `optimize_memory` is accepted and ignored (both settings of
the reference compute the same values).
"""

from __future__ import annotations

from typing import Any, Dict, Optional, OrderedDict, Union

import torch

from .. import half_ops, ops
from ..layout import (MEAN_F32, MEAN_F64, MEAN_I32, MEAN_I64, STEP, STEP_F64, Layout,
                      split_names_by_size)
from ..staging import ClientSlab, HostDownloader
from .base_aggregator import (BaseAggregator, DeviceStateMixin, cfg_get, cfg_require,
                              resolve_device)


class _ChunkState:
    """Working set of one chunk of a streamed aggregation (its own layout, vectors, slab); the
    values live in the aggregator's global state between calls."""

    def __init__(self, layout, flats, slab):
        self.layout, self.flats, self.slab = layout, flats, slab


def _download_flats(lay: Layout, flats: Dict[str, torch.Tensor],
                    dl: Optional[HostDownloader] = None) -> Dict[str, torch.Tensor]:
    """Fresh CPU copy of flat state vectors: one D2H copy per dtype, entries are views of it."""
    dl = dl or HostDownloader()
    host = {}
    for key, n in lay.flat_sizes().items():
        host[key] = dl.fetch(flats[key], n, wait=False)
    out = lay.views(host)          # while the transfers are in flight
    dl.wait(flats["f32"].device)
    return out


def _native(fn):
    """Marks a compute_steps implementation of this package (a user override lacks the mark and
    switches `aggregate` to the un-fused compute_steps -> apply sequence)."""
    fn._appfl_amd_native = True
    return fn


class FedAvgAggregator(BaseAggregator, DeviceStateMixin):
    """
    :param `model`: optional model; with it, entries that are not `named_parameters` (batch-norm
        statistics) take the plain mean and `get_parameters` works before the first round.
    :param `aggregator_configs`: `aggregator_kwargs` of the server configuration.
    :param `logger`: optional logger (`.info` / `.warning`).
    """

    _OPT = "none"

    def __init__(self, model: Optional[torch.nn.Module] = None, aggregator_configs: Any = None,
                 logger: Optional[Any] = None, *, device=None):
        self.model = model
        self.logger = logger
        self.aggregator_configs = aggregator_configs if aggregator_configs is not None else {}
        self.client_weights_mode = cfg_get(self.aggregator_configs, "client_weights_mode", "equal")
        self.optimize_memory = cfg_get(self.aggregator_configs, "optimize_memory", True)
        if self.model is not None:
            self.named_parameters = {name for name, _ in self.model.named_parameters()}
        else:
            self.named_parameters = None
        self.step: Dict[str, torch.Tensor] = {}
        self.model_chunk_size = None
        self.device = (torch.device(device) if device is not None
                       else resolve_device(self.aggregator_configs))
        self._ops = ops          # the flat kernels (libappfl_amd.so through ctypes)
        self._plain_ops = ops
        self._layout: Optional[Layout] = None
        self._flats: Optional[Dict[str, torch.Tensor]] = None      # global state
        self._slab: Optional[ClientSlab] = None
        self._m: Optional[torch.Tensor] = None          # moments / step of the fp32 weighted entries
        self._v: Optional[torch.Tensor] = None
        self._step_flat: Optional[torch.Tensor] = None
        self._m64: Optional[torch.Tensor] = None        # ... and of the float64 ones
        self._v64: Optional[torch.Tensor] = None
        self._step_flat64: Optional[torch.Tensor] = None
        self._staged_for = None
        self._prestaged: Dict[Any, tuple] = {}   # client_id -> (row, model) staged on arrival
        self._next_row = 0                       # slab rows handed out since the last round (never reused within one)
        self._init_client = None                 # the client whose staged tensors initialised the state (until the first round)
        self._restaged = False                   # a client was staged twice: rows are no longer in arrival order
        self._row_of: list = []                  # slab row of the k-th client of this round
        self._chunks: Dict[tuple, "_ChunkState"] = {}    # streamed aggregation: working sets per key set
        # entries of the global state that streamed rounds created outside the flat layout
        # (name -> device tensor, in the order the reference's dict would hold them)
        self._extra_state: Dict[str, torch.Tensor] = {}
        self._dl = HostDownloader()
        self._uploads_pending = False     # `_stage` left the compute stream un-ordered against the uploads
        # clients staged on arrival whose terms of the weighted step are folded already, under
        # weights guessed from the configuration: {"w": [...], "event": after the last fold}
        self._folded: Optional[dict] = None
        self._folded_round: Optional[dict] = None     # ... as `_stage` found it for the round being aggregated
        self._fold_flat: Optional[torch.Tensor] = None   # the partial step of the clients folded on arrival
        self._round_plans: Dict[tuple, Any] = {}  # kept arguments of recent fused rounds (ops.RoundPlan)
        self._last_round: Optional[tuple] = None  # the plan of the previous fused round and what it was built from
        self._init_lock()

    # -- attributes the reference exposes -------------------------------------------------------
    @property
    def global_state(self) -> Optional[Dict[str, torch.Tensor]]:
        """{name: device tensor} views of the flat global state (None before the first round)."""
        if self._layout is None:
            return dict(self._extra_state) if self._extra_state else None
        out = self._layout.views(self._flats)
        out.update(self._extra_state)
        return out

    def _bind_ops(self) -> None:
        """The operations for the current layout: the plain ones for fp32 / fp64 entries, the
        16-bit ones (appfl_amd.half_ops) when the float entries are bfloat16 / float16."""
        self._ops = half_ops.for_layout(self._layout, self._unrounded_ops())

    def _unrounded_ops(self):
        """`_ops` as it was before a 16-bit layout put the 16-bit operations in its place."""
        if getattr(self._ops, "half", None) is not None:     # HalfOps (or a test's stand-in for it)
            return self._plain_ops
        self._plain_ops = self._ops
        return self._ops

    def _moment_views(self, flat, flat64=None) -> Dict[str, torch.Tensor]:
        if flat is None and flat64 is None:
            return {}
        src = {STEP: flat, STEP_F64: flat64}
        return {e.name: src[e.cls][e.offset:e.offset + e.numel].view(e.shape)
                for e in self._layout.entries if src.get(e.cls) is not None}

    @property
    def m_vector(self) -> Dict[str, torch.Tensor]:
        return self._moment_views(self._m, self._m64)

    @property
    def v_vector(self) -> Dict[str, torch.Tensor]:
        return self._moment_views(self._v, self._v64)

    def set_model_chunk_size(self, chunk_size: int):
        """Chunk size in bytes for chunked `get_parameters` (set by the gRPC communicator)."""
        self.model_chunk_size = chunk_size

    # -- get_parameters: fedavg_aggregator.py:67-148 ---------------------------------------------
    def get_parameters(self, **kwargs) -> Union[Dict, tuple]:
        with self._lock:
            request_chunked = kwargs.get("_request_chunked", False)
            requested_chunk_id = kwargs.get("_chunk_id", None)
            if self.model_chunk_size is not None and (request_chunked or requested_chunk_id is not None):
                if self._layout is None and not self._extra_state:
                    if self.model is None:
                        raise ValueError("Model is not provided to the aggregator.")
                    source = self.model.state_dict()
                else:
                    source = None
                if source is not None:
                    sizes = {k: v.numel() * v.element_size() for k, v in source.items()}
                else:
                    names_layout = self._names_layout()
                    sizes = {e.name: e.numel * torch.empty((), dtype=e.dtype).element_size()
                             for e in (names_layout.entries if names_layout is not None else [])}
                    sizes.update({k: v.numel() * v.element_size() for k, v in self._extra_state.items()})
                if sum(sizes.values()) > self.model_chunk_size:
                    chunks = split_names_by_size(sizes, self.model_chunk_size)
                    chunk_idx = requested_chunk_id if requested_chunk_id is not None else 0
                    if requested_chunk_id is None and self.logger:
                        self.logger.info(
                            f"GetGlobalModel: Model size {sum(sizes.values()) / 2**20:.2f} MB will "
                            f"be sent in {len(chunks)} chunks")
                    keys = chunks[chunk_idx]
                    if source is not None:
                        with torch.no_grad():
                            result = {k: source[k].clone().detach() for k in keys}
                    else:
                        result = self._download(keys)
                    meta = {"_chunk_idx": chunk_idx, "_total_chunks": len(chunks),
                            "_chunk_keys": keys}
                    return (result, meta)
            if self._layout is None and self._extra_state:
                return self._download()
            if self._layout is None:
                if self.model is None:
                    raise ValueError("Model is not provided to the aggregator.")
                with torch.no_grad():
                    return {k: v.clone().detach() for k, v in self.model.state_dict().items()}
            return self._download()

    def _names_layout(self) -> Layout:
        """The layout that lists the global state's entries by name."""
        return self._layout

    # -- state ------------------------------------------------------------------------------------
    def _init_global_state(self, first_model: Dict[str, torch.Tensor]):
        """fedavg_aggregator.py:228-271: the model's entries for the names the first client sent,
        or (no model) the first client's tensors."""
        if self.model is not None:
            model_state = self.model.state_dict()
            init = {n: model_state[n] for n in first_model if n in model_state}
        else:
            init = dict(first_model)
        self._layout = Layout([(n, tuple(t.shape), t.dtype) for n, t in init.items()],
                              self.named_parameters)
        self._flats = self._layout.alloc_state(self.device)
        with torch.no_grad():
            for n, t in init.items():
                self._layout.view(self._flats, n).copy_(t.detach())
        self._slab = ClientSlab(self._layout, self.device)
        self._round_plans.clear()        # they point into the state vectors just replaced
        self._last_round = None
        self._bind_ops()

    def _adopt_streamed_state(self) -> None:
        """A whole-model round after streamed rounds only: the global state is what the chunks
        built (fedavg_aggregator.py:228 finds `global_state` set), in their order."""
        init, self._extra_state = self._extra_state, {}
        self._layout = Layout([(n, tuple(t.shape), t.dtype) for n, t in init.items()],
                              self.named_parameters)
        self._flats = self._layout.alloc_state(self.device)
        for n, t in init.items():
            self._layout.view(self._flats, n).copy_(t)
        self._slab = ClientSlab(self._layout, self.device)
        self._round_plans.clear()
        self._last_round = None
        self._bind_ops()

    def _download(self, names=None) -> Dict[str, torch.Tensor]:
        """Fresh CPU copy of the global state: one D2H copy per dtype, entries are views of it."""
        lay = self._layout
        if names is not None:
            return {n: (lay.view(self._flats, n) if lay is not None and n in lay.by_name
                        else self._extra_state[n]).cpu()
                    for n in names if (lay is not None and n in lay.by_name) or n in self._extra_state}
        out = _download_flats(lay, self._flats, self._dl) if lay is not None else {}
        out.update({k: v.cpu() for k, v in self._extra_state.items()})
        return out

    def _client_weights(self, local_models) -> list:
        """fedavg_aggregator.py:343-358."""
        sizes = getattr(self, "client_sample_size", None) or {}
        total = sum(sizes[cid] for cid in local_models if cid in sizes)
        out = []
        for cid in local_models:
            if self.client_weights_mode == "sample_size" and cid in sizes and total > 0:
                out.append(sizes[cid] / total)
            else:
                out.append(1.0 / len(local_models))
        return out

    # bytes per pack / transfer window of a client staged on arrival (measured, time from the last
    # arrival to the result for ResNet-18 x 100 / ResNet-50 FedAdam x 64: 4 MiB 5.1 / 9.1 ms,
    # 8 MiB 3.2 / 6.0, 16 MiB 2.6 / 4.8, 32 MiB 2.5 / 5.0; without the pipeline 2.75-3.0 / 5.1-5.4)
    STAGE_WINDOW = 32 << 20
    PIPELINE_MIN_ELEMENTS = 4 << 20   # rounds of at least this many weighted elements may be pipelined

    def stage(self, client_id, local_model) -> None:
        """Optional hook for the scheduler: pack and upload ONE client's update the moment it
        arrives (sync_scheduler.py:46 is where the reference merely stores it), so the PCIe
        transfer overlaps the wait for the remaining clients.  `aggregate` reuses the staged
        row when it later receives the same object for this client; not calling `stage` at all
        is equally valid.  The first staged client of the first round initialises the global
        state exactly as the first client of `aggregate` would (:228-271)."""
        with self._lock:
            if self._layout is None and self._extra_state:
                self._adopt_streamed_state()        # streamed rounds built the state already
            elif self._layout is None:
                self._init_global_state(local_model)
                self._init_client = client_id
            elif client_id == self._init_client and self.model is None:
                # the client whose tensors became the initial global state (:260-271) sends again
                # before the first round: the scheduler's dict now holds this message in that
                # client's place, so the state starts from it
                if self._layout.matches(local_model):
                    with torch.no_grad():
                        for n, t in local_model.items():
                            self._layout.view(self._flats, n).copy_(t.detach())
                else:            # other entries than before: everything staged so far is void
                    self._init_global_state(local_model)
                    self._prestaged, self._next_row = {}, 0
            # a client staged again takes a NEW row (its old one may still be read by a fold
            # enqueued earlier); rows are counted, not derived from the dict, which stops growing then
            row = self._next_row
            self._next_row = row + 1
            hint = cfg_get(self.aggregator_configs, "num_clients", 0) or 0
            self._slab.reserve(max(row + 1, int(hint)), preserve=row)
            if row == 0:
                self._slab.begin_round()
            # the windows' events are kept: `aggregate` can then start on the head of this row
            # while its tail still travels, and send results back meanwhile (`_round_pipelined`)
            self._slab.upload(row, local_model, fill_missing_from=self._fill_source(), window=self.STAGE_WINDOW)
            if self._slab.has_device_sources:
                # the caller may overwrite its device tensors as soon as this returns
                self._slab.wait_uploads()
            if client_id in self._prestaged:
                self._restaged = True
                self._folded = {"w": None}
            self._prestaged[client_id] = (row, local_model)
            self._fold_on_arrival(row, int(hint))

    FOLD_ON_ARRIVAL = True

    def _guessed_weight(self, client_id, num_clients: int) -> Optional[float]:
        """The weight `_client_weights` will give this client IF the round turns out to hold
        exactly the configured `num_clients` clients (all with a known sample size, in
        `sample_size` mode); `_round_pipelined` checks the guess against the real weights."""
        if self.client_weights_mode == "sample_size":
            sizes = getattr(self, "client_sample_size", None) or {}
            if len(sizes) != num_clients or client_id not in sizes:
                return None
            total = sum(sizes.values())
            return sizes[client_id] / total if total > 0 else None
        return 1.0 / num_clients

    def _fold_on_arrival(self, row: int, num_clients: int) -> None:
        """Fold the weighted-step term of the client staged BEFORE this one (its row has landed
        long ago; the newest row is still travelling) into the partial step, so that the round
        after the last arrival only has that last client's term left to add.  Same loop over the
        clients in the same order as the fused kernel, cut between two clients: same bits."""
        lay = self._layout
        n_step = lay.count[STEP]
        if not (self.FOLD_ON_ARRIVAL and not self._restaged and self._fold_means
                and self.device.type == "cuda" and num_clients >= 2 and row >= 1 and n_step >= self.PIPELINE_MIN_ELEMENTS
                and not lay.n_f64 and hasattr(self._ops, "round_sync")
                and getattr(type(self).compute_steps, "_appfl_amd_native", False)):
            return
        st = self._folded
        if row == 1:
            st = self._folded = {"w": [], "event": None}
        if st is None or st["w"] is None or len(st["w"]) != row - 1 or row > num_clients - 1:
            return          # out of step with the arrivals (or a round larger than configured): no guess
        prev = next((cid for cid, (r, _) in self._prestaged.items() if r == row - 1), None)
        w = None if prev is None else self._guessed_weight(prev, num_clients)
        if w is None:
            st["w"] = None
            return
        if self._fold_flat is None:
            self._fold_flat = torch.empty(n_step, dtype=lay.float_dtype, device=self.device)
        compute = torch.cuda.current_stream(self.device)
        compute.wait_event(self._slab.row_event[row - 1])
        st["w"].append(w)
        ptrs = [self._slab.row_ptr(r) for r in range(row)]
        self._ops.reduce_step(ptrs, st["w"], self._flats["f32"], self._fold_flat, n=n_step,
                              k_done=row - 1)
        ev = torch.cuda.Event()
        ev.record(compute)
        st["event"] = ev

    def stage_bytes(self, client_id, model_bytes) -> Dict:
        """`stage` for an update still in wire form (`torch.save` bytes, comm/mpi/serializer.py:46-50,
        comm/grpc/utils.py:20-24): the tensors are read in place from `model_bytes` and packed
        straight into the pinned staging row, skipping the per-tensor CPU copies `torch.load`
        makes (agent/server.py:505-513).  Returns the state dict of views to hand to the
        scheduler / `aggregate` in place of the deserialised model."""
        import io

        from .. import wire

        try:
            local_model = wire.bytes_to_model(model_bytes)
        except wire.WireFormatError:
            # not a plain torch.save archive of tensors (legacy format, other containers): let
            # torch read it, still without executing anything from the payload
            local_model = torch.load(io.BytesIO(model_bytes), map_location="cpu", weights_only=True)
        self.stage(client_id, local_model)
        return local_model

    def _stage(self, local_models, defer_wait: bool = False) -> None:
        """Pack and upload every client's state dict into the device slab (client order =
        dict order = arrival order, which is the reduction order); clients already uploaded by
        `stage` keep their rows.  Inside `aggregate` the round's dict is staged once: a
        `compute_steps` override that calls back into this class finds it done.  A stand-alone
        `compute_steps` call always stages afresh (the schedulers refill and reuse one dict
        object across rounds, sync_scheduler.py:46,80, so identity alone proves nothing).
        `defer_wait`: leave the compute stream un-ordered against the uploads; the caller's
        `_launch_round` then starts on the rows that have landed while the last ones travel."""
        if self._staged_for is not None and self._staged_for is local_models:
            return
        if self._init_client is not None and self.model is None and local_models:
            # before the first round, no model: `stage` started the global state from the first
            # client STAGED; the reference starts it from the round's first client (:260-271).
            # Under the schedulers the two are one object; if they are not, start over from the
            # round's (rows staged so far may hold fills from the void state, so they go too)
            first_id = next(iter(local_models))
            staged = self._prestaged.get(first_id)
            if first_id != self._init_client or staged is None or staged[1] is not local_models[first_id]:
                self._init_global_state(local_models[first_id])
                self._prestaged, self._next_row, self._folded, self._restaged = {}, 0, None, False
        K = len(local_models)
        taken = self._next_row
        reuse = {cid: row for cid, (row, model) in self._prestaged.items()
                 if cid in local_models and local_models[cid] is model}
        fresh = K - len(reuse)
        self._slab.reserve(taken + fresh, preserve=taken)
        if taken == 0:
            self._slab.begin_round()
        self._row_of = []
        nxt = taken
        for cid, model in local_models.items():
            if cid in reuse:
                self._row_of.append(reuse[cid])
                continue
            self._slab.upload(nxt, model, fill_missing_from=self._fill_source())
            self._row_of.append(nxt)
            nxt += 1
        self._uploads_pending = defer_wait
        if not defer_wait:
            self._slab.end_round()
        self._prestaged = {}
        self._next_row, self._restaged, self._init_client = 0, False, None
        self._folded_round, self._folded = self._folded, None

    def _hyper(self) -> Dict[str, float]:
        return {}

    def _ensure_moments(self) -> None:
        """Subclasses allocate m / v on the first round (e.g. fedadam_aggregator.py:41-50)."""

    def _row_indices(self, K: int):
        """Slab rows of this round's K clients, in client order."""
        return self._row_of if len(self._row_of) == K else range(K)

    def _rows(self, K: int, byte_offset: int):
        return [self._slab.row_ptr(r) + byte_offset for r in self._row_indices(K)]

    def _fill_source(self):
        """Where a client's missing weighted entries are filled from (the global flat vectors)."""
        return self._flats

    # -- compute_steps: fedavg_aggregator.py:322-368 ------------------------------------------------
    @_native
    def compute_steps(self, local_models: Dict[Union[str, int], Union[Dict, OrderedDict]]):
        """Averaged pseudo-gradient of the weighted entries into `self.step` (device views)."""
        with self._lock:
            self._stage(local_models)
            lay = self._layout
            n, n64 = lay.count[STEP], lay.count[STEP_F64]
            weights = self._client_weights(local_models)
            if self._step_flat is None:
                self._step_flat = torch.empty(max(n, 1), dtype=lay.float_dtype, device=self.device)
            if n:
                self._ops.reduce_step(self._rows(len(local_models), 0), weights,
                                      self._flats["f32"], self._step_flat, n=n)
            if n64:
                if self._step_flat64 is None:
                    self._step_flat64 = torch.empty(n64, dtype=torch.float64, device=self.device)
                self._ops.reduce_step(self._rows(len(local_models), lay.seg_byte[STEP_F64]), weights,
                                      self._flats["f64"], self._step_flat64, n=n64)
            src = {STEP: self._step_flat, STEP_F64: self._step_flat64}
            self.step = {e.name: src[e.cls][e.offset:e.offset + e.numel].view(e.shape)
                         for e in lay.entries if e.cls in src}

    # whether the plain-mean segments of a round may ride on the fused launch (subclasses whose
    # `_launch_step` does something else than one fused call over the rows switch it off)
    _fold_means = True

    def _launch_round(self, K: int, weights, fused: bool = True, events=None) -> None:
        """Enqueue one round's kernels over the K staged rows: the fused weighted-step /
        optimizer / apply kernel (or, un-fused, the apply of a step computed by compute_steps),
        then the plain means of the entries that bypass the step.  `events` = (start, end)
        torch.cuda.Event pair recorded around the fused kernel (benchmark timing)."""
        lay = self._layout
        pending, self._uploads_pending = self._uploads_pending, False
        if (fused and self._fold_means and lay.count[STEP] and not lay.count[STEP_F64]
                and not lay.count[MEAN_F64] and hasattr(self._ops, "round_sync")):
            # one call: the small plain-mean segments ride on the fused launch
            n_step = lay.count[STEP]
            self._ensure_moments()
            k_ready, rest = 0, None
            if pending:
                # the rows that have landed are folded while the last clients' rows still travel
                # (a client staged on arrival leaves about one row's transfer to hide: the
                # ResNet-18 x 100 round then ends 0.6 ms earlier)
                k_ready = self._slab.ready_prefix(self._row_of if len(self._row_of) == K else range(K))
                if max(1, K // 2) <= k_ready < K:      # (no landed row: nothing to start on -- wait)
                    rest = self._slab.last_copy
                else:
                    k_ready = 0
                    self._slab.end_round()
            ws = None
            if K > self._ops.kchunk() or k_ready:
                if self._step_flat is None:
                    self._step_flat = torch.empty(n_step, dtype=lay.float_dtype, device=self.device)
                ws = self._step_flat
            hyper = self._hyper()
            plan = None
            if ws is None and hasattr(self._ops, "RoundPlan"):
                # a steady federation repeats the same call: keep its arguments.  The round before
                # this one is recognised by identity of its tensors and equality of the rest, which
                # is cheaper than building the key of the table of kept plans.
                rows = self._row_indices(K)
                last = self._last_round
                if (last is not None and last[1] is self._slab.slab and last[2] is self._flats["f32"]
                        and last[3] is self._m and last[4] is self._v and last[5] == weights
                        and last[6] == rows and last[7] == hyper):
                    plan = last[0]
                else:
                    key = (K, self._slab.slab.data_ptr(), tuple(rows), tuple(weights),
                           self._flats["f32"].data_ptr(), id(self._m), id(self._v), tuple(hyper.values()))
                    plan = self._round_plans.get(key)
            if plan is None:
                means = [(lay.seg_byte[cls], out, lay.count[cls])
                         for cls, out in ((MEAN_F32, self._flats["f32"][n_step:]),
                                          (MEAN_I64, self._flats["i64"]), (MEAN_I32, self._flats["i32"]))
                         if lay.count[cls]]
                if ws is None and hasattr(self._ops, "RoundPlan"):
                    if len(self._round_plans) >= 16:
                        self._round_plans.clear()
                    plan = self._round_plans[key] = self._ops.RoundPlan(
                        self._rows(K, 0), weights, self._flats["f32"], self._m, self._v, self._OPT,
                        n=n_step, means=means, key=key, **hyper)
            if plan is not None and (self._last_round is None or self._last_round[0] is not plan):
                self._last_round = (plan, self._slab.slab, self._flats["f32"], self._m, self._v, list(weights),
                                    list(rows) if not isinstance(rows, range) else rows, dict(hyper))
            if events:
                events[0].record()
            if plan is not None:
                plan.launch()
            else:
                self._ops.round_sync(self._rows(K, 0), weights, self._flats["f32"], self._m, self._v,
                                     self._OPT, step_ws=ws, n=n_step, means=means, k_ready=k_ready,
                                     rest_event=rest, **hyper)
            if events:
                events[1].record()
            return
        if pending:
            self._slab.end_round()
        self._launch_step(K, weights, fused, events)
        self._launch_means(K)

    def _launch_step(self, K: int, weights, fused: bool = True, events=None) -> None:
        lay = self._layout
        n_step = lay.count[STEP]
        g = self._flats["f32"]
        if fused:
            if n_step:
                self._ensure_moments()
                ws = None
                if K > self._ops.kchunk():
                    if self._step_flat is None:
                        self._step_flat = torch.empty(n_step, dtype=lay.float_dtype, device=self.device)
                    ws = self._step_flat
                if events:
                    events[0].record()
                self._ops.reduce_opt(self._rows(K, 0), weights, g, self._m, self._v, self._OPT,
                                     step_ws=ws, n=n_step, **self._hyper())
                if events:
                    events[1].record()
            n64 = lay.count[STEP_F64]
            if n64:
                self._ensure_moments()
                ws = None
                if K > self._ops.kchunk():
                    if self._step_flat64 is None:
                        self._step_flat64 = torch.empty(n64, dtype=torch.float64, device=self.device)
                    ws = self._step_flat64
                self._ops.reduce_opt(self._rows(K, lay.seg_byte[STEP_F64]), weights, self._flats["f64"],
                                     self._m64, self._v64, self._OPT, step_ws=ws, n=n64, **self._hyper())
        else:
            for e in lay.entries:
                if e.name in self.step:
                    st = self.step[e.name]
                    if st.dtype != e.dtype or st.device != self.device:    # a user's step in another form
                        st = st.to(device=self.device, dtype=e.dtype)
                    self._ops.apply_step(lay.view(self._flats, e.name).view(-1), st.contiguous().view(-1))

    def _launch_means(self, K: int) -> None:
        """Entries that bypass the step: plain mean over clients (fedavg_aggregator.py:285-296)."""
        lay = self._layout
        n_step = lay.count[STEP]
        if lay.count[MEAN_F32]:
            self._ops.mean(self._rows(K, lay.seg_byte[MEAN_F32]), self._flats["f32"][n_step:],
                           n=lay.count[MEAN_F32])
        if lay.count[MEAN_F64]:
            self._ops.mean(self._rows(K, lay.seg_byte[MEAN_F64]), self._flats["f64"][lay.count[STEP_F64]:],
                           n=lay.count[MEAN_F64])
        if lay.count[MEAN_I64]:
            self._ops.mean(self._rows(K, lay.seg_byte[MEAN_I64]), self._flats["i64"], n=lay.count[MEAN_I64])
        if lay.count[MEAN_I32]:
            self._ops.mean(self._rows(K, lay.seg_byte[MEAN_I32]), self._flats["i32"], n=lay.count[MEAN_I32])

    # -- aggregate: fedavg_aggregator.py:150-320 -----------------------------------------------------
    def aggregate(self, local_models: Dict[Union[str, int], Union[Dict, OrderedDict]], **kwargs) -> Dict:
        """Weighted average of the clients' models (then the server-optimizer step); returns the
        new global model as CPU tensors.  Extra kwargs (client metadata dicts) are ignored."""
        secure_enabled = cfg_get(self.aggregator_configs, "use_secure_agg", False)
        assert not (secure_enabled and "_chunk_idx" in kwargs), (
            "Secure aggregation does not support chunked aggregation yet.")
        if "_chunk_idx" in kwargs:
            return self._aggregate_chunk(local_models, **kwargs)
        if secure_enabled and all(isinstance(m, dict) and m.get("type") == "masked_update_flat"
                                  for m in local_models.values()):
            return self._aggregate_masked_flat(local_models)
        with self._lock:
            first = next(iter(local_models.values()))
            if self._layout is None:
                if self._extra_state:
                    self._adopt_streamed_state()
                else:
                    self._init_global_state(first)
            K = len(local_models)
            fused = getattr(type(self).compute_steps, "_appfl_amd_native", False)
            self._stage(local_models, defer_wait=fused)
            if not fused:
                self._staged_for = local_models           # staged for the duration of this call
                try:
                    self.compute_steps(local_models)      # user override (may call super())
                finally:
                    self._staged_for = None
            out = self._round_pipelined(K, self._client_weights(local_models)) if fused else None
            if out is None:
                self._launch_round(K, self._client_weights(local_models), fused)
                out = self._download()
            self.step = {}
            self._sync_model(self.model, out)
            return out

    def _round_pipelined(self, K: int, weights) -> Optional[Dict[str, torch.Tensor]]:
        """The round after the LAST arrival as a pipeline over the windows of that client's row:
        every other row has landed (clients staged on arrival), so the fused kernel runs on the
        stretch of the flat axis a window covers as soon as that window has landed, and the new
        values of the stretch travel back while the next window is still on its way up -- PCIe
        carries both directions at once.  Same kernel on the same elements as the plain path
        (every element is independent), so the same bits.  Returns None when it does not apply
        (the caller then takes the plain path)."""
        lay, slab = self._layout, self._slab
        n_step = lay.count[STEP]
        if not (self._uploads_pending and self._fold_means and self.device.type == "cuda"
                and n_step >= self.PIPELINE_MIN_ELEMENTS and not lay.n_f64
                and hasattr(self._ops, "round_sync") and len(self._row_of) == K and K >= 2):
            return None
        rows = self._row_of
        wins = slab.row_windows.get(rows[-1])
        if not wins or slab.ready_prefix(rows[:-1]) < K - 1:
            return None
        out_f32 = self._dl.lease(lay.float_dtype, lay.n_f32)
        if out_f32 is None:
            return None
        fsize = lay.itemsize(STEP)         # bytes per weighted element in a row: 4, or 2 for a 16-bit model
        self._uploads_pending = False
        if getattr(self, "_down_stream", None) is None:
            self._down_stream = torch.cuda.Stream(device=self.device)
        compute, down = torch.cuda.current_stream(self.device), self._down_stream
        self._ensure_moments()
        ws = None
        if K > self._ops.kchunk():
            if self._step_flat is None:
                self._step_flat = torch.empty(n_step, dtype=lay.float_dtype, device=self.device)
            ws = self._step_flat
        g, ptrs, hyper = self._flats["f32"], self._rows(K, 0), self._hyper()
        # the clients folded on arrival: valid when they are this round's first K - 1 clients in
        # this order and their guessed weights are the real ones to the bit
        k_done, st = 0, self._folded_round
        self._folded_round = None
        if (st and st.get("w") and len(st["w"]) == K - 1
                and list(rows[:K - 1]) == list(range(K - 1)) and st["w"] == list(weights[:K - 1])
                and self._fold_flat is not None):
            k_done, ws = K - 1, self._fold_flat
            compute.wait_event(st["event"])
        done = 0
        for a, b, landed in wins:
            upto = n_step if b >= fsize * n_step else (b // fsize) & ~1023      # whole KiB-aligned stretches
            if upto <= done:
                continue
            compute.wait_event(landed)
            self._ops.round_sync([p + fsize * done for p in ptrs], weights, g[done:],
                                 None if self._m is None else self._m[done:],
                                 None if self._v is None else self._v[done:], self._OPT,
                                 step_ws=None if ws is None else ws[done:], n=upto - done,
                                 k_done=k_done, **hyper)
            folded = torch.cuda.Event()
            folded.record(compute)
            down.wait_event(folded)
            with torch.cuda.stream(down):
                out_f32[done:upto].copy_(g[done:upto], non_blocking=True)
            done = upto
        slab.end_round()                   # the plain means read every row whole
        if done < n_step:                  # (a row whose windows stop short of the weighted segment)
            self._ops.round_sync([p + fsize * done for p in ptrs], weights, g[done:],
                                 None if self._m is None else self._m[done:],
                                 None if self._v is None else self._v[done:], self._OPT,
                                 step_ws=None if ws is None else ws[done:], n=n_step - done,
                                 k_done=k_done, **hyper)
            out_f32[done:n_step].copy_(g[done:n_step], non_blocking=True)
        self._launch_means(K)
        host = {"f32": out_f32, "f64": torch.empty(0, dtype=torch.float64)}
        if lay.n_f32 > n_step:
            out_f32[n_step:].copy_(g[n_step:lay.n_f32], non_blocking=True)
        for key, n in (("i64", lay.count[MEAN_I64]), ("i32", lay.count[MEAN_I32])):
            host[key] = self._dl.fetch(self._flats[key], n, wait=False)
        out = lay.views(host)              # while the last transfers are in flight
        out.update({k: v.cpu() for k, v in self._extra_state.items()})
        compute.synchronize()
        down.synchronize()
        return out


    # -- secure-aggregation payloads: fedavg_aggregator.py:171-225 ------------------------------------
    def _aggregate_masked_flat(self, local_models) -> Dict:
        """Clients sent `{"type": "masked_update_flat", "flat", "shapes", "num_examples"}`: the
        pairwise masks cancel in the weighted sum of the flat fp32 vectors, which is ADDED to the
        global state (or becomes it on the first round).  The flat vector keeps the clients'
        entry order (`shapes`), so this path has a layout of its own in which every entry is
        fp32 and takes the sum."""
        with self._lock:
            payloads = list(local_models.values())
            shapes = [(name, tuple(int(d) for d in shp)) for name, shp in payloads[0]["shapes"]]
            first_round = self._layout is None
            if first_round:
                self._layout = Layout([(n, shp, torch.float32) for n, shp in shapes], None)
                self._flats = self._layout.alloc_state(self.device)
                self._slab = ClientSlab(self._layout, self.device)
                self._secure_layout = True
            elif not getattr(self, "_secure_layout", False) or \
                    self._layout.names != [n for n, _ in shapes]:
                raise NotImplementedError(
                    "mixing secure-aggregation payloads with plain state dicts (or changing the "
                    "payload's entries) on one aggregator is not supported on the MI355X path")
            n = self._layout.n_f32
            K = len(payloads)
            total_examples = sum(p["num_examples"] for p in payloads)
            weights = []
            for p in payloads:
                if self.client_weights_mode == "sample_size":
                    weights.append(float(p["num_examples"]) / float(total_examples))
                else:
                    weights.append(1.0 / K)
            self._slab.reserve(K)
            self._slab.begin_round()
            with self._slab._on_copy_stream():
                for k, p in enumerate(payloads):
                    flat = p["flat"].detach().reshape(-1)
                    if flat.numel() != n:
                        raise ValueError(f"flat payload has {flat.numel()} elements, expected {n}")
                    self._slab.row_view(k, 0, n, torch.float32).copy_(
                        flat.to(torch.float32), non_blocking=False)
                ev = self._slab._copy_event()
            self._slab.last_copy = ev
            self._slab.end_round()
            self._row_of = []
            ws = None
            if K > self._ops.kchunk():
                ws = torch.empty(n, dtype=torch.float32, device=self.device)
            self._ops.weighted_mix(self._rows(K, 0), weights, self._flats["f32"],
                                   "sum_set" if first_round else "sum_add", acc_ws=ws, n=n)
            out = FedAvgAggregator._download(self)
            self._sync_model(self.model, out)
            return out

    def _state_value(self, key) -> Optional[torch.Tensor]:
        """The device tensor that holds the global value of entry `key` (writable), or None."""
        main = self._layout
        if main is not None and key in main.by_name:
            return main.view(self._flats, key)
        return self._extra_state.get(key)

    def _state_written(self, key) -> None:
        """Called after the tensor `_state_value(key)` returned was overwritten (subclasses that
        keep further copies of the state refresh them)."""

    # -- streamed aggregation: fedavg_aggregator.py:406-497 ------------------------------------------
    def _aggregate_chunk(self, local_models, **kwargs) -> Dict:
        """One chunk of a streamed aggregation: only the keys in `_chunk_keys`.  A key the global
        state already holds (from earlier rounds of either kind) continues from its value, a
        new key starts from the first client's tensor (:416-423).  As in the reference every
        non-integer key takes the plain FedAvg step here -- no named-parameter test (:466-471)
        and no server optimizer (subclasses do not override this path)."""
        chunk_keys = list(list(kwargs["_chunk_keys"].values())[0])
        with self._lock:
            first = next(iter(local_models.values()))
            current = self._state_value

            init = {}
            for k in chunk_keys:
                cur = current(k)
                if cur is not None:
                    init[k] = cur
                elif k in first:
                    init[k] = first[k]
            tag = tuple((k, tuple(t.shape), t.dtype) for k, t in init.items())
            st = self._chunks.get(tag)
            if st is None:
                lay = Layout([(n, tuple(t.shape), t.dtype) for n, t in init.items()], None)
                st = _ChunkState(lay, lay.alloc_state(self.device), ClientSlab(lay, self.device))
                self._chunks[tag] = st
            lay, K = st.layout, len(local_models)
            with torch.no_grad():
                for n, t in init.items():
                    lay.view(st.flats, n).copy_(t.detach())
            st.slab.reserve(K)
            st.slab.begin_round()
            for k, model in enumerate(local_models.values()):
                st.slab.upload(k, model, fill_missing_from=st.flats)
            st.slab.end_round()
            rows = lambda off: [st.slab.row_ptr(k) + off for k in range(K)]
            ops_ = half_ops.for_layout(lay, self._unrounded_ops())
            if lay.count[STEP]:
                ws = None
                if K > ops_.kchunk():
                    ws = torch.empty(lay.count[STEP], dtype=lay.float_dtype, device=self.device)
                ops_.reduce_opt(rows(0), self._client_weights(local_models), st.flats["f32"],
                                step_ws=ws, n=lay.count[STEP])
            if lay.count[STEP_F64]:
                ws = None
                if K > ops_.kchunk():
                    ws = torch.empty(lay.count[STEP_F64], dtype=torch.float64, device=self.device)
                ops_.reduce_opt(rows(lay.seg_byte[STEP_F64]), self._client_weights(local_models),
                                st.flats["f64"], step_ws=ws, n=lay.count[STEP_F64])
            if lay.count[MEAN_I64]:
                ops_.mean(rows(lay.seg_byte[MEAN_I64]), st.flats["i64"], n=lay.count[MEAN_I64])
            if lay.count[MEAN_I32]:
                ops_.mean(rows(lay.seg_byte[MEAN_I32]), st.flats["i32"], n=lay.count[MEAN_I32])
            # the chunk's new values become part of the global state
            with torch.no_grad():
                for n in init:
                    cur = current(n)
                    if cur is not None:
                        cur.copy_(lay.view(st.flats, n))
                        self._state_written(n)
                    else:
                        self._extra_state[n] = lay.view(st.flats, n).clone()
            out = _download_flats(lay, st.flats, self._dl)
            self._sync_model(self.model, out)        # partial update (:447-452)
            return {k: out[k] for k in chunk_keys if k in out}


class FedAvgMAggregator(FedAvgAggregator):
    """Server momentum (fedavgm_aggregator.py:9-75).  Needs `server_momentum_param_1`."""

    _OPT = "avgm"

    def _hyper(self):
        return {"beta1": cfg_require(self.aggregator_configs, "server_momentum_param_1")}

    def _ensure_moments(self):
        if self._v is None:   # fedavgm_aggregator.py:37-45
            self._v = torch.zeros(self._layout.count[STEP], dtype=self._layout.float_dtype, device=self.device)
            if self._layout.count[STEP_F64]:
                self._v64 = torch.zeros(self._layout.count[STEP_F64], dtype=torch.float64, device=self.device)

    @_native
    def compute_steps(self, local_models):
        with self._lock:
            FedAvgAggregator.compute_steps(self, local_models)
            n, n64 = self._layout.count[STEP], self._layout.count[STEP_F64]
            if n or n64:
                self._ensure_moments()
            if n:
                self._ops.server_opt_step(self._step_flat, self._m, self._v, self._OPT, n=n, **self._hyper())
            if n64:
                self._ops.server_opt_step(self._step_flat64, self._m64, self._v64, self._OPT, n=n64, **self._hyper())


class _AdaptiveAggregator(FedAvgMAggregator):
    """Common part of FedAdam / FedYogi / FedAdagrad: first moment m, second moment v started at
    tau**2, step = lr*m / (sqrt(v) + tau)."""

    _NEEDS_BETA2 = True

    def _hyper(self):
        c = self.aggregator_configs
        return {
            "lr": cfg_require(c, "server_learning_rate"),
            "tau": cfg_require(c, "server_adapt_param"),
            "beta1": cfg_require(c, "server_momentum_param_1"),
            "beta2": cfg_require(c, "server_momentum_param_2") if self._NEEDS_BETA2 else 0.0,
        }

    def _ensure_moments(self):
        if self._m is None:   # fedadam_aggregator.py:41-50: zeros, zeros + tau**2
            n = self._layout.count[STEP]
            tau = cfg_require(self.aggregator_configs, "server_adapt_param")
            self._m = torch.zeros(n, dtype=self._layout.float_dtype, device=self.device)
            self._v = torch.empty(n, dtype=self._layout.float_dtype, device=self.device)
            self._ops.fill(self._v, tau ** 2)
            n64 = self._layout.count[STEP_F64]
            if n64:
                self._m64 = torch.zeros(n64, dtype=torch.float64, device=self.device)
                self._v64 = torch.empty(n64, dtype=torch.float64, device=self.device)
                self._ops.fill(self._v64, tau ** 2)


class FedAdamAggregator(_AdaptiveAggregator):
    """FedAdam (fedadam_aggregator.py:9-133): `server_learning_rate`, `server_adapt_param`,
    `server_momentum_param_1`, `server_momentum_param_2`."""

    _OPT = "adam"


class FedYogiAggregator(_AdaptiveAggregator):
    """FedYogi (fedyogi_aggregator.py:9-132); same configuration fields as FedAdam."""

    _OPT = "yogi"


class FedAdagradAggregator(_AdaptiveAggregator):
    """FedAdagrad (fedadagrad_aggregator.py:9-118): `server_learning_rate`,
    `server_adapt_param`, `server_momentum_param_1`."""

    _OPT = "adagrad"
    _NEEDS_BETA2 = False
