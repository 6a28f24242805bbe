"""FedAsync and FedBuff with the global model (and FedBuff's buffer) resident in HBM.

Drop-in for algorithm/aggregator/fedasync_aggregator.py:14-277 and
algorithm/aggregator/fedbuff_aggregator.py:9-158 of the reference: one client model per
`aggregate(client_id, local_model)` call, mixed into the global model with a
staleness-dependent weight.  The staleness bookkeeping (`global_step`, `client_step`) is host
state exactly as in the reference; the arithmetic is one streaming kernel per arrival.

Deliberate differences, none of which changes a returned value: `global_state` / `step` hold
device tensors; the caller's `local_model` is never modified (the reference subtracts the
global model from it in place, fedasync_aggregator.py:253-255) and the global state never
aliases the model's or the client's tensors (:115-118, :162).
"""

from __future__ import annotations

from typing import Any, Dict, Optional, OrderedDict, Union

import torch

from .. import half_ops, ops
from ..layout import MEAN_F32, MEAN_F64, MEAN_I32, MEAN_I64, STEP, STEP_F64, Layout
from ..staging import ClientSlab, HostDownloader
from .base_aggregator import (BaseAggregator, DeviceStateMixin, cfg_get, cfg_require,
                              resolve_device)
from .staleness import fedasync_staleness


class FedAsyncAggregator(BaseAggregator, DeviceStateMixin):
    """FedAsync (https://arxiv.org/pdf/1903.03934.pdf).  Configuration fields: `alpha`,
    `staleness_fn` (constant | polynomial | hinge), `staleness_fn_kwargs` {a, b},
    `client_weights_mode`, `num_clients`, `gradient_based`."""

    def __init__(self, model: Optional[torch.nn.Module] = None, aggregator_configs: Any = None,
                 logger: Optional[Any] = None):
        self.model = model
        self.logger = logger
        self.aggregator_configs = aggregator_configs if aggregator_configs is not None else {}
        self.client_weights_mode = cfg_get(self.aggregator_configs, "client_weights_mode", "equal")
        self.optimize_memory = cfg_get(self.aggregator_configs, "optimize_memory", True)
        if model is not None:
            self.named_parameters = {name for name, _ in self.model.named_parameters()}
        else:
            self.named_parameters = None
        kw = cfg_get(self.aggregator_configs, "staleness_fn_kwargs", None) or {}
        self.staleness_fn = fedasync_staleness(
            cfg_get(self.aggregator_configs, "staleness_fn", "constant"), **dict(kw))
        self.alpha = cfg_get(self.aggregator_configs, "alpha", 0.9)
        self.global_step = 0
        self.client_step = {}
        self.step: Dict[str, torch.Tensor] = {}
        self.device = resolve_device(self.aggregator_configs)
        self._ops = ops          # the flat kernels; the rounding ones once a 16-bit layout is known
        self._layout: Optional[Layout] = None
        self._flats: Optional[Dict[str, torch.Tensor]] = None
        self._slab: Optional[ClientSlab] = None
        self._step_flats: Optional[Dict[str, torch.Tensor]] = None
        self._dl = HostDownloader()
        self._init_lock()

    @property
    def global_state(self) -> Optional[Dict[str, torch.Tensor]]:
        if self._layout is None:
            return None
        return self._layout.views(self._flats)

    # -- get_parameters: fedasync_aggregator.py:54-101 -------------------------------------------
    def get_parameters(self, client_id: Optional[Union[str, int]] = None, **kwargs) -> Dict:
        """The global model; `client_id` records which version the client trains from."""
        with self._lock:
            if client_id is not None:
                self.client_step[client_id] = self.global_step
            if self._layout is None:
                if self.model is None:
                    raise ValueError("Model is not provided to the aggregator.")
                with torch.no_grad():
                    return {k: v.clone().detach() for k, v in self.model.state_dict().items()}
            return self._download()

    def _download(self) -> Dict[str, torch.Tensor]:
        lay = self._layout
        host = {}
        for key, n in lay.flat_sizes().items():
            host[key] = self._dl.fetch(self._flats[key], n, wait=False)
        out = lay.views(host)          # while the transfers are in flight
        self._dl.wait(self.device)
        return out

    def _init_global_state(self, local_model):
        """fedasync_aggregator.py:110-149: the model's entries when it has every name the client
        sent, else the client's tensors."""
        init = None
        if self.model is not None:
            model_state = self.model.state_dict()
            if all(n in model_state for n in local_model):
                init = {n: model_state[n] for n in local_model}
        if init is None:
            init = dict(local_model)
        self._layout = Layout([(n, tuple(t.shape), t.dtype) for n, t in init.items()],
                              self.named_parameters)
        self._flats = self._layout.alloc_state(self.device)
        with torch.no_grad():
            for n, t in init.items():
                self._layout.view(self._flats, n).copy_(t.detach())
        self._slab = ClientSlab(self._layout, self.device, slots=2)
        self._slab.reserve(1)
        self._ops = half_ops.for_layout(self._layout, ops)

    def _warn_unrecorded_download(self, client_id) -> None:
        if getattr(self, "_download_warning_issued", False) or self.logger is None:
            return
        self._download_warning_issued = True
        self.logger.warning(
            f"No model download is on record for client {client_id}: its staleness counts from "
            "its last upload. `get_parameters(client_id=...)` records downloads. (Shown once.)")

    _WARN_UNRECORDED = True

    def _alpha_t(self, client_id) -> float:
        """fedasync_aggregator.py:207-229 (Python double arithmetic, as in the reference)."""
        if client_id not in self.client_step:
            if self._WARN_UNRECORDED:
                self._warn_unrecorded_download(client_id)
            self.client_step[client_id] = 0
        sizes = getattr(self, "client_sample_size", None) or {}
        if self.client_weights_mode == "sample_size" and client_id in sizes:
            weight = sizes[client_id] / sum(sizes.values())
        else:
            weight = 1.0 / cfg_get(self.aggregator_configs, "num_clients", 1)
        return self.alpha * self.staleness_fn(self.global_step - self.client_step[client_id]) * weight

    def _upload(self, local_model) -> None:
        self._slab.begin_round()
        # every entry must be present: the reference indexes local_model[name] for each name
        for e in self._layout.entries:
            if e.name not in local_model:
                raise KeyError(e.name)
        self._slab.upload(0, local_model)
        self._slab.end_round()

    def _copy_non_step_entries(self) -> None:
        """Entries outside the step are replaced by the client's (fedasync_aggregator.py:162)."""
        lay, n_step = self._layout, self._layout.count[STEP]
        if lay.count[MEAN_F32]:
            self._flats["f32"][n_step:lay.n_f32].copy_(
                self._slab.row_view(0, lay.seg_byte[MEAN_F32], lay.count[MEAN_F32], lay.float_dtype))
        if lay.count[MEAN_F64]:
            self._flats["f64"][lay.count[STEP_F64]:lay.n_f64].copy_(
                self._slab.row_view(0, lay.seg_byte[MEAN_F64], lay.count[MEAN_F64], torch.float64))
        if lay.count[MEAN_I64]:
            self._flats["i64"][:lay.count[MEAN_I64]].copy_(
                self._slab.row_view(0, lay.seg_byte[MEAN_I64], lay.count[MEAN_I64], torch.int64))
        if lay.count[MEAN_I32]:
            self._flats["i32"][:lay.count[MEAN_I32]].copy_(
                self._slab.row_view(0, lay.seg_byte[MEAN_I32], lay.count[MEAN_I32], torch.int32))

    def _launch_arrival(self, alpha_t: float) -> None:
        """Enqueue the kernels of one arrival staged in row 0 of the slab."""
        n_step = self._layout.count[STEP]
        if n_step:
            self._ops.async_mix(self._slab.row_view(0, 0, n_step, self._layout.float_dtype), alpha_t,
                          self._flats["f32"],
                          gradient_based=cfg_get(self.aggregator_configs, "gradient_based", False),
                          sub_form=bool(self.optimize_memory), n=n_step)
        n64 = self._layout.count[STEP_F64]
        if n64:
            self._ops.async_mix(self._slab.row_view(0, self._layout.seg_byte[STEP_F64], n64, torch.float64),
                          alpha_t, self._flats["f64"],
                          gradient_based=cfg_get(self.aggregator_configs, "gradient_based", False),
                          sub_form=bool(self.optimize_memory), n=n64)
        self._copy_non_step_entries()

    # Arrivals of at least this many weighted elements run as a three-stage pipeline.
    PIPELINE_MIN_ELEMENTS = 4 << 20

    def _arrival_pipelined(self, local_model, alpha_t: float) -> Optional[Dict[str, torch.Tensor]]:
        """One arrival as a pipeline over windows of the row: while window w+1 is packed and
        uploaded, the mix kernel runs on window w and the new global values of window w-1 travel
        back -- PCIe carries both directions at once, so a large arrival costs about one
        transfer instead of upload + download.  Same kernels on the same elements as the plain
        path, so the same bits.  Returns None when it does not apply."""
        lay = self._layout
        n_step = lay.count[STEP]
        if n_step < self.PIPELINE_MIN_ELEMENTS or lay.n_f64 or any(
                not isinstance(t, torch.Tensor) or t.is_cuda for t in local_model.values()):
            return None
        out_f32 = self._dl.lease(lay.float_dtype, lay.n_f32)
        if out_f32 is None:
            return None
        if getattr(self, "_down_stream", None) is None:
            self._down_stream = torch.cuda.Stream(device=self.device)
        compute = torch.cuda.current_stream(self.device)
        g = self._flats["f32"]
        fsize = lay.itemsize(STEP)          # bytes per weighted element: 4, or 2 for a 16-bit model
        row = self._slab.row_view(0, 0, n_step, lay.float_dtype)
        gradient_based = cfg_get(self.aggregator_configs, "gradient_based", False)
        done = 0

        def on_window(a: int, b: int, landed) -> None:
            nonlocal done
            upto = min(n_step, b // fsize)      # weighted elements whose bytes have been enqueued
            if upto < n_step:
                upto -= upto % (16 // fsize)    # keep every piece 16-byte aligned
            if upto <= done:
                return
            compute.wait_event(landed)
            self._ops.async_mix(row[done:], alpha_t, g[done:], gradient_based=gradient_based,
                          sub_form=bool(self.optimize_memory), n=upto - done)
            mixed = torch.cuda.Event()
            mixed.record(compute)
            self._down_stream.wait_event(mixed)
            with torch.cuda.stream(self._down_stream):
                out_f32[done:upto].copy_(g[done:upto], non_blocking=True)
            done = upto

        self._slab.begin_round()
        for e in lay.entries:
            if e.name not in local_model:
                raise KeyError(e.name)
        self._slab.upload(0, local_model, on_window=on_window)
        self._slab.end_round()
        assert done == n_step
        self._copy_non_step_entries()
        host = {"f32": out_f32}
        if lay.n_f32 > n_step:
            out_f32[n_step:].copy_(g[n_step:lay.n_f32], non_blocking=True)
        for key, n in (("i64", lay.count[MEAN_I64]), ("i32", lay.count[MEAN_I32])):
            host[key] = self._dl.fetch(self._flats[key], n, wait=False)
        out = lay.views(host)          # while the last transfers are in flight
        compute.synchronize()
        self._down_stream.synchronize()
        return out

    # -- aggregate: fedasync_aggregator.py:103-184 ------------------------------------------------
    def aggregate(self, client_id: Union[str, int], local_model: Union[Dict, OrderedDict],
                  **kwargs) -> Dict:
        with self._lock:
            if self._layout is None:
                self._init_global_state(local_model)
            alpha_t = self._alpha_t(client_id)
            out = self._arrival_pipelined(local_model, alpha_t) if type(self)._launch_arrival is \
                FedAsyncAggregator._launch_arrival else None
            if out is None:
                self._upload(local_model)
                self._launch_arrival(alpha_t)
                out = self._download()
            self.global_step += 1
            self.client_step[client_id] = self.global_step
            self._sync_model(self.model, out)
            return out


class FedBuffAggregator(FedAsyncAggregator):
    """FedBuff (https://proceedings.mlr.press/v151/nguyen22b/nguyen22b.pdf): FedAsync steps are
    accumulated against a frozen global model and applied every `K` arrivals."""

    _WARN_UNRECORDED = False   # fedbuff_aggregator.py:126-127 does not warn

    def __init__(self, model=None, aggregator_configs: Any = None, logger=None):
        super().__init__(model, aggregator_configs, logger)
        self.buff_size = 0
        self.K = cfg_require(self.aggregator_configs, "K")

    def _launch_arrival(self, alpha_t: float, first: bool, flush: bool) -> None:
        """Enqueue the kernels of one buffered arrival staged in row 0 of the slab."""
        lay = self._layout
        n_step = lay.count[STEP]
        g, st = self._flats, self._step_flats
        if n_step:
            self._ops.buff_accum(self._slab.row_view(0, 0, n_step, lay.float_dtype), alpha_t, g["f32"],
                           st["f32"], first,
                           gradient_based=cfg_get(self.aggregator_configs, "gradient_based", False),
                           flush=flush, n=n_step)
        n64 = lay.count[STEP_F64]
        if n64:
            self._ops.buff_accum(self._slab.row_view(0, lay.seg_byte[STEP_F64], n64, torch.float64), alpha_t,
                           g["f64"], st["f64"], first,
                           gradient_based=cfg_get(self.aggregator_configs, "gradient_based", False),
                           flush=flush, n=n64)
        # entries outside the step are summed raw and flushed as a mean (:146-152, :84-86)
        for cls, key, dtype, lo in ((MEAN_F32, "f32", lay.float_dtype, n_step),
                                    (MEAN_F64, "f64", torch.float64, n64),
                                    (MEAN_I64, "i64", torch.int64, 0),
                                    (MEAN_I32, "i32", torch.int32, 0)):
            cnt = lay.count[cls]
            if not cnt:
                continue
            self._ops.buff_accum_raw(self._slab.row_view(0, lay.seg_byte[cls], cnt, dtype),
                               st[key][lo:lo + cnt], first, n=cnt)
            if flush:
                self._ops.buff_flush_mean(g[key][lo:lo + cnt], st[key][lo:lo + cnt], self.K, n=cnt)

    # -- aggregate: fedbuff_aggregator.py:25-105 with compute_steps :107-158 ------------------------
    def aggregate(self, client_id: Union[str, int], local_model: Union[Dict, OrderedDict],
                  **kwargs) -> Dict:
        with self._lock:
            if self._layout is None:
                self._init_global_state(local_model)
            lay = self._layout
            if self._step_flats is None:
                self._step_flats = lay.alloc_state(self.device)
            first = self.buff_size == 0            # :116-124 zero the buffer
            flush = self.buff_size + 1 == self.K   # :74-75
            early = None if flush else self._download_early()
            self._upload(local_model)
            self._launch_arrival(self._alpha_t(client_id), first, flush)
            self.buff_size += 1
            if flush:
                self.global_step += 1
                self.buff_size = 0
            self.client_step[client_id] = self.global_step   # :100
            if early is not None:
                torch.cuda.current_stream(self.device).synchronize()   # the arrival's kernels
                self._down_stream.synchronize()
                return early
            out = self._download()
            if flush:
                self._sync_model(self.model, out)
            return out

    def _download_early(self) -> Optional[Dict[str, torch.Tensor]]:
        """An arrival that does not flush the buffer leaves the global model as it is, so its
        copy for the caller (fedbuff_aggregator.py:101-105 clones it on every call) can travel
        back while the arrival is still being uploaded: the transfers share the full-duplex
        link instead of queueing.  None when no pinned lease is to be had."""
        lay = self._layout
        if lay.n_f32 < self.PIPELINE_MIN_ELEMENTS or lay.n_f64:
            return None
        host = {}
        for key, dtype, n in (("f32", lay.float_dtype, lay.n_f32), ("i64", torch.int64, lay.count[MEAN_I64]),
                              ("i32", torch.int32, lay.count[MEAN_I32])):
            if n == 0:
                host[key] = torch.empty(0, dtype=dtype)
                continue
            host[key] = self._dl.lease(dtype, n)
            if host[key] is None:
                return None
        if getattr(self, "_down_stream", None) is None:
            self._down_stream = torch.cuda.Stream(device=self.device)
        # after whatever changed the global model last (a flush on the compute stream)
        self._down_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._down_stream):
            for key, t in host.items():
                if t.numel():
                    t.copy_(self._flats[key][:t.numel()], non_blocking=True)
        return lay.views(host)
