"""FedCompass and FedQueue with the global model resident in HBM.

Drop-in for algorithm/aggregator/fedcompass_aggregator.py:14-387 and
algorithm/aggregator/fedqueue_aggregator.py:13-191 of the reference.  Both are weighted mixes
of whole client models (no subtraction of the global model first):

* FedCompass, one arrival:  g = g*(1 - a_t) + x*a_t          (gradient_based: g = g - x*a_t)
* FedCompass, a group:      g = g*(1 - sum a_k) + sum x_k*a_k (gradient_based: sequential -=)
* FedQueue:                 g = sum x_k*f_k, f normalised    (gradient_based: sequential -=)

with a_k = alpha * staleness_fn(staleness_k) / num_clients and f_k = staleness_fn * data share.
Entries that are integers or not `named_parameters` take the plain mean of the group (or, for
a single FedCompass arrival, the client's value).  The scalar prologue is Python double
arithmetic exactly as in the reference; one fa_weighted_mix launch does the rest.
"""

from __future__ import annotations

from typing import Any, Dict, Optional, OrderedDict, Union

import torch

from .. import half_ops, ops
from ..layout import MEAN_F32, MEAN_F64, MEAN_I32, MEAN_I64, STEP, STEP_F64, Layout
from ..staging import ClientSlab, HostDownloader
from .base_aggregator import BaseAggregator, DeviceStateMixin, cfg_get, resolve_device
from .staleness import fedasync_staleness, fedqueue_staleness


class _MixAggregator(BaseAggregator, DeviceStateMixin):
    def __init__(self, model: Optional[torch.nn.Module] = None, aggregator_configs: Any = None,
                 logger: Optional[Any] = None):
        self.model = model
        self.logger = logger
        self.aggregator_configs = aggregator_configs if aggregator_configs is not None else {}
        self.optimize_memory = cfg_get(self.aggregator_configs, "optimize_memory", True)
        if model is not None:
            self.named_parameters = {name for name, _ in model.named_parameters()}
        else:
            self.named_parameters = None
        self.device = resolve_device(self.aggregator_configs)
        self._layout: Optional[Layout] = None
        self._flats: Optional[Dict[str, torch.Tensor]] = None
        self._slab: Optional[ClientSlab] = None
        self._acc_ws: Optional[torch.Tensor] = None
        self._acc_ws64: Optional[torch.Tensor] = None
        self._dl = HostDownloader()
        self._init_lock()

    @property
    def global_state(self) -> Optional[Dict[str, torch.Tensor]]:
        if self._layout is None:
            return None
        return self._layout.views(self._flats)

    def get_parameters(self, **kwargs) -> Dict:
        with self._lock:
            if self._layout is None:
                if self.model is None:
                    raise ValueError("Model is not provided to the aggregator.")
                with torch.no_grad():
                    return {k: v.clone().detach() for k, v in self.model.state_dict().items()}
            return self._download()

    def _download(self) -> Dict[str, torch.Tensor]:
        lay = self._layout
        host = {}
        for key, n in (("f32", lay.n_f32), ("f64", lay.n_f64), ("i64", lay.count[MEAN_I64]),
                       ("i32", lay.count[MEAN_I32])):
            host[key] = self._dl.fetch(self._flats[key], n, wait=False)
        out = lay.views(host)          # while the transfers are in flight
        self._dl.wait(self.device)
        return out

    def _init_global_state(self, first_model):
        """The model's entries for the names the first client sent, else the client's tensors
        (fedcompass_aggregator.py:57-147, fedqueue_aggregator.py:45-71)."""
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
        self._ops = half_ops.for_layout(self._layout, ops)      # 16-bit models: the rounding kernels

    def _stage(self, models) -> int:
        K = len(models)
        for m in models:
            for e in self._layout.entries:
                if e.name not in m:
                    raise KeyError(e.name)
        self._slab.reserve(K)
        self._slab.begin_round()
        for k, m in enumerate(models):
            self._slab.upload(k, m)
        self._slab.end_round()
        return K

    def _rows(self, K: int, byte_offset: int):
        return [self._slab.row_ptr(k) + byte_offset for k in range(K)]

    def _mix(self, K: int, weights, mode: str, coeff: float = 0.0) -> None:
        """The weighted entries: fp32 and float64 segments take the same mix (for float64 tensors
        PyTorch keeps the Python scalars in double, and so does the f64 kernel)."""
        lay = self._layout
        for cls, key, dtype, ws_attr in ((STEP, "f32", lay.float_dtype, "_acc_ws"),
                                         (STEP_F64, "f64", torch.float64, "_acc_ws64")):
            n = lay.count[cls]
            if not n:
                continue
            ws = None
            if K > self._ops.kchunk() and mode != "subtract":
                if getattr(self, ws_attr) is None:
                    setattr(self, ws_attr, torch.empty(n, dtype=dtype, device=self.device))
                ws = getattr(self, ws_attr)
            self._ops.weighted_mix(self._rows(K, lay.seg_byte[cls]), weights, self._flats[key], mode, coeff,
                             acc_ws=ws, n=n)

    def _non_step_segments(self):
        """(class, flat key, dtype, offset in the flat vector) of the entries that bypass the mix."""
        lay = self._layout
        return ((MEAN_F32, "f32", lay.float_dtype, lay.count[STEP]),
                (MEAN_F64, "f64", torch.float64, lay.count[STEP_F64]),
                (MEAN_I64, "i64", torch.int64, 0), (MEAN_I32, "i32", torch.int32, 0))

    def _plain_means(self, K: int) -> None:
        lay = self._layout
        for cls, key, _, lo in self._non_step_segments():
            if lay.count[cls]:
                self._ops.mean(self._rows(K, lay.seg_byte[cls]), self._flats[key][lo:], n=lay.count[cls])

    def _replace_non_step(self) -> None:
        lay = self._layout
        for cls, key, dtype, lo in self._non_step_segments():
            cnt = lay.count[cls]
            if cnt:
                self._flats[key][lo:lo + cnt].copy_(self._slab.row_view(0, lay.seg_byte[cls], cnt, dtype))

    def _finish(self) -> Dict[str, torch.Tensor]:
        out = self._download()
        self._sync_model(self.model, out)
        return out


class FedCompassAggregator(_MixAggregator):
    """FedCompass (https://arxiv.org/abs/2309.14675).  Configuration fields: `alpha`,
    `staleness_fn` (constant | polynomial | hinge), `staleness_fn_kwargs`, `num_clients`,
    `gradient_based`."""

    def __init__(self, model=None, aggregator_configs: Any = None, logger=None):
        super().__init__(model, aggregator_configs, logger)
        kw = cfg_get(self.aggregator_configs, "staleness_fn_kwargs", None) or {}
        self.staleness_fn = fedasync_staleness(
            cfg_get(self.aggregator_configs, "staleness_fn", "constant"), **dict(kw))
        self.alpha = cfg_get(self.aggregator_configs, "alpha", 0.9)

    def aggregate(self, client_id: Optional[Union[str, int]] = None,
                  local_model: Optional[Union[Dict, OrderedDict]] = None,
                  local_models: Optional[Dict[Union[str, int], Union[Dict, OrderedDict]]] = None,
                  staleness: Optional[Union[int, Dict[Union[str, int], int]]] = None,
                  **kwargs) -> Dict:
        gb = cfg_get(self.aggregator_configs, "gradient_based", False)
        weight = 1.0 / cfg_get(self.aggregator_configs, "num_clients", 1)
        with self._lock:
            if client_id is not None and local_model is not None:
                # one arrival: fedcompass_aggregator.py:152-206
                if self._layout is None:
                    self._init_global_state(local_model)
                self._stage([local_model])
                alpha_t = self.alpha * self.staleness_fn(staleness) * weight
                if gb:
                    self._mix(1, [alpha_t], "subtract")
                else:
                    self._mix(1, [alpha_t], "convex_one", 1 - alpha_t)
                self._replace_non_step()
                return self._finish()
            # a group of clients: fedcompass_aggregator.py:208-338
            if self._layout is None:
                self._init_global_state(next(iter(local_models.values())))
            K = self._stage(list(local_models.values()))
            alphas, alpha_t_sum = [], 0
            for cid in local_models:
                alpha_t = self.alpha * self.staleness_fn(staleness[cid]) * weight
                alpha_t_sum += alpha_t
                alphas.append(alpha_t)
            if gb:
                self._mix(K, alphas, "subtract")
            else:
                self._mix(K, alphas, "convex", 1 - alpha_t_sum)
            self._plain_means(K)
            return self._finish()


class FedQueueAggregator(_MixAggregator):
    """FedQueue: data-size client weights with staleness decay, normalised over the group.
    Configuration fields: `staleness_fn` (constant | harmonic | exponential),
    `staleness_fn_kwargs` {beta}, `gradient_based`."""

    def __init__(self, model=None, aggregator_configs: Any = None, logger=None):
        super().__init__(model, aggregator_configs, logger)
        kw = cfg_get(self.aggregator_configs, "staleness_fn_kwargs", None) or {}
        self.staleness_fn = fedqueue_staleness(
            cfg_get(self.aggregator_configs, "staleness_fn", "constant"), **dict(kw))

    def _get_aggregation_factors(self, local_models, staleness) -> Dict:
        """Un-normalised factor per client: s(staleness) x its share of the group's samples, or
        x 1/K when a sample size is unknown or they sum to zero (fedqueue_aggregator.py:138-169)."""
        sizes = getattr(self, "client_sample_size", None)
        share = None
        if sizes is not None and all(cid in sizes for cid in local_models):
            total = sum(sizes[cid] for cid in local_models)
            if total > 0:
                share = {cid: sizes[cid] / total for cid in local_models}
        if share is None:
            if self.logger is not None:
                self.logger.warning("FedQueueAggregator: sample sizes missing for some clients of "
                                    "this group; weighting them equally.")
            share = {cid: 1.0 / len(local_models) for cid in local_models}
        return {cid: self.staleness_fn(staleness[cid]) * share[cid] for cid in local_models}

    def aggregate(self, local_models: Dict[Union[str, int], Union[Dict, OrderedDict]],
                  staleness: Dict[Union[str, int], int],
                  local_steps: Optional[Dict[Union[str, int], int]] = None) -> Dict:
        with self._lock:
            if self._layout is None:
                self._init_global_state(next(iter(local_models.values())))
            gb = cfg_get(self.aggregator_configs, "gradient_based", False)
            factors = self._get_aggregation_factors(local_models, staleness)
            total = sum(factors.values())
            factors = {cid: f / total for cid, f in factors.items()}
            K = self._stage(list(local_models.values()))
            self._mix(K, [float(factors[cid]) for cid in local_models], "subtract" if gb else "replace")
            self._plain_means(K)
            return self._finish()
