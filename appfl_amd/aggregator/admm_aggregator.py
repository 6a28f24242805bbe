"""IIADMM / ICEADMM server updates with every client's primal and dual vectors resident in HBM.

Drop-in for the reference's classes of the same names (constructor, `aggregate`,
`get_parameters`, the `prim_res` / `dual_res` / `penalty` / `dual_states` attributes):

* algorithm/aggregator/iiadmm_aggregator.py:12-171
* algorithm/aggregator/iceadmm_aggregator.py:12-175

Clients send ``{"primal": state dict, "penalty": float}`` (ICEADMM also ``"dual"``: the named
parameters' duals), or None to stand by their last message.  The MODEL is the global state: its
named parameters are read at the start of every round and loaded back at the end, the other
entries of its state dict (batch-norm statistics) pass through unchanged.  One `aggregate` call =
upload of the round's messages, ONE kernel over the K clients' rows (dual update, sum, division
and both residual norms in one pass -- fa_admm_ii / fa_admm_ice), one download.

Each client keeps two primal rows on the device, written in turn: the row of the client's last
message is the "previous primal" of the dual residual (:141-149 / :137-145) when its next message
arrives, so no copy of the K rows is made between rounds.

Deliberate differences from the reference, none of which changes a returned tensor: the state
dictionaries hold DEVICE tensors (views of the rows); the two residual norms are float64 sums of
the same fp32 squares (the reference adds them in torch.sum's own order), equal to ~1e-6
relative; parameters must be fp32.
"""

from __future__ import annotations

from typing import Any, Dict, Optional, OrderedDict, Union

import torch

from .. import _native as N
from .. import ops
from ..layout import STEP, Layout
from ..staging import ClientSlab, HostDownloader
from .base_aggregator import BaseAggregator, DeviceStateMixin, cfg_get, resolve_device


class _ADMMAggregator(BaseAggregator, DeviceStateMixin):
    _ALG = "ii"
    _CLIENT_DUALS = False      # clients send their duals (ICEADMM) / the server keeps them (IIADMM)

    def __init__(self, model: torch.nn.Module, aggregator_configs: Any = None,
                 logger: Optional[Any] = None, *, device=None):
        self.model = model
        self.logger = logger
        self.aggregator_configs = aggregator_configs if aggregator_configs is not None else {}
        self.optimize_memory = cfg_get(self.aggregator_configs, "optimize_memory", True)
        # as in the reference, a missing model fails here (AttributeError on None)
        params = list(self.model.named_parameters())
        self.named_parameters = {name for name, _ in params}
        for name, p in params:
            if p.dtype != torch.float32:
                raise NotImplementedError(
                    f"parameter {name} is {p.dtype}: the MI355X ADMM kernels take fp32 parameters "
                    "(there is no CPU fallback)")
        self.is_first_iter = True
        self.penalty: Dict[Any, float] = {}
        self.prim_res = 0
        self.dual_res = 0
        self.global_state: Dict[str, torch.Tensor] = {}     # unused by the reference too
        self.device = (torch.device(device) if device is not None
                       else resolve_device(self.aggregator_configs))
        self._layout = Layout([(n, tuple(p.shape), p.dtype) for n, p in params], None)
        self._n = self._layout.count[STEP]
        self._gslab = ClientSlab(self._layout, self.device, slots=1)
        self._gslab.reserve(1)
        self._primal = (ClientSlab(self._layout, self.device), ClientSlab(self._layout, self.device))
        self._dual_slab: Optional[ClientSlab] = (ClientSlab(self._layout, self.device)
                                                 if self._CLIENT_DUALS else None)
        self._dual: Optional[torch.Tensor] = None     # IIADMM: [clients, row] fp32, zero at the start
        self._rows: Dict[Any, int] = {}               # client id -> row, fixed by the first round
        self._cur: Dict[Any, int] = {}                # client id -> which primal slab holds its last message
        self._res_ws = torch.zeros(N.FA_ADMM_RES_WS, dtype=torch.float64, device=self.device)
        self._norms = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._carry: Optional[torch.Tensor] = None
        self._prev_same: Dict[Any, bool] = {}         # client id -> its previous primal IS its current one
        self._staged: Dict[Any, Any] = {}             # client id -> message uploaded on arrival (`stage`)
        self._dl = HostDownloader()
        self._init_lock()

    # -- attributes the reference exposes (device views) ------------------------------------------
    def _row_views(self, row: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self._layout.views({"f32": row})

    def _primal_row(self, cid, previous: bool = False) -> torch.Tensor:
        which = self._cur[cid] ^ (1 if previous else 0)
        return self._primal[which].row_view(self._rows[cid], 0, self._n, torch.float32)

    @property
    def primal_states(self) -> Dict[Any, Dict[str, torch.Tensor]]:
        with self._lock:
            return {cid: self._row_views(self._primal_row(cid)) for cid in self._cur}

    primal_states_curr = primal_states

    @property
    def primal_states_prev(self) -> Dict[Any, Dict[str, torch.Tensor]]:
        """The primal rows as they stood before the last round (empty until a second round has
        run): a client's other row if it sent a new message in that round, else its current one."""
        with self._lock:
            return {cid: self._row_views(self._primal_row(cid, previous=not same))
                    for cid, same in self._prev_same.items()}

    @property
    def dual_states(self) -> Dict[Any, Dict[str, torch.Tensor]]:
        with self._lock:
            if self._CLIENT_DUALS:
                return {cid: self._row_views(self._dual_slab.row_view(2 * r + self._cur[cid], 0, self._n,
                                                                      torch.float32))
                        for cid, r in self._rows.items() if cid in self._cur}
            if self._dual is None:
                return {}
            return {cid: self._row_views(self._dual[r, :self._n]) for cid, r in self._rows.items()}

    # -- the round ------------------------------------------------------------------------------
    def _first_round(self, local_models) -> None:
        """iiadmm_aggregator.py:53-64 / iceadmm_aggregator.py:53-59: the clients of the first
        round are THE clients; their number divides IIADMM's sum from then on."""
        self.num_clients = len(local_models)
        self._rows = {cid: k for k, cid in enumerate(local_models)}
        K = self.num_clients
        for slab in self._primal:
            slab.reserve(K)
        if self._CLIENT_DUALS:
            # two dual rows per client, like the primal: a message staged on arrival lands in the
            # row the current round does not read, and is only adopted when `aggregate` takes it
            self._dual_slab.reserve(2 * K)
        else:
            self._dual = torch.zeros((K, self._layout.row_bytes // 4), dtype=torch.float32,
                                     device=self.device)

    def stage(self, client_id, message) -> None:
        """Optional hook for the scheduler: upload ONE client's message the moment it arrives, so
        the PCIe transfer overlaps the wait for the other clients.  `aggregate` reuses the staged
        rows when it later receives the same object for this client.  Only from the second round
        on (the first round fixes the set of clients); not calling it at all is equally valid."""
        with self._lock:
            if not self._rows or client_id not in self._rows or message is None:
                return
            for name in self._layout.names:
                if name not in message["primal"] or (self._CLIENT_DUALS and name not in message["dual"]):
                    raise KeyError(name)
            which = 1 - self._cur[client_id] if client_id in self._cur else 0
            slab, row = self._primal[which], self._rows[client_id]
            slab.begin_round()
            slab.upload(row, message["primal"])
            if self._CLIENT_DUALS:
                self._dual_slab.begin_round()
                self._dual_slab.upload(2 * row + which, message["dual"])
            self._staged[client_id] = message

    def aggregate(self, local_models: Dict[Union[str, int], Union[Dict, OrderedDict, None]],
                  **kwargs) -> Dict:
        with self._lock:
            if not self._rows:
                self._first_round(local_models)
            for cid, m in local_models.items():
                if cid not in self._rows:
                    raise KeyError(cid)          # the reference fails on a client it has no state for
                if m is None:
                    if cid not in self._cur:     # no earlier message to stand by
                        raise KeyError(next(iter(self._layout.names)))
                    continue
                for name in self._layout.names:
                    if name not in m["primal"] or (self._CLIENT_DUALS and name not in m["dual"]):
                        raise KeyError(name)
            model_state = self.model.state_dict()
            # this round's g: the model's named parameters (:66-69 / :61-64)
            self._gslab.begin_round()
            self._gslab.upload(0, model_state)
            used = {self._gslab}
            fresh = set()
            for cid, m in local_models.items():
                if m is None:
                    continue                      # :72 / :67: its last primal, dual and penalty stand
                which = 1 - self._cur[cid] if cid in self._cur else 0
                slab, row = self._primal[which], self._rows[cid]
                if self._staged.get(cid) is m:
                    # uploaded on arrival (`stage`): the rows are on their way already
                    used.add(slab)
                    if self._CLIENT_DUALS:
                        used.add(self._dual_slab)
                else:
                    if slab not in used:
                        slab.begin_round()
                        used.add(slab)
                    slab.upload(row, m["primal"])
                    if self._CLIENT_DUALS:
                        if self._dual_slab not in used:
                            self._dual_slab.begin_round()
                            used.add(self._dual_slab)
                        self._dual_slab.upload(2 * row + which, m["dual"])
                self.penalty[cid] = m["penalty"]
                fresh.add(cid)
            self._staged = {}
            first = self.is_first_iter
            prev = None
            if not first:
                # a client that sent nothing new stands by its last message: previous == current
                # (`_cur` still names the rows of the last messages: it flips below)
                prev = [self._primal[self._cur[cid]].row_ptr(self._rows[cid]) for cid in local_models]
            for cid in fresh:
                self._cur[cid] = 1 - self._cur[cid] if cid in self._cur else 0
            primal = [self._primal[self._cur[cid]].row_ptr(self._rows[cid]) for cid in local_models]
            if self._CLIENT_DUALS:
                dual = [self._dual_slab.row_ptr(2 * self._rows[cid] + self._cur[cid]) for cid in local_models]
            else:
                dual = [self._dual.data_ptr() + self._rows[cid] * self._layout.row_bytes
                        for cid in local_models]
            for slab in used:
                slab.end_round()
                if slab.has_device_sources:
                    slab.wait_uploads()
            K = len(local_models)
            if K > N.FA_ADMM_KCHUNK and self._carry is None:
                self._carry = torch.empty(ops.admm_carry_floats(self._n), dtype=torch.float32,
                                          device=self.device)
            g = self._gslab.row_view(0, 0, self._n, torch.float32)
            ops.admm_round(self._ALG, primal, dual, prev, [float(self.penalty[cid]) for cid in local_models],
                           g, self._norms, self._res_ws, num_clients=self.num_clients,
                           carry_ws=self._carry, n=self._n)
            host = self._dl.fetch(g, self._n, wait=False)
            norms = self._norms.cpu()            # synchronises the stream: `host` has landed too
            HostDownloader.wait(self.device)
            self.prim_res = float(norms[0])
            if not first:
                self.dual_res = float(norms[1])
            if not first:
                # :113-122 / :109-118: the snapshot covers every client; one without a new message
                # this round has previous == current
                self._prev_same = {cid: cid not in fresh for cid in self._cur}
            self.is_first_iter = False
            new = self._layout.views({"f32": host})
            with torch.no_grad():
                out = {k: (new[k] if k in new else v.clone().detach()) for k, v in model_state.items()}
            self.model.load_state_dict(out)
            return out

    def get_parameters(self, **kwargs) -> Dict:
        with self._lock, torch.no_grad():
            return {k: v.clone().detach() for k, v in self.model.state_dict().items()}


class IIADMMAggregator(_ADMMAggregator):
    """IIADMM (iiadmm_aggregator.py:12-171; used with the IIADMM trainer): the server keeps one
    dual vector per client, updates it with the client's penalty and averages primal - dual /
    penalty over the clients of the first round."""

    _ALG = "ii"
    _CLIENT_DUALS = False


class ICEADMMAggregator(_ADMMAggregator):
    """ICEADMM (iceadmm_aggregator.py:12-175; used with the ICEADMM trainer): clients send primal
    and dual; the new model is the penalty-weighted sum of the primals plus the scaled duals."""

    _ALG = "ice"
    _CLIENT_DUALS = True
