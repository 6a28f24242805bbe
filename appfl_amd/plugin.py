"""Making the device aggregators discoverable by an installed APPFL.

APPFL resolves `server_configs.aggregator: "<Name>"` with
`getattr(import_module("appfl.algorithm.aggregator"), name)(model, cfg, logger)`
(misc/utils.py:132-144 of the reference), so a plugin becomes a drop-in by being an attribute
of that module.  `register()` adds the device classes there:

* under a prefixed name (default ``MI355XFedAvgAggregator`` ...) next to the stock ones, and
* with ``replace=True`` also under the stock names, so existing YAML files run unchanged.

Each registered class also inherits from the stock class of the same name (after ours in the
MRO, so every method resolves to the device implementation): the gRPC communicator only
forwards `set_model_chunk_size` to aggregators that pass
`isinstance(aggregator, FedAvgAggregator)` (comm/grpc/grpc_server_communicator.py:72-88).


`register(stage_on_arrival=True)` also makes upload-on-arrival real behind the reference's own
synchronous scheduler: it installs, into `appfl.algorithm.scheduler`, a subclass of the stock
`SyncScheduler` (algorithm/scheduler/sync_scheduler.py:10-107) whose `schedule` hands every
arriving model to `aggregator.stage(client_id, local_model)` -- right where the stock method
merely stores it (:46) -- and then runs the stock method unchanged.  With the stock scheduler
nothing calls `stage`, and a round whose K updates all sit in host memory when `aggregate` is
called is PCIe-bound; with it, the transfers overlap the wait for the remaining clients.
"""

from __future__ import annotations

import importlib
import threading
from typing import Dict, Optional

from . import aggregator as _dev

NAMES = ("FedAvgAggregator", "FedAvgMAggregator", "FedAdamAggregator", "FedYogiAggregator",
         "FedAdagradAggregator", "FedAsyncAggregator", "FedBuffAggregator",
         "FedCompassAggregator", "FedQueueAggregator", "IIADMMAggregator", "ICEADMMAggregator")


def _deserialize(model_bytes):
    """`torch.save` bytes -> state dict without executing anything from the payload."""
    import io

    import torch

    from . import wire
    try:
        return wire.bytes_to_model(model_bytes)
    except wire.WireFormatError:
        return torch.load(io.BytesIO(bytes(model_bytes)), map_location="cpu", weights_only=True)


def staging_scheduler_class(stock: type, name: str = "StagingSyncScheduler") -> type:
    """A subclass of the (stock) synchronous scheduler class `stock` that stages on arrival.

    `schedule(client_id, local_model, **kwargs)` first calls `aggregator.stage(client_id,
    local_model)` -- or `aggregator.stage_bytes(...)` when the caller hands over the update still
    in wire form (`torch.save` bytes; agent/server.py:104-105 deserialises them before the
    scheduler sees them, a communicator that skips that step may pass them straight through) and
    continues with the state dict of views it returns -- then the stock `schedule`, whose
    Futures, result object and epoch counting are untouched.  Aggregators without a `stage`
    method, chunks of a streamed round (`_chunk_idx`) and secure-aggregation payloads go
    straight to the stock method.  One extra lock keeps "stage, then store" atomic per arrival,
    so the slab's row order is the scheduler's dict order."""

    class _Staging(stock):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._arrival_lock = threading.Lock()

        def schedule(self, client_id, local_model, **kwargs):
            with self._arrival_lock:
                agg = self.aggregator
                if "_chunk_idx" not in kwargs:
                    if isinstance(local_model, (bytes, bytearray, memoryview)):
                        stage_bytes = getattr(agg, "stage_bytes", None)
                        if callable(stage_bytes):
                            local_model = stage_bytes(client_id, local_model)
                        else:
                            local_model = _deserialize(local_model)
                    elif (isinstance(local_model, dict) and local_model.get("type") != "masked_update_flat"
                            and callable(getattr(agg, "stage", None))):
                        agg.stage(client_id, local_model)
                return super().schedule(client_id, local_model, **kwargs)

    _Staging.__name__ = _Staging.__qualname__ = name
    _Staging.__doc__ = staging_scheduler_class.__doc__
    return _Staging


def register_staging_scheduler(prefix: str = "MI355X", replace: bool = False, module=None) -> Dict[str, type]:
    """Install the staging subclass of `SyncScheduler` into `appfl.algorithm.scheduler` (or
    `module`) as ``<prefix>SyncScheduler`` and, with `replace`, under the stock name too (the
    stock class stays reachable as ``StockSyncScheduler``), so that `scheduler: "SyncScheduler"`
    in existing YAML files resolves to it (misc/utils.py:147-159)."""
    target = module if module is not None else importlib.import_module("appfl.algorithm.scheduler")
    stock = getattr(target, "StockSyncScheduler", None) or getattr(target, "SyncScheduler")
    cls = staging_scheduler_class(stock, prefix + "SyncScheduler")
    setattr(target, prefix + "SyncScheduler", cls)
    out = {prefix + "SyncScheduler": cls}
    if replace:
        setattr(target, "StockSyncScheduler", stock)
        setattr(target, "SyncScheduler", cls)
        out["SyncScheduler"] = cls
    return out


def register(prefix: str = "MI355X", replace: bool = False, module=None,
             stage_on_arrival: bool = False, scheduler_module=None) -> Dict[str, type]:
    """Install the device aggregators into `appfl.algorithm.aggregator` (or `module`) and, with
    `stage_on_arrival`, the staging scheduler into `appfl.algorithm.scheduler` (or
    `scheduler_module`).  Returns {registered name: class}."""
    target = module if module is not None else importlib.import_module("appfl.algorithm.aggregator")
    out: Dict[str, type] = {}
    for name in NAMES:
        ours = getattr(_dev, name)
        stock: Optional[type] = getattr(target, name, None)
        bases = (ours,) if stock is None or issubclass(ours, stock) else (ours, stock)
        cls = type(prefix + name, bases, {"__module__": ours.__module__, "__doc__": ours.__doc__})
        setattr(target, prefix + name, cls)
        out[prefix + name] = cls
        if replace:
            setattr(target, "Stock" + name, stock)
            setattr(target, name, cls)
            out[name] = cls
    if stage_on_arrival:
        out.update(register_staging_scheduler(prefix, replace, scheduler_module))
    return out
