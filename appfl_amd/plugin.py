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
"""

from __future__ import annotations

import importlib
from typing import Dict, Optional

from . import aggregator as _dev

NAMES = ("FedAvgAggregator", "FedAvgMAggregator", "FedAdamAggregator", "FedYogiAggregator",
         "FedAdagradAggregator", "FedAsyncAggregator", "FedBuffAggregator",
         "FedCompassAggregator", "FedQueueAggregator", "IIADMMAggregator", "ICEADMMAggregator")


def register(prefix: str = "MI355X", replace: bool = False, module=None) -> Dict[str, type]:
    """Install the device aggregators into `appfl.algorithm.aggregator` (or `module`).
    Returns {registered name: class}."""
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
    return out
