"""The aggregator plugin interface (same as algorithm/aggregator/base_aggregator.py:5-26 of the
reference) plus the small helpers every device aggregator shares."""

from __future__ import annotations

import abc
import threading
from typing import Any, Dict, Optional, OrderedDict, Tuple, Union

import torch


class BaseAggregator:
    def set_client_sample_size(self, client_id: Union[str, int], sample_size: int):
        """Set the sample size of a client"""
        if not hasattr(self, "client_sample_size"):
            self.client_sample_size = {}
        self.client_sample_size[client_id] = sample_size

    @abc.abstractmethod
    def aggregate(self, *args, **kwargs) -> Union[Dict, OrderedDict, Tuple[Union[Dict, OrderedDict], Dict]]:
        """Aggregate local model(s) from clients and return the global model"""

    @abc.abstractmethod
    def get_parameters(self, **kwargs) -> Union[Dict, OrderedDict, Tuple[Union[Dict, OrderedDict], Dict]]:
        """Return global model parameters"""


def cfg_get(cfg: Any, key: str, default=None):
    """aggregator_configs may be an OmegaConf DictConfig, a plain dict or any attribute bag."""
    if cfg is None:
        return default
    getter = getattr(cfg, "get", None)
    if callable(getter):
        try:
            return getter(key, default)
        except TypeError:
            pass
    return getattr(cfg, key, default)


def cfg_require(cfg: Any, key: str):
    """Attribute-style access like the reference's `aggregator_configs.<key>`: a missing
    hyper-parameter is an error, not a silent default."""
    missing = object()
    v = cfg_get(cfg, key, missing)
    if v is missing:
        raise AttributeError(f"aggregator_configs has no '{key}'")
    return v


def resolve_device(cfg: Any) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "appfl_amd aggregators run on an AMD GPU (HIP); no GPU is visible and there is no "
            "CPU fallback")
    dev = cfg_get(cfg, "device", None)
    if dev is None or str(dev) == "cpu":
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device(dev)
    if d.type != "cuda":
        raise ValueError(f"aggregator device must be a GPU, got {dev}")
    return d if d.index is not None else torch.device("cuda", torch.cuda.current_device())


class _DeviceLock:
    """Re-entrant lock that also makes the aggregator's GPU the current device while held:
    kernel launches and stream queries go to the current device, and a server process may
    have set another one (or run several aggregators on different GPUs)."""

    def __init__(self, device: Optional[torch.device]):
        self._lock = threading.RLock()
        self._index = device.index if device is not None and device.type == "cuda" else None
        self._prev = []

    def __enter__(self):
        self._lock.acquire()
        prev = None
        if self._index is not None:
            prev = torch.cuda.current_device()
            if prev != self._index:
                torch.cuda.set_device(self._index)
        self._prev.append(prev)
        return self

    def __exit__(self, *exc):
        prev = self._prev.pop()
        if prev is not None and prev != self._index:
            torch.cuda.set_device(prev)
        self._lock.release()
        return False


class DeviceStateMixin:
    """Lock shared by aggregate / get_parameters: the schedulers serialise `aggregate`, but
    `get_parameters` is called without their lock (base_scheduler.py:67,76 of the reference),
    and here both enqueue work on one stream and read one set of device buffers."""

    def _init_lock(self):
        self._lock = _DeviceLock(getattr(self, "device", None))

    @staticmethod
    def _sync_model(model: Optional[torch.nn.Module], state: Dict[str, torch.Tensor]):
        if model is not None:
            model.load_state_dict(state, strict=False)
