"""Minimal stand-ins for the reference's scheduling plumbing, enough to drive an aggregator the
way APPFL's serial simulation does (examples/serial/run_serial.py:70-91 and
tests/test_mnist.py:601-624): per-client `global_update(..., blocking=False)` returning a
Future, aggregation when the last client of the round has arrived, one shared result object for
every client's Future (algorithm/scheduler/sync_scheduler.py:25-107); the initial-model barrier
of algorithm/scheduler/base_scheduler.py:38-79; and the lock-and-forward of
algorithm/scheduler/async_scheduler.py:20-42.  Not a scheduler re-implementation: no
communicators, no virtual time.  Their behaviour is pinned against records produced by the
reference's own schedulers (tests/golden).
"""

from __future__ import annotations

import threading
from concurrent.futures import Future
from typing import Any, Dict, Union


class SerialSyncHarness:
    def __init__(self, aggregator: Any, num_clients: int, stage_on_arrival: bool = True):
        self.aggregator = aggregator
        self.num_clients = num_clients
        self.stage_on_arrival = stage_on_arrival     # False: store only, as the reference's scheduler does
        self.local_models: Dict[Union[int, str], Dict] = {}
        self.aggregation_kwargs: Dict[str, Dict] = {}
        self.future: Dict[Union[int, str], Future] = {}
        self._num_global_epochs = 0
        self._access_lock = threading.Lock()
        self.same_init_model = True
        self._init_requests = 0
        self._init_futures: list = []

    def set_sample_size(self, client_id, sample_size: int) -> None:
        """ServerAgent.set_sample_size (agent/server.py:146-190), without the sync barrier."""
        self.aggregator.set_client_sample_size(client_id, sample_size)

    def get_parameters(self, **kwargs):
        """BaseScheduler.get_parameters (base_scheduler.py:38-79): with `init_model` (the default)
        and neither `serial_run` nor `globus_compute_run` every caller gets a Future that resolves
        to ONE shared initial model once `num_clients` requests have arrived; otherwise the
        aggregator's reply."""
        if (kwargs.get("init_model", True) and self.same_init_model
                and not kwargs.get("serial_run", False) and not kwargs.get("globus_compute_run", False)):
            self._init_requests += kwargs.get("num_batched_clients", 1)
            future: Future = Future()
            self._init_futures.append(future)
            if self._init_requests == self.num_clients:
                self._init_requests = 0
                init_model = self.aggregator.get_parameters(**kwargs)
                while self._init_futures:
                    self._init_futures.pop().set_result(init_model)
            return future
        return self.aggregator.get_parameters(**kwargs)

    def global_update(self, client_id, local_model: Dict, blocking: bool = False, **kwargs):
        """ServerAgent.global_update -> SyncScheduler.schedule (sync_scheduler.py:40-107)."""
        with self._access_lock:
            future: Future = Future()
            self.local_models[client_id] = local_model
            # scheduler-side staging: start this client's upload while the others are awaited
            # (a chunk of a streamed round is aggregated per chunk, not staged as a whole update)
            stage = getattr(self.aggregator, "stage", None)
            if self.stage_on_arrival and callable(stage) and "_chunk_idx" not in kwargs:
                stage(client_id, local_model)
            for key, value in kwargs.items():
                self.aggregation_kwargs.setdefault(key, {})[client_id] = value
            self.future[client_id] = future
            if len(self.local_models) == self.num_clients:
                final_chunk = True       # a streamed round counts once, on its last chunk (:60-70)
                if "_chunk_idx" in self.aggregation_kwargs and "_total_chunks" in self.aggregation_kwargs:
                    idx = list(self.aggregation_kwargs["_chunk_idx"].values())[0]
                    final_chunk = idx == list(self.aggregation_kwargs["_total_chunks"].values())[0] - 1
                aggregated = self.aggregator.aggregate(self.local_models, **self.aggregation_kwargs)
                futures = dict(self.future)
                self.local_models.clear()
                self.aggregation_kwargs.clear()
                self.future.clear()
                while futures:
                    _, f = futures.popitem()
                    f.set_result(aggregated)          # the SAME object for every client
                if final_chunk:
                    self._num_global_epochs += 1
        return future.result() if blocking else future

    def get_num_global_epochs(self) -> int:
        with self._access_lock:
            return self._num_global_epochs


class SerialAsyncHarness:
    """AsyncScheduler.schedule (async_scheduler.py:20-42): every arrival is aggregated at once
    under the scheduler's lock and its caller gets the new global model."""

    def __init__(self, aggregator: Any):
        self.aggregator = aggregator
        self._num_global_epochs = 0
        self._access_lock = threading.Lock()

    def get_parameters(self, **kwargs):
        return self.aggregator.get_parameters(**kwargs)

    def global_update(self, client_id, local_model: Dict, **kwargs):
        with self._access_lock:
            global_model = self.aggregator.aggregate(client_id, local_model, **kwargs)
            self._num_global_epochs += 1
        return global_model

    def get_num_global_epochs(self) -> int:
        with self._access_lock:
            return self._num_global_epochs
