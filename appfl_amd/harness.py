"""Minimal stand-in for the reference's synchronous scheduling plumbing, enough to drive an
aggregator the way APPFL's serial simulation does (examples/serial/run_serial.py:70-91 and
tests/test_mnist.py:601-624): per-client `global_update(..., blocking=False)` returning a
Future, aggregation when the last client of the round has arrived, one shared result object for
every client's Future (algorithm/scheduler/sync_scheduler.py:25-107).  Not a scheduler
re-implementation: no communicators, no chunk bookkeeping, no virtual time.
"""

from __future__ import annotations

import threading
from concurrent.futures import Future
from typing import Any, Dict, Union


class SerialSyncHarness:
    def __init__(self, aggregator: Any, num_clients: int):
        self.aggregator = aggregator
        self.num_clients = num_clients
        self.local_models: Dict[Union[int, str], Dict] = {}
        self.aggregation_kwargs: Dict[str, Dict] = {}
        self.future: Dict[Union[int, str], Future] = {}
        self._num_global_epochs = 0
        self._access_lock = threading.Lock()

    def set_sample_size(self, client_id, sample_size: int) -> None:
        """ServerAgent.set_sample_size (agent/server.py:146-190), without the sync barrier."""
        self.aggregator.set_client_sample_size(client_id, sample_size)

    def get_parameters(self, **kwargs):
        """BaseScheduler.get_parameters with serial_run=True (base_scheduler.py:38-79)."""
        return self.aggregator.get_parameters(**kwargs)

    def global_update(self, client_id, local_model: Dict, blocking: bool = False, **kwargs):
        """ServerAgent.global_update -> SyncScheduler.schedule (sync_scheduler.py:40-107)."""
        with self._access_lock:
            future: Future = Future()
            self.local_models[client_id] = local_model
            # scheduler-side staging: start this client's upload while the others are awaited
            stage = getattr(self.aggregator, "stage", None)
            if callable(stage):
                stage(client_id, local_model)
            for key, value in kwargs.items():
                self.aggregation_kwargs.setdefault(key, {})[client_id] = value
            self.future[client_id] = future
            if len(self.local_models) == self.num_clients:
                aggregated = self.aggregator.aggregate(self.local_models, **self.aggregation_kwargs)
                futures = dict(self.future)
                self.local_models.clear()
                self.aggregation_kwargs.clear()
                self.future.clear()
                while futures:
                    _, f = futures.popitem()
                    f.set_result(aggregated)          # the SAME object for every client
                self._num_global_epochs += 1
        return future.result() if blocking else future

    def get_num_global_epochs(self) -> int:
        with self._access_lock:
            return self._num_global_epochs
