"""Device aggregators under the names `appfl.algorithm.aggregator` uses for the CPU ones."""

from .admm_aggregator import ICEADMMAggregator, IIADMMAggregator
from .base_aggregator import BaseAggregator
from .fedavg_aggregator import (FedAdagradAggregator, FedAdamAggregator, FedAvgAggregator,
                                FedAvgMAggregator, FedYogiAggregator)
from .fedasync_aggregator import FedAsyncAggregator, FedBuffAggregator
from .fedcompass_aggregator import FedCompassAggregator, FedQueueAggregator

__all__ = [
    "BaseAggregator",
    "FedAvgAggregator",
    "FedAvgMAggregator",
    "FedAdamAggregator",
    "FedYogiAggregator",
    "FedAdagradAggregator",
    "FedAsyncAggregator",
    "FedBuffAggregator",
    "FedCompassAggregator",
    "FedQueueAggregator",
    "IIADMMAggregator",
    "ICEADMMAggregator",
]
