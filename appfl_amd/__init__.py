"""MI355X-native server-side aggregation for APPFL (FedAvg family, FedAsync / FedBuff).

``appfl_amd.aggregator`` mirrors ``appfl.algorithm.aggregator``; the arithmetic runs in the HIP
kernels of ``libappfl_amd.so`` (``include/appfl_amd.h``).
"""

__version__ = "0.1.0"
