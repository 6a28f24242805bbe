"""Development tool: cProfile of the host side of the last arrival (stage + aggregate) of a
ResNet-18 x 100 FedAvg round.  python tools/e2e_profile_last_arrival.py [window MiB]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import appfl_amd.aggregator as A  # noqa: E402
from appfl_amd import workloads as W  # noqa: E402

torch.cuda.set_device(0)
win = int(sys.argv[1]) if len(sys.argv) > 1 else 4
A.FedAvgAggregator.STAGE_WINDOW = win << 20
wl = W.WORKLOADS["resnet18_fedavg_k100"]
rows, K = wl["table"](), wl["K"]
clients = {}
for k in range(K):
    g = torch.Generator().manual_seed(k)
    clients[k] = {name: (torch.randn(shape, generator=g) * 0.01 if dtype == torch.float32
                         else torch.full(shape, 3, dtype=dtype)) for name, shape, dtype, _ in rows}
agg = A.FedAvgAggregator(None, dict(W.EXAMPLE_CONFIGS["FedAvgAggregator"], num_clients=K), None)
ids = list(clients)
prof = cProfile.Profile()
for rnd in range(4):
    for c in ids[:-1]:
        agg.stage(c, clients[c])
    torch.cuda.synchronize()
    if rnd == 3:
        prof.enable()
    agg.stage(ids[-1], clients[ids[-1]])
    agg.aggregate(clients)
    if rnd == 3:
        prof.disable()
pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
