"""Development tool: time from the last client's arrival to the new model with and without the
earlier clients' terms folded on arrival (FedAvgAggregator.FOLD_ON_ARRIVAL), same process.
python tools/e2e_fold_on_arrival.py [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import appfl_amd.aggregator as A  # noqa: E402
import bench  # noqa: E402
from appfl_amd import workloads as W  # noqa: E402

torch.cuda.set_device(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rep in range(reps):
    for on in (True, False):
        A.FedAvgAggregator.FOLD_ON_ARRIVAL = on
        for k in ("resnet18_fedavg_k100", "resnet50_fedadam_k64"):
            d = bench.end_to_end(W.WORKLOADS[k], rounds=2, arrivals=9)
            print(f"fold_on_arrival={on} {k}: after last arrival {d['after_last_arrival_ms']} ms",
                  flush=True)
