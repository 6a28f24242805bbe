import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from appfl_amd import workloads as W
import appfl_amd.aggregator as A
torch.cuda.set_device(0)
wins = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
for w in wins:
    A.FedAvgAggregator.STAGE_WINDOW = w << 20
    for k in ("resnet18_fedavg_k100", "resnet50_fedadam_k64"):
        d = bench.end_to_end(W.WORKLOADS[k], rounds=3, arrivals=7)
        print(w, "MiB", k, d["ms_per_round"], "ms/round, after last arrival", d["after_last_arrival_ms"], "ms", flush=True)
