"""Development tool: what the time after the LAST arrival of a wire-form round is made of
(ResNet-18 x K payloads): parse, pack + upload, aggregate, serialisation of the reply."""
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import appfl_amd.aggregator as A  # noqa: E402
from appfl_amd import wire, workloads as W  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rows = W.resnet(18)
    payloads = []
    for k in range(K):
        state = {n: (torch.randn(s) * 0.01 if d == torch.float32 else torch.full(s, 3, dtype=d)) for n, s, d, _ in rows}
        b = io.BytesIO()
        torch.save(state, b)
        payloads.append(b.getvalue())
    agg = A.FedAvgAggregator(None, {"num_clients": K}, None)
    for rnd in range(4):
        models = {}
        for k, p in enumerate(payloads[:-1]):
            models[k] = agg.stage_bytes(k, p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = wire.bytes_to_model(payloads[-1])
        t1 = time.perf_counter()
        agg.stage(K - 1, m)
        models[K - 1] = m
        t2 = time.perf_counter()
        out = agg.aggregate(models)
        t3 = time.perf_counter()
        reply = wire.model_to_bytes(out)
        t4 = time.perf_counter()
        b = io.BytesIO()
        torch.save(out, b)
        t5 = time.perf_counter()
        print(f"round {rnd}: parse {1e3 * (t1 - t0):.2f} ms, stage {1e3 * (t2 - t1):.2f}, aggregate {1e3 * (t3 - t2):.2f}, "
              f"model_to_bytes {1e3 * (t4 - t3):.2f} ({len(reply)} B), torch.save for comparison {1e3 * (t5 - t4):.2f}", flush=True)


if __name__ == "__main__":
    main()
