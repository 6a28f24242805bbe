"""Development probe: many rounds on one aggregator (sync with stage-on-arrival, FedAsync, FedBuff);
prints device memory, pinned-pool size and process RSS every so many rounds -- they must level off."""
import os
import sys

import psutil
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import appfl_amd.aggregator as A  # noqa: E402


def state(n, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return {"w": torch.randn(n, generator=g).to(dtype), "b": torch.randn(1000, generator=g).to(dtype),
            "steps": torch.tensor(seed)}


def report(tag, r):
    rss = psutil.Process().memory_info().rss / 2 ** 20
    print(f"{tag} round {r:5d}: device {torch.cuda.memory_allocated() / 2 ** 20:9.1f} MiB "
          f"(reserved {torch.cuda.memory_reserved() / 2 ** 20:9.1f}), rss {rss:9.1f} MiB", flush=True)


def main():
    n, rounds = 5_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 400
    models = [state(n, s) for s in range(8)]
    cfg = {"server_learning_rate": 0.1, "server_adapt_param": 1e-3, "server_momentum_param_1": 0.9,
           "server_momentum_param_2": 0.99, "client_weights_mode": "equal", "num_clients": 4}
    agg = A.FedAdamAggregator(None, cfg, None)
    for r in range(rounds):
        ids = [(r + i) % 8 for i in range(4)]
        local = {c: models[c] for c in ids}
        for c in ids[:r % 5]:
            agg.stage(c, local[c])
        out = agg.aggregate(local)
        if r % (rounds // 8) == 0:
            report("FedAdam+stage", r)
    del out
    for cls, extra in (("FedAsyncAggregator", {}), ("FedBuffAggregator", {"K": 3})):
        agg = getattr(A, cls)(None, dict(alpha=0.9, num_clients=4, staleness_fn="polynomial",
                                         staleness_fn_kwargs={"a": 0.5}, **extra), None)
        for r in range(rounds):
            out = agg.aggregate(r % 4, models[r % 8])
            if r % 7 == 0:
                agg.get_parameters(client_id=(r + 1) % 4)
            if r % (rounds // 8) == 0:
                report(cls, r)
        del out


if __name__ == "__main__":
    main()
