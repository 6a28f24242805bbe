"""World-size-2 rehearsal of the parameter-sharded aggregator on CPU (gloo): the sharding,
padding and collective logic, with the C oracle standing in for the HIP kernels."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle as OC
from oracle import synth

HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
N, K, ROUNDS = 10007, 5, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(r):
    base = synth.fill(N, 7, scale=0.05)
    xs = [synth.fill(N, 100 * r + k, scale=0.02, base=base) for k in range(K)]
    w = [(k + 1) / sum(range(1, K + 1)) for k in range(K)]
    return base, xs, w


def _single_process_reference():
    base, _, _ = _inputs(0)
    g, m, v = base.copy(), np.zeros(N, np.float32), np.full(N, np.float32(HP["tau"] ** 2))
    for r in range(ROUNDS):
        _, xs, w = _inputs(r)
        OC.reduce_opt(xs, w, g, m, v, OC.OPT_ADAM, **HP)
    return g


def _worker(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        from appfl_amd.sharded import ShardedFlatAggregator, shard_bounds
        agg = ShardedFlatAggregator(N, "adam", device="cpu", backend=cpu_backend, **HP)
        base, _, _ = _inputs(0)
        agg.load_global(torch.from_numpy(base))
        lo, hi = shard_bounds(N, world, rank)
        assert (lo, hi) == (agg.lo, agg.hi)
        for r in range(ROUNDS):
            _, xs, w = _inputs(r)
            if mode == "sliced":
                agg.round_sliced([torch.from_numpy(x[lo:hi].copy()) for x in xs], w)
            elif mode == "pipelined":
                # slices padded to the shard length like rows of the staging slab
                rows = [torch.from_numpy(np.concatenate([x[lo:hi], np.zeros(agg.L - (hi - lo), np.float32)]))
                        for x in xs]
                full = agg.round_sliced_gather(rows, w, pieces=3)
                assert torch.equal(full, agg.all_gather_global())
            else:
                mine = [k for k in range(K) if k % world == rank]
                agg.round_whole([torch.from_numpy(xs[k]) for k in mine], [w[k] for k in mine])
        full = agg.all_gather_global()
        if rank == 0:
            np.save(out, full.numpy()[:N])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("sliced", 2), ("pipelined", 2), ("whole", 2),
                                        ("pipelined", 3), ("whole", 3)])
def test_multi_rank_sharded_adam(mode, world, tmp_path):
    out = str(tmp_path / "g.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    got, want = np.load(out), _single_process_reference()
    if mode in ("sliced", "pipelined"):
        # every element sees the same operations in the same order as on one GPU
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        # the cross-rank sum reorders the fp32 additions of the averaged step
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))


def test_shard_bounds_cover_the_vector():
    from appfl_amd.sharded import shard_bounds, shard_length
    for n in (1, 3, 4, 5, 1023, 10007, 25557032):
        for world in (1, 2, 4, 8):
            L = shard_length(n, world)
            assert L % 4 == 0 and L * world >= n
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d


def _sd_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="sample_size")
        agg = ShardedFedAvgAggregator(S.ToyModel(), cfg, None, algorithm="FedAdam", device="cpu",
                                      backend=cpu_backend)
        for c in range(5):
            agg.set_client_sample_size(c, 4 + 3 * c)
        recs = []
        for r, K in enumerate([2, 5, 3]):
            local = {c: S.make_state(S.TOY, seed=700 + 10 * r + c) for c in range(K)}
            if r == 1:
                del local[2]["fc.weight"]            # a client lacking a weighted entry
            recs.append({k: v.numpy().copy() for k, v in agg.aggregate(local).items()})
        if rank == 1:                                   # every rank holds the whole result
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_two_rank_state_dict_aggregator_matches_single_process_oracle(tmp_path):
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_sd_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out, allow_pickle=True)
    cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="sample_size")
    ora = OA.OracleFedAdam(S.ToyModel(), cfg, None)
    for c in range(5):
        ora.set_client_sample_size(c, 4 + 3 * c)
    for r, K in enumerate([2, 5, 3]):
        local = {c: S.make_state(S.TOY, seed=700 + 10 * r + c) for c in range(K)}
        if r == 1:
            del local[2]["fc.weight"]
        want = ora.aggregate(local)
        assert list(got[r]) == list(want)
        for k in want:
            assert got[r][k].dtype == want[k].dtype and got[r][k].shape == want[k].shape, k
            assert got[r][k].tobytes() == want[k].tobytes(), (r, k)


TINY = {"w": ((5,), "f32"), "n": ((), "i64")}


def _tiny_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        agg = ShardedFedAvgAggregator(None, {"client_weights_mode": "equal"}, None, algorithm="FedAvg",
                                      device="cpu", backend=cpu_backend)
        recs = []
        for r in range(2):
            local = {c: S.make_state(TINY, seed=40 * r + c) for c in range(3)}
            recs.append({k: v.numpy().copy() for k, v in agg.aggregate(local).items()})
        assert agg._flat.hi - agg._flat.lo == (4, 1, 0)[rank]
        if rank == 2:                                   # the rank whose slice is empty
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_three_ranks_on_a_five_element_model_one_slice_empty(tmp_path):
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_tiny_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out, allow_pickle=True)
    ora = OA.OracleFedAvg(None, {"client_weights_mode": "equal"}, None)
    for r in range(2):
        want = ora.aggregate({c: S.make_state(TINY, seed=40 * r + c) for c in range(3)})
        for k in want:
            assert got[r][k].tobytes() == want[k].tobytes(), (r, k)
