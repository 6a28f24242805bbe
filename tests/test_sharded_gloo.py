"""Multi-rank rehearsals of the parameter-sharded aggregators on CPU (gloo, world size 2, 3 and 8):
the block map, padding and collective logic, with the C oracle standing in for the HIP kernels."""

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


def _worker(rank, world, port, mode, pieces, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        from appfl_amd.sharded import ShardedFlatAggregator
        agg = ShardedFlatAggregator(N, "adam", device="cpu", backend=cpu_backend, pieces=pieces, **HP)
        base, _, _ = _inputs(0)
        agg.load_global(torch.from_numpy(base))
        smap = agg.map
        assert smap.pieces == pieces and smap.L % 4 == 0 and smap.P >= N
        for r in range(ROUNDS):
            _, xs, w = _inputs(r)
            if mode == "sliced":
                agg.round_sliced([smap.take(torch.from_numpy(x)) for x in xs], w)
            elif mode == "pipelined":
                full = agg.round_sliced_gather([smap.take(torch.from_numpy(x)) for x in xs], w)
                assert torch.equal(full, agg.all_gather_global())
            else:
                mine = [k for k in range(K) if k % world == rank]
                agg.round_whole([torch.from_numpy(xs[k]) for k in mine], [w[k] for k in mine])
                assert torch.equal(agg.g_full, agg.all_gather_global())
        full = agg.all_gather_global()
        assert not full[N:].any()                   # the padding of the gathered vector stays zero
        if rank == 0:
            np.save(out, full.numpy()[:N])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world,pieces", [("sliced", 2, 1), ("pipelined", 2, 3), ("whole", 2, 2),
                                               ("pipelined", 3, 4), ("whole", 3, 1),
                                               ("sliced", 3, 5), ("pipelined", 8, 2), ("whole", 8, 1)])
def test_multi_rank_sharded_adam(mode, world, pieces, tmp_path):
    out = str(tmp_path / "g.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, pieces, out), nprocs=world, join=True)
    got, want = np.load(out), _single_process_reference()
    if mode in ("sliced", "pipelined"):
        # every element sees the same operations in the same order as on one GPU
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        # the cross-rank sum reorders the fp32 additions of the averaged step
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))


def test_shard_map_covers_the_vector_exactly_once():
    from appfl_amd.sharded import ShardMap, default_pieces
    for n in (0, 1, 3, 4, 5, 1023, 10007, 25557032):
        for world in (1, 2, 3, 8):
            for pieces in (1, 2, 4, 7):
                maps = [ShardMap(n, world, r, pieces) for r in range(world)]
                m0 = maps[0]
                assert m0.per % 4 == 0 and m0.L == pieces * m0.per and m0.P == world * m0.L >= n
                if n > 1024 * world * pieces:       # blocks of a real model start on 256-byte boundaries
                    assert m0.per % 64 == 0
                spans = sorted((lo, hi) for m in maps for lo, hi, _ in m.windows())
                pos = 0
                for lo, hi in spans:
                    assert lo == pos and hi > lo
                    pos = hi
                assert pos == n
                for m in maps:
                    wins = m.windows()
                    # live elements are a prefix of the local storage: full blocks, then at most
                    # one partial block, then nothing
                    assert [off for _, _, off in wins] == [p * m.per for p in range(len(wins))]
                    assert all(hi - lo == m.per for lo, hi, _ in wins[:-1])
                    assert m.n_local == sum(hi - lo for lo, hi, _ in wins)
                    # block (p, r) sits where the in-place gather of row p puts it
                    for p in range(pieces):
                        a, _ = m.row(p)
                        lo, _ = m.block(p)
                        assert lo == min(n, a + m.rank * m.per)
    assert default_pieces(1_300_000_000, 8) == 4 and default_pieces(25_557_032, 8) == 1
    assert default_pieces(1_300_000_000, 2) == 8


def _sd_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="sample_size")
        agg = ShardedFedAvgAggregator(S.ToyModel(), cfg, None, algorithm="FedAdam", device="cpu",
                                      backend=cpu_backend, pieces=3)
        for c in range(5):
            agg.set_client_sample_size(c, 4 + 3 * c)
        recs = []
        if rank == 0:
            # before the first round only the serving rank answers a client: must not hang
            recs.append({k: v.numpy().copy() for k, v in agg.get_parameters().items()})
        for r, K in enumerate([2, 5, 3]):
            local = {c: S.make_state(S.TOY, seed=700 + 10 * r + c) for c in range(K)}
            if r == 1:
                del local[2]["fc.weight"]            # a client lacking a weighted entry
            if r == 2:
                for c, m in local.items():          # staged on arrival, as a scheduler would
                    agg.stage(c, m)
            out_r = agg.aggregate(local)
            if rank == 0:
                recs.append({k: v.numpy().copy() for k, v in out_r.items()})
                # between rounds ONLY rank 0 serves get_parameters: local, no collective
                again = agg.get_parameters()
                for k in out_r:
                    assert again[k].numpy().tobytes() == out_r[k].numpy().tobytes(), k
            elif rank == 1 and r == 2:
                chunk, meta = None, None
                agg.set_model_chunk_size(600)
                chunk, meta = agg.get_parameters(_request_chunked=True)
                assert meta["_total_chunks"] == 2 and list(chunk) == meta["_chunk_keys"]
        if rank == 0:
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
    wants = [ora.get_parameters()]
    for r, K in enumerate([2, 5, 3]):
        local = {c: S.make_state(S.TOY, seed=700 + 10 * r + c) for c in range(K)}
        if r == 1:
            del local[2]["fc.weight"]
        wants.append(ora.aggregate(local))
    assert len(got) == len(wants)
    for i, want in enumerate(wants):
        assert list(got[i]) == list(want)
        for k in want:
            w = want[k].numpy() if isinstance(want[k], torch.Tensor) else want[k]
            assert got[i][k].dtype == w.dtype and got[i][k].shape == w.shape, k
            assert got[i][k].tobytes() == w.tobytes(), (i, k)


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
        assert agg._map.n_local == (4, 1, 0)[rank]
        if rank == 2:                                   # the rank whose blocks are all empty
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_three_ranks_on_a_five_element_model_one_rank_empty(tmp_path):
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


def _f64_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        cfg = dict(S.HP["FedYogiAggregator"], client_weights_mode="equal")
        agg = ShardedFedAvgAggregator(S.Mixed64Model(), cfg, None, algorithm="FedYogi", device="cpu",
                                      backend=cpu_backend, pieces=2)
        recs = []
        for r, K in enumerate([3, 2]):
            local = {c: S.make_state(S.MIXED64, seed=900 + 10 * r + c) for c in range(K)}
            recs.append({k: v.numpy().copy() for k, v in agg.aggregate(local).items()})
        if rank == 1:
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_float64_entries_kept_whole_on_every_rank(tmp_path):
    """float64 parameters and buffers beside the sharded fp32 ones: every rank steps them whole,
    with moments of its own; bit-identical to the single-process oracle."""
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_f64_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out, allow_pickle=True)
    cfg = dict(S.HP["FedYogiAggregator"], client_weights_mode="equal")
    ora = OA.OracleFedYogi(S.Mixed64Model(), cfg, None)
    for r, K in enumerate([3, 2]):
        want = ora.aggregate({c: S.make_state(S.MIXED64, seed=900 + 10 * r + c) for c in range(K)})
        assert list(got[r]) == list(want)
        for k in want:
            w = want[k].numpy() if isinstance(want[k], torch.Tensor) else want[k]
            assert got[r][k].dtype == w.dtype and got[r][k].tobytes() == w.tobytes(), (r, k)


# -- whole-update mode at the state-dict level ----------------------------------------------------

def _close(got, want, rtol):
    """Within rtol of the tensor's scale where the expected value is finite; the same inf / nan
    elsewhere (the toy states hold values whose squares overflow fp32)."""
    fin = np.isfinite(want)
    if not np.array_equal(np.isfinite(got), fin):
        return False
    if not np.array_equal(np.isnan(got), np.isnan(want)) or not np.array_equal(got[np.isinf(want)], want[np.isinf(want)]):
        return False
    # the toy states also hold a few finite values around 1e28-1e30 whose terms nearly cancel: they
    # are compared on their own (1e-4 of their magnitude), so that they do not set the scale
    big = fin & (np.abs(want) > 1e10)
    if big.any() and not np.all(np.abs(got[big] - want[big]) <= 1e-4 * np.abs(want[big])):
        return False
    fin = fin & ~big
    if not fin.any():
        return True
    scale = max(float(np.max(np.abs(want[fin]))), 1e-30)
    return float(np.max(np.abs(got[fin] - want[fin]))) <= rtol * scale


WHOLE_ROUNDS = [2, 5, 3, 4]


def _whole_clients(r, K):
    import scenarios as S
    local = {c: S.make_state(S.TOY, seed=1700 + 10 * r + c) for c in range(K)}
    if r == 1:
        del local[2]["fc.weight"]            # a client lacking a weighted entry
    return local


def _whole_worker(rank, world, port, alg, with_model, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode="sample_size")
        agg = ShardedFedAvgAggregator(S.ToyModel() if with_model else None, cfg, None, algorithm=alg,
                                      device="cpu", backend=cpu_backend, pieces=2, mode="whole")
        for c in range(5):
            if c != 3:                              # one client without a recorded sample size
                agg.set_client_sample_size(c, 4 + 3 * c)
        recs = []
        for r, K in enumerate(WHOLE_ROUNDS):
            local = _whole_clients(r, K)
            mine = {c: local[c] for c in local if c % world == rank}     # client k -> rank k mod R
            if r in (0, 2):
                for c, m in mine.items():           # staged on arrival (round 0: before the state exists)
                    agg.stage(c, m)
            out_r = agg.aggregate(mine)
            again = agg.get_parameters()            # local, no collective
            for k in out_r:
                assert again[k].numpy().tobytes() == out_r[k].numpy().tobytes(), k
            recs.append({k: v.numpy().copy() for k, v in out_r.items()})
            if r == 3:
                m, v = agg.gather_moments()         # collective accessor of the sharded moments
                assert (agg.m_vector is m or not m) and (agg.v_vector is v or not v)
                recs.append({"m/" + k: t.numpy().copy() for k, t in m.items()})
                recs.append({"v/" + k: t.numpy().copy() for k, t in v.items()})
        if rank == world - 1:
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,alg,with_model", [(2, "FedAdam", True), (3, "FedAvg", False),
                                                  (3, "FedYogi", True), (2, "FedAvgM", False)])
def test_whole_update_mode_of_the_state_dict_aggregator(world, alg, with_model, tmp_path):
    """Rank r gets only the clients k = r (mod R) of every round (rank 2 of 3 gets none in the
    2-client round): weighted fp32 entries within 1e-6 of the tensor's scale of the
    single-process oracle (the reduce-scatter reorders the sum), everything else bit-exact."""
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_whole_worker, args=(world, _free_port(), alg, with_model, out), nprocs=world, join=True)
    got = list(np.load(out, allow_pickle=True))
    cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode="sample_size")
    model = S.ToyModel() if with_model else None
    ora = OA.ORACLES[alg + "Aggregator"](model, cfg, None)
    for c in range(5):
        if c != 3:
            ora.set_client_sample_size(c, 4 + 3 * c)
    params = None if model is None else {n for n, _ in model.named_parameters()}
    i = 0
    for r, K in enumerate(WHOLE_ROUNDS):
        want = ora.aggregate(_whole_clients(r, K))
        rec = got[i]
        i += 1
        assert list(rec) == list(want)
        for k in want:
            w = np.asarray(want[k])
            assert rec[k].dtype == w.dtype and rec[k].shape == w.shape, k
            weighted = w.dtype == np.float32 and (params is None or k in params)
            if weighted:
                assert _close(rec[k], w, 1e-6), (r, k)
            else:
                assert rec[k].tobytes() == w.tobytes(), (r, k)
    if alg != "FedAvg":
        moments = [getattr(ora, "m_vector", {}), ora.v_vector]
        for tag, want in zip("mv", moments):
            rec = got[i]
            i += 1
            assert sorted(rec) == sorted(tag + "/" + k for k in want)
            for k in want:
                assert _close(rec[tag + "/" + k], want[k], 2e-6), (tag, k)


def _holes_worker(rank, world, port, out):
    """The reference-visible behaviour beside the fused round, sliced mode: compute_steps,
    m_vector / v_vector, a subclass overriding compute_steps, streamed rounds, flat payloads."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        recs = {}
        cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="equal")
        agg = ShardedFedAvgAggregator(S.ToyModel(), cfg, None, algorithm="FedAdam", device="cpu",
                                      backend=cpu_backend, pieces=2)
        local = {c: S.make_state(S.TOY, seed=2100 + c) for c in range(3)}
        recs["round0"] = {k: v.numpy().copy() for k, v in agg.aggregate(local).items()}
        if world > 1:
            with pytest.raises(RuntimeError):
                agg.m_vector                        # sharded: needs the collective accessor first
        m, v = agg.gather_moments()
        recs["m"] = {k: t.numpy().copy() for k, t in m.items()}
        recs["v"] = {k: t.numpy().copy() for k, t in agg.v_vector.items()}
        local = {c: S.make_state(S.TOY, seed=2200 + c) for c in range(4)}
        agg.compute_steps(local)                    # collective; leaves the step, moments updated
        recs["step"] = {k: t.numpy().copy() for k, t in agg.step.items()}

        class Halved(ShardedFedAvgAggregator):      # a user subclass post-processing the step
            def compute_steps(self, local_models):
                super().compute_steps(local_models)
                for t in self.step.values():
                    t.mul_(0.5)
        cfg2 = dict(S.HP["FedAvgMAggregator"])
        sub = Halved(None, cfg2, None, algorithm="FedAvgM", device="cpu", backend=cpu_backend, pieces=3)
        for r in range(2):
            local = {c: S.make_state(S.TOY, seed=2300 + 10 * r + c) for c in range(3)}
            recs[f"halved{r}"] = {k: v.numpy().copy() for k, v in sub.aggregate(local).items()}

        # streamed rounds: chunk by chunk, then a whole round on the same state
        st = ShardedFedAvgAggregator(None, {}, None, device="cpu", backend=cpu_backend, pieces=2)
        chunks = S.chunk_keys_by_size(S.TOY, 600)
        for r in range(2):
            states = {c: S.make_state(S.TOY, seed=2400 + 10 * r + c) for c in range(3)}
            if r == 1:
                recs["streamed_whole"] = {k: v.numpy().copy() for k, v in st.aggregate(
                    {c: S.make_state(S.TOY, seed=2450 + c) for c in range(2)}).items()}
            for ci, keys in enumerate(chunks):
                o = st.aggregate({c: {k: states[c][k].clone() for k in keys} for c in states},
                                 _chunk_idx={c: ci for c in states},
                                 _total_chunks={c: len(chunks) for c in states},
                                 _chunk_keys={c: list(keys) for c in states})
                recs[f"streamed{r}.{ci}"] = {k: v.numpy().copy() for k, v in o.items()}
        recs["streamed_final"] = {k: v.numpy().copy() for k, v in st.get_parameters().items()}

        # flat secure-aggregation payloads: replicated on every rank
        sec = ShardedFedAvgAggregator(None, {"use_secure_agg": True, "client_weights_mode": "sample_size"},
                                      None, device="cpu", backend=cpu_backend)
        names = [n for n, (_, kind) in S.TOY.items() if kind == "f32"]
        for r in range(2):
            payload = {}
            for c in range(3):
                sd = S.make_state(S.TOY, seed=2500 + 10 * r + c, gradient=True, scale=0.01)
                payload[c] = {"type": "masked_update_flat",
                              "flat": torch.cat([sd[n].reshape(-1) for n in names]),
                              "shapes": [(n, tuple(sd[n].shape)) for n in names], "num_examples": 5 + c}
            recs[f"secure{r}"] = {k: v.numpy().copy() for k, v in sec.aggregate(payload).items()}
        if rank == world - 1:
            np.save(out, np.array([recs], dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_class_keeps_the_reference_visible_behaviour(world, tmp_path):
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_holes_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out, allow_pickle=True)[0]

    def same(rec, want, what):
        assert list(rec) == list(want), what
        for k in want:
            w = np.asarray(want[k])
            assert rec[k].dtype == w.dtype and rec[k].tobytes() == w.tobytes(), (what, k)

    cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="equal")
    ora = OA.OracleFedAdam(S.ToyModel(), cfg, None)
    same(got["round0"], ora.aggregate({c: S.make_state(S.TOY, seed=2100 + c) for c in range(3)}), "round0")
    same(got["m"], ora.m_vector, "m")
    same(got["v"], ora.v_vector, "v")
    ora.compute_steps({c: {k: v.numpy() for k, v in S.make_state(S.TOY, seed=2200 + c).items()}
                       for c in range(4)})
    same(got["step"], ora.step, "step")

    class HalvedOracle(OA.OracleFedAvgM):
        def compute_steps(self, local_models):
            super().compute_steps(local_models)
            for k in self.step:
                self.step[k] = self.step[k] * np.float32(0.5)
    sub = HalvedOracle(None, dict(S.HP["FedAvgMAggregator"]), None)
    for r in range(2):
        same(got[f"halved{r}"], sub.aggregate({c: S.make_state(S.TOY, seed=2300 + 10 * r + c)
                                               for c in range(3)}), f"halved{r}")

    st = OA.OracleFedAvg(None, {}, None)
    chunks = S.chunk_keys_by_size(S.TOY, 600)
    for r in range(2):
        states = {c: S.make_state(S.TOY, seed=2400 + 10 * r + c) for c in range(3)}
        if r == 1:
            same(got["streamed_whole"], st.aggregate({c: S.make_state(S.TOY, seed=2450 + c)
                                                      for c in range(2)}), "streamed_whole")
        for ci, keys in enumerate(chunks):
            o = st.aggregate({c: {k: states[c][k].clone() for k in keys} for c in states},
                             _chunk_idx={c: ci for c in states},
                             _total_chunks={c: len(chunks) for c in states},
                             _chunk_keys={c: list(keys) for c in states})
            same(got[f"streamed{r}.{ci}"], o, f"streamed{r}.{ci}")
    same(got["streamed_final"], st.get_parameters(), "streamed_final")

    sec = OA.OracleFedAvg(None, {"use_secure_agg": True, "client_weights_mode": "sample_size"}, None)
    names = [n for n, (_, kind) in S.TOY.items() if kind == "f32"]
    for r in range(2):
        payload = {}
        for c in range(3):
            sd = S.make_state(S.TOY, seed=2500 + 10 * r + c, gradient=True, scale=0.01)
            payload[c] = {"type": "masked_update_flat",
                          "flat": torch.cat([sd[n].reshape(-1) for n in names]),
                          "shapes": [(n, tuple(sd[n].shape)) for n in names], "num_examples": 5 + c}
        same(got[f"secure{r}"], sec.aggregate(payload), f"secure{r}")


# -- bfloat16 / float16 models: sliced rounds on 16-bit blocks ---------------------------------------

HALF_ROUNDS = [2, 5, 3, 4]


def _half_clients(spec, r, K):
    import scenarios as S
    local = {c: S.make_state(spec, seed=3100 + 10 * r + c) for c in range(K)}
    if r == 1:
        del local[2][next(n for n, (shp, _) in spec.items() if len(shp) == 2)]   # lacks a weighted entry
    return local


def _half_worker(rank, world, port, kind, alg, spec_name, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        spec = S.SPECS[f"{spec_name}_{kind}"]
        model = S.MODELS[f"toy_{kind}"]() if spec_name == "toy" else None
        cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode="sample_size")
        agg = ShardedFedAvgAggregator(model, cfg, None, algorithm=alg, device="cpu", backend=cpu_backend,
                                      pieces=3)
        for c in range(5):
            agg.set_client_sample_size(c, 4 + 3 * c)
        recs = []
        for r, K in enumerate(HALF_ROUNDS):
            local = _half_clients(spec, r, K)
            if r == 2:
                for c, m in local.items():          # staged on arrival
                    agg.stage(c, m)
            if r == 3:
                agg.compute_steps(local)            # collective; the step as 16-bit tensors
                assert all(t.dtype == S.HALF_KINDS[kind] for t in agg.step.values())
                recs.append(S._snap(agg.step))
                m, v = agg.gather_moments()
                recs.append(S._snap(m))
                recs.append(S._snap(v))
                break
            out_r = agg.aggregate(local)
            assert agg._map.itemsize == 2 and agg._gathered.dtype == S.HALF_KINDS[kind]
            assert all(out_r[n].dtype == (S.HALF_KINDS[kind] if k == kind else out_r[n].dtype)
                       for n, (_, k) in spec.items())
            recs.append(S._snap(out_r))
            again = agg.get_parameters()            # local, no collective
            assert all(torch.equal(again[k].reshape(-1).view(torch.uint8), out_r[k].reshape(-1).view(torch.uint8)) for k in out_r)
        if rank == world - 1:
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,alg,spec_name,world", [("bf16", "FedAdam", "toy", 2), ("f16", "FedYogi", "ragged", 3),
                                                      ("bf16", "FedAvgM", "ragged", 2), ("f16", "FedAvg", "toy", 2)])
def test_sliced_rounds_of_a_16_bit_model_match_the_single_process_oracle(kind, alg, spec_name, world, tmp_path):
    """Every rank holds its blocks of the bfloat16 / float16 rows (2-byte elements in the slab, the
    state, the moments and the gathered model); results, steps and moments bit for bit those of
    the oracle's 16-bit arithmetic."""
    import golden_util
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_half_worker, args=(world, _free_port(), kind, alg, spec_name, out), nprocs=world, join=True)
    got = list(np.load(out, allow_pickle=True))
    spec = S.SPECS[f"{spec_name}_{kind}"]
    model = S.MODELS[f"toy_{kind}"]() if spec_name == "toy" else None
    cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode="sample_size")
    ora = OA.ORACLES[alg + "Aggregator"](model, cfg, None)
    for c in range(5):
        ora.set_client_sample_size(c, 4 + 3 * c)
    wants = []
    for r, K in enumerate(HALF_ROUNDS):
        local = _half_clients(spec, r, K)
        if r == 3:
            ora.compute_steps({c: {k: OA.to_np(v) for k, v in m.items()} for c, m in local.items()})
            wants += [S._snap(ora.step), S._snap(getattr(ora, "m_vector", {})),
                      S._snap(getattr(ora, "v_vector", {}))]
            break
        wants.append(S._snap(ora.aggregate(local)))
    assert len(got) == len(wants)
    for i, (rec, want) in enumerate(zip(got, wants)):
        assert sorted(rec) == sorted(want), i
        for k in want:
            assert golden_util.same_bits(rec[k], want[k]), (i, k)


def _half_flat_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        from appfl_amd.sharded import ShardedFlatAggregator
        agg = ShardedFlatAggregator(N, "adam", device="cpu", backend=cpu_backend, pieces=3,
                                    dtype=torch.bfloat16, **HP)
        assert agg.map.itemsize == 2 and agg.map.per % 8 == 0 and agg.g.dtype == torch.bfloat16
        base, _, _ = _inputs(0)
        agg.load_global(torch.from_numpy(base).bfloat16())
        for r in range(ROUNDS):
            _, xs, w = _inputs(r)
            full = agg.round_sliced_gather([agg.map.take(torch.from_numpy(x).bfloat16()) for x in xs], w)
            assert full.dtype == torch.bfloat16 and torch.equal(full, agg.all_gather_global())
        with pytest.raises(NotImplementedError, match="whole-update"):
            agg.round_whole([], [])
        if rank == 0:
            np.save(out, full[:N].view(torch.int16).numpy())
    finally:
        dist.destroy_process_group()


def test_flat_sliced_rounds_on_bfloat16_vectors(tmp_path):
    import cpu_backend
    out = str(tmp_path / "g.npy")
    mp.spawn(_half_flat_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ops16 = cpu_backend.for_half(torch.bfloat16)
    base, _, _ = _inputs(0)
    g = torch.from_numpy(base).bfloat16()
    m, v = torch.zeros_like(g), torch.zeros_like(g)
    ops16.fill(v, HP["tau"] ** 2)
    for r in range(ROUNDS):
        _, xs, w = _inputs(r)
        ops16.reduce_opt([torch.from_numpy(x).bfloat16() for x in xs], w, g, m, v, "adam", **HP)
    assert np.array_equal(np.load(out), g.view(torch.int16).numpy())


def _half_whole_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        agg = ShardedFedAvgAggregator(None, {}, None, device="cpu", backend=cpu_backend, mode="whole")
        local = {c: S.make_state(S.SPECS["toy_bf16"], seed=c) for c in range(4) if c % world == rank}
        with pytest.raises(NotImplementedError, match="whole-update"):     # on every rank, after the
            agg.aggregate(local)                                           # collective set-up: nobody hangs
    finally:
        dist.destroy_process_group()


def test_whole_mode_refuses_a_16_bit_model_on_every_rank():
    mp.spawn(_half_whole_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_shard_map_of_two_byte_elements():
    from appfl_amd.sharded import ShardMap, default_pieces
    for n in (0, 1, 7, 9, 1023, 10007, 25557032):
        for world in (1, 2, 3, 8):
            maps = [ShardMap(n, world, r, 2, itemsize=2) for r in range(world)]
            m0 = maps[0]
            assert m0.per % 8 == 0 and m0.P >= n            # blocks start on 16-byte boundaries
            if n > 1024 * world * 2:
                assert m0.per % 128 == 0                    # ... of a real model on 256-byte ones
            spans = sorted((lo, hi) for m in maps for lo, hi, _ in m.windows())
            assert sum(hi - lo for lo, hi in spans) == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert default_pieces(1_300_000_000, 8, 2) == 2 and default_pieces(1_300_000_000, 8) == 4
    with pytest.raises(ValueError):
        ShardMap(10, 1, 0, itemsize=8)


def _half_streamed_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        spec = S.SPECS["toy_bf16"]
        st = ShardedFedAvgAggregator(None, {}, None, device="cpu", backend=cpu_backend, pieces=2)
        chunks = S.chunk_keys_by_size(spec, 300)
        recs = []
        for r in range(2):
            states = {c: S.make_state(spec, seed=5400 + 10 * r + c) for c in range(3)}
            if r == 1:      # a whole round between the streamed ones, on the same state
                recs.append(S._snap(st.aggregate({c: S.make_state(spec, seed=5450 + c) for c in range(2)})))
            for ci, keys in enumerate(chunks):
                o = st.aggregate({c: {k: states[c][k].clone() for k in keys} for c in states},
                                 _chunk_idx={c: ci for c in states},
                                 _total_chunks={c: len(chunks) for c in states},
                                 _chunk_keys={c: list(keys) for c in states})
                recs.append(S._snap(o))
        recs.append(S._snap(st.get_parameters()))
        if rank == world - 1:
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_streamed_rounds_of_a_16_bit_model_on_two_ranks(tmp_path):
    import golden_util
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_half_streamed_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = list(np.load(out, allow_pickle=True))
    spec = S.SPECS["toy_bf16"]
    ora = OA.OracleFedAvg(None, {}, None)
    chunks = S.chunk_keys_by_size(spec, 300)
    assert len(chunks) > 1
    wants = []
    for r in range(2):
        states = {c: S.make_state(spec, seed=5400 + 10 * r + c) for c in range(3)}
        if r == 1:
            wants.append(S._snap(ora.aggregate({c: S.make_state(spec, seed=5450 + c) for c in range(2)})))
        for ci, keys in enumerate(chunks):
            wants.append(S._snap(ora.aggregate({c: {k: states[c][k].clone() for k in keys} for c in states},
                                               _chunk_idx={c: ci for c in states},
                                               _total_chunks={c: len(chunks) for c in states},
                                               _chunk_keys={c: list(keys) for c in states})))
    wants.append(S._snap(ora.get_parameters()))
    assert len(got) == len(wants)
    for i, (rec, want) in enumerate(zip(got, wants)):
        assert list(rec) == list(want), i
        for k in want:
            assert golden_util.same_bits(rec[k], want[k]), (i, k)


# -- random rounds on several ranks --------------------------------------------------------------------

def _random_plan(seed):
    """What every rank and the checker derive alike from the seed."""
    rng = np.random.default_rng(40 + seed)
    kind = ["f32", "bf16", "f16"][int(rng.integers(0, 3))]
    alg = ["FedAvg", "FedAvgM", "FedAdam", "FedYogi", "FedAdagrad"][int(rng.integers(0, 5))]
    n = int(rng.choice([3, 61, 1003, 20_011]))
    spec = {"w": ((n,), kind), "steps": ((3,), "i64"), "b": ((7, 3), kind), "c32": ((5,), "i32"),
            "v": ((n // 3 + 1,), kind)}
    pieces = int(rng.choice([1, 2, 5]))
    mode = str(rng.choice(["sample_size", "equal"]))
    rounds = []
    for r in range(4):
        K = int(rng.integers(1, 6))
        ids = [int(c) for c in rng.permutation(6)[:K]]
        rounds.append((ids, int(rng.integers(0, K + 1)), bool(K > 1 and r and rng.random() < 0.3)))
    return kind, alg, spec, pieces, mode, rounds


def _random_clients(spec, seed, r, ids, drop):
    import scenarios as S
    models = {c: S.make_state(spec, seed=10_000 * seed + 100 * r + c, huge=False) for c in ids}
    if drop:
        del models[ids[-1]]["v"]
    return models


def _random_worker(rank, world, port, seed, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        kind, alg, spec, pieces, mode, rounds = _random_plan(seed)
        cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode=mode)
        agg = ShardedFedAvgAggregator(None, cfg, None, algorithm=alg, device="cpu", backend=cpu_backend,
                                      pieces=pieces)
        for c in range(6):
            agg.set_client_sample_size(c, 3 + 2 * c)
        recs = []
        for r, (ids, n_staged, drop) in enumerate(rounds):
            models = _random_clients(spec, seed, r, ids, drop)
            for c in ids[:n_staged]:
                agg.stage(c, models[c])
            recs.append(S._snap(agg.aggregate(models)))
        if rank == world - 1:
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed,world", [(0, 2), (1, 3), (2, 2), (3, 3)])
def test_random_sliced_rounds_on_several_ranks(seed, world, tmp_path):
    import golden_util
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_random_worker, args=(world, _free_port(), seed, out), nprocs=world, join=True)
    got = list(np.load(out, allow_pickle=True))
    kind, alg, spec, pieces, mode, rounds = _random_plan(seed)
    ora = OA.ORACLES[alg + "Aggregator"](None, dict(S.HP[alg + "Aggregator"], client_weights_mode=mode), None)
    for c in range(6):
        ora.set_client_sample_size(c, 3 + 2 * c)
    for r, (ids, n_staged, drop) in enumerate(rounds):
        want = S._snap(ora.aggregate(_random_clients(spec, seed, r, ids, drop)))
        assert list(got[r]) == list(want), (seed, r)
        for k in want:
            assert golden_util.same_bits(got[r][k], want[k]), (seed, kind, alg, pieces, r, k)


def _random_whole_plan(seed):
    rng = np.random.default_rng(140 + seed)
    alg = ["FedAvg", "FedAvgM", "FedAdam", "FedYogi"][int(rng.integers(0, 4))]
    n = int(rng.choice([3, 61, 1003, 20_011]))
    spec = {"w": ((n,), "f32"), "steps": ((3,), "i64"), "b": ((7, 3), "f32"), "c32": ((5,), "i32"),
            "v": ((n // 3 + 1,), "f32"), "d": ((4,), "f64")}
    pieces = int(rng.choice([1, 2, 5]))
    mode = str(rng.choice(["sample_size", "equal"]))
    unsized = int(rng.integers(0, 7))          # a client without a recorded sample size (6: none)
    rounds = []
    for r in range(4):
        K = int(rng.integers(1, 7))
        rounds.append((K, bool(rng.integers(0, 2)), bool(K > 1 and r and rng.random() < 0.3)))
    return alg, spec, pieces, mode, unsized, rounds


def _random_whole_clients(spec, seed, r, K, drop):
    import scenarios as S
    models = {c: S.make_state(spec, seed=20_000 * seed + 100 * r + c, huge=False) for c in range(K)}
    if drop:
        del models[K - 1]["v"]
    return models


def _random_whole_worker(rank, world, port, seed, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_backend
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        alg, spec, pieces, mode, unsized, rounds = _random_whole_plan(seed)
        cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode=mode)
        agg = ShardedFedAvgAggregator(None, cfg, None, algorithm=alg, device="cpu", backend=cpu_backend,
                                      pieces=pieces, mode="whole")
        for c in range(6):
            if c != unsized:
                agg.set_client_sample_size(c, 3 + 2 * c)
        recs = []
        for r, (K, staged, drop) in enumerate(rounds):
            models = _random_whole_clients(spec, seed, r, K, drop)
            mine = {c: models[c] for c in models if c % world == rank}      # client k -> rank k mod R
            if staged:
                for c, m in mine.items():
                    agg.stage(c, m)
            recs.append({k: v.numpy().copy() for k, v in agg.aggregate(mine).items()})
        if rank == world - 1:
            np.save(out, np.array(recs, dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed,world", [(0, 2), (1, 3), (2, 3), (3, 2)])
def test_random_whole_update_rounds_on_several_ranks(seed, world, tmp_path):
    import scenarios as S
    from oracle import aggregators as OA
    out = str(tmp_path / "recs.npy")
    mp.spawn(_random_whole_worker, args=(world, _free_port(), seed, out), nprocs=world, join=True)
    got = list(np.load(out, allow_pickle=True))
    alg, spec, pieces, mode, unsized, rounds = _random_whole_plan(seed)
    ora = OA.ORACLES[alg + "Aggregator"](None, dict(S.HP[alg + "Aggregator"], client_weights_mode=mode), None)
    for c in range(6):
        if c != unsized:
            ora.set_client_sample_size(c, 3 + 2 * c)
    for r, (K, staged, drop) in enumerate(rounds):
        want = ora.aggregate(_random_whole_clients(spec, seed, r, K, drop))
        assert list(got[r]) == list(want), (seed, r)
        for k in want:
            w = np.asarray(want[k])
            assert got[r][k].dtype == w.dtype and got[r][k].shape == w.shape, (seed, r, k)
            if w.dtype == np.float32:      # no model: every fp32 entry is weighted -- the sum is reordered
                assert _close(got[r][k], w, 1e-6 * (r + 1)), (seed, alg, pieces, r, k)
            else:
                assert got[r][k].tobytes() == w.tobytes(), (seed, alg, pieces, r, k)
