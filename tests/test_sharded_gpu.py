"""The sharded aggregators on one GPU with the HIP backend (world size 1: every mode must equal
the plain fused round), complementing the multi-rank gloo rehearsals."""

import numpy as np
import pytest

import golden_util
import torch

from oracle import aggregators as OA
from oracle import c_oracle as OC
from oracle import synth

pytestmark = pytest.mark.gpu
HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)


@pytest.mark.parametrize("mode,pieces", [("sliced", 1), ("pipelined", 5), ("whole", 3), ("whole", 1)])
def test_single_rank_modes_match_oracle(mode, pieces):
    from appfl_amd.sharded import ShardedFlatAggregator
    n, K = 100003, 6
    base = synth.fill(n, 7, scale=0.05)
    g, m, v = base.copy(), np.zeros(n, np.float32), np.full(n, np.float32(HP["tau"] ** 2))
    agg = ShardedFlatAggregator(n, "adam", pieces=pieces, **HP)
    agg.load_global(torch.from_numpy(base).cuda())
    for r in range(3):
        xs = [synth.fill(n, 100 * r + k, scale=0.02, base=base) for k in range(K)]
        w = [1.0 / K] * K
        OC.reduce_opt(xs, w, g, m, v, OC.OPT_ADAM, **HP)
        dx = [torch.from_numpy(x).cuda() for x in xs]
        if mode == "sliced":
            agg.round_sliced([agg.map.take(x) for x in dx], w)
        elif mode == "pipelined":
            full = agg.round_sliced_gather([agg.map.take(x) for x in dx], w)
            assert torch.equal(full, agg.all_gather_global())
        else:
            agg.round_whole(dx, w)
    got = agg.all_gather_global().cpu().numpy()
    assert np.array_equal(got[:n].view(np.uint32), g.view(np.uint32)) and not got[n:].any()


@pytest.mark.parametrize("spec_name", ["toy", "mixed64"])
def test_state_dict_sharded_aggregator_single_rank_matches_plain_aggregator_and_oracle(spec_name):
    import scenarios as S
    import appfl_amd.aggregator as A
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
    spec, model = (S.TOY, S.ToyModel) if spec_name == "toy" else (S.MIXED64, S.Mixed64Model)
    cfg = dict(S.HP["FedYogiAggregator"], client_weights_mode="sample_size")
    plain = A.FedYogiAggregator(model(), cfg, None)
    shard = ShardedFedAvgAggregator(model(), cfg, None, algorithm="FedYogi", pieces=3)
    ora = OA.OracleFedYogi(model(), cfg, None)
    for agg in (plain, shard, ora):
        for c in range(4):
            agg.set_client_sample_size(c, 5 + 2 * c)
    first = shard.get_parameters()
    for k, t in ora.get_parameters().items():
        assert golden_util.same_bits(first[k].numpy(), np.asarray(t)), k
    for r, K in enumerate([3, 4, 2]):
        local = {c: S.make_state(spec, seed=900 + 10 * r + c) for c in range(K)}
        want = plain.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        truth = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        if r == 1:
            for c, m in local.items():
                shard.stage(c, m)
        got = shard.aggregate(local)
        assert list(got) == list(want) == list(truth)
        for k in want:
            assert golden_util.same_bits(got[k].numpy(), want[k].numpy()), (r, k)
            assert golden_util.same_bits(got[k].numpy(), np.asarray(truth[k])), (r, k)
    again = shard.get_parameters()
    for k in got:
        assert golden_util.same_bits(again[k].numpy(), got[k].numpy()), k
    gs = shard.global_state
    for k in got:
        assert golden_util.same_bits(gs[k].cpu().numpy(), got[k].numpy()), k


def _two_rank_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    # the loopback interface: the box's own host name need not resolve
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    torch.cuda.set_device(0)
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)      # a stuck rank says where, then leaves
    print(f"[rank {rank}] init", flush=True)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    print(f"[rank {rank}] group up", flush=True)

    # gloo's own handling of device tensors proved unreliable with two ranks on one GPU (both
    # ranks stuck inside a blocking all_gather_into_tensor); what this test is about are the
    # kernels and the block map, so the collectives carry host copies here.
    class _Done:
        def wait(self):
            return True

    real_gather, real_reduce = dist.all_gather_into_tensor, dist.all_reduce

    def gather_via_host(out, inp, group=None, async_op=False):
        host = torch.empty(out.shape, dtype=out.dtype)
        real_gather(host, inp.cpu(), group=group)
        out.copy_(host)
        return _Done()

    def reduce_via_host(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        host = t.cpu()
        real_reduce(host, op=op, group=group)
        t.copy_(host)
        return _Done()

    dist.all_gather_into_tensor, dist.all_reduce = gather_via_host, reduce_via_host
    try:
        import scenarios as S
        from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
        from appfl_amd.sharded import ShardedFlatAggregator
        # state-dict aggregator: HIP kernels on this rank's blocks, gloo moving device tensors
        cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="sample_size")
        agg = ShardedFedAvgAggregator(S.ToyModel(), cfg, None, algorithm="FedAdam", pieces=3)
        for c in range(5):
            agg.set_client_sample_size(c, 4 + 3 * c)
        recs = []
        for r, K in enumerate([2, 5, 3]):
            local = {c: S.make_state(S.TOY, seed=700 + 10 * r + c) for c in range(K)}
            recs.append({k: v.numpy().copy() for k, v in agg.aggregate(local).items()})
            print(f"[rank {rank}] state-dict round {r} done", flush=True)
        # a bfloat16 model: 2-byte blocks, the 16-bit kernels
        agg16 = ShardedFedAvgAggregator(S.MODELS["toy_bf16"](), cfg, None, algorithm="FedAdam", pieces=3)
        recs16 = []
        for r, K in enumerate([3, 2]):
            local = {c: S.make_state(S.SPECS["toy_bf16"], seed=800 + 10 * r + c) for c in range(K)}
            recs16.append(S._snap(agg16.aggregate(local)))
            print(f"[rank {rank}] 16-bit state-dict round {r} done", flush=True)
        # flat aggregator, pipelined and whole-update rounds, on a vector with several block rows
        n, K = 300007, 6
        base = synth.fill(n, 7, scale=0.05)
        flat = ShardedFlatAggregator(n, "adam", pieces=4, **HP)
        flat.load_global(torch.from_numpy(base).cuda())
        for r in range(2):
            xs = [torch.from_numpy(synth.fill(n, 100 * r + k, scale=0.02, base=base)).cuda() for k in range(K)]
            w = [1.0 / K] * K
            if r == 0:
                flat.round_sliced_gather([flat.map.take(x) for x in xs], w)
            else:
                mine = [k for k in range(K) if k % world == rank]
                flat.round_whole([xs[k] for k in mine], [w[k] for k in mine])
            print(f"[rank {rank}] flat round {r} done", flush=True)
        g = flat.all_gather_global().cpu().numpy()[:n]
        if rank == 1:
            np.save(out, np.array([recs, g, recs16], dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_hip_kernels_with_gloo_collectives(tmp_path):
    """World size 2 with the REAL kernels: both ranks share cuda:0 and gloo carries the
    collectives through host copies (RCCL refuses two ranks on one device), so the block map, the
    windowed rows and the HIP launches are exercised together; results against the
    single-process oracles."""
    import socket

    import scenarios as S
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "recs.npy")
    ctx = mp.spawn(_two_rank_worker, args=(2, port, out), nprocs=2, join=False)
    import time
    deadline = time.time() + 90
    while not ctx.join(timeout=1.0):          # True once every worker has exited cleanly
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            pytest.fail("the two-rank run did not finish within 90 s")
    recs, g, recs16 = np.load(out, allow_pickle=True)
    cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="sample_size")
    ora = OA.OracleFedAdam(S.ToyModel(), cfg, None)
    for c in range(5):
        ora.set_client_sample_size(c, 4 + 3 * c)
    for r, K in enumerate([2, 5, 3]):
        want = ora.aggregate({c: S.make_state(S.TOY, seed=700 + 10 * r + c) for c in range(K)})
        for k in want:
            assert golden_util.same_bits(recs[r][k], want[k]), (r, k)
    ora16 = OA.OracleFedAdam(S.MODELS["toy_bf16"](), cfg, None)
    for r, K in enumerate([3, 2]):
        want = S._snap(ora16.aggregate({c: S.make_state(S.SPECS["toy_bf16"], seed=800 + 10 * r + c)
                                        for c in range(K)}))
        for k in want:
            assert golden_util.same_bits(recs16[r][k], want[k]), ("bf16", r, k)
    n, K = 300007, 6
    base = synth.fill(n, 7, scale=0.05)
    gw, m, v = base.copy(), np.zeros(n, np.float32), np.full(n, np.float32(HP["tau"] ** 2))
    for r in range(2):
        xs = [synth.fill(n, 100 * r + k, scale=0.02, base=base) for k in range(K)]
        OC.reduce_opt(xs, [1.0 / K] * K, gw, m, v, OC.OPT_ADAM, **HP)
    # the second round sums the two ranks' partial steps: fp32 additions reordered
    assert np.max(np.abs(g - gw)) <= 1e-6 * np.max(np.abs(gw))


@pytest.mark.parametrize("alg,with_model", [("FedAdam", True), ("FedAvg", False), ("FedAvgM", True)])
def test_whole_update_mode_single_rank_matches_oracle_bit_for_bit(alg, with_model):
    """`mode="whole"` on one rank: no collective reorders anything, so the un-fused sequence
    (partial step, optimizer epilogue, apply, means over the gathered row tails) must give the
    oracle's bits -- clients staged before the state exists, a client lacking a weighted entry,
    `get_parameters`, `compute_steps` and the moments included."""
    import scenarios as S
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
    cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode="sample_size")
    model = S.ToyModel if with_model else (lambda: None)
    agg = ShardedFedAvgAggregator(model(), cfg, None, algorithm=alg, pieces=3, mode="whole")
    ora = OA.ORACLES[alg + "Aggregator"](model(), cfg, None)
    for a in (agg, ora):
        for c in range(5):
            if c != 3:
                a.set_client_sample_size(c, 4 + 3 * c)
    for r, K in enumerate([2, 5, 3]):
        local = {c: S.make_state(S.TOY, seed=1700 + 10 * r + c) for c in range(K)}
        if r == 1:
            del local[2]["fc.weight"]
        if r in (0, 2):
            for c, m in local.items():
                agg.stage(c, m)
        got = agg.aggregate(local)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        assert list(got) == list(want)
        for k in want:
            assert golden_util.same_bits(got[k].numpy(), np.asarray(want[k])), (r, k)
    again = agg.get_parameters()
    for k in got:
        assert golden_util.same_bits(again[k].numpy(), got[k].numpy()), k
    local = {c: S.make_state(S.TOY, seed=1800 + c) for c in range(4)}
    agg.compute_steps(local)
    ora.compute_steps({c: {k: v.numpy() for k, v in m.items()} for c, m in local.items()})
    assert sorted(agg.step) == sorted(ora.step)
    for k, t in ora.step.items():
        assert golden_util.same_bits(agg.step[k].cpu().numpy(), t), ("step", k)
    if alg != "FedAvg":
        m, v = agg.gather_moments()
        for name, got_d, want_d in (("m", m, getattr(ora, "m_vector", {})), ("v", v, ora.v_vector)):
            assert sorted(got_d) == sorted(want_d), name
            for k, t in want_d.items():
                assert golden_util.same_bits(got_d[k].cpu().numpy(), t), (name, k)
                assert golden_util.same_bits(getattr(agg, name + "_vector")[k].cpu().numpy(), t), (name, k)


def test_sliced_sharded_class_serves_compute_steps_moments_streamed_and_flat_payload_rounds():
    """What the reference's class exposes beside the fused round, on the HIP kernels (one rank):
    `compute_steps` + `step`, `m_vector` / `v_vector`, a subclass overriding `compute_steps`,
    streamed rounds sharing the state with whole rounds, flat secure-aggregation payloads."""
    import scenarios as S
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator

    def same(got, want, what):
        assert list(got) == list(want), what
        for k in want:
            g = got[k].cpu().numpy() if isinstance(got[k], torch.Tensor) else got[k]
            assert golden_util.same_bits(g, np.asarray(want[k])), (what, k)

    cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="equal")
    agg = ShardedFedAvgAggregator(S.ToyModel(), cfg, None, algorithm="FedAdam", pieces=2)
    ora = OA.OracleFedAdam(S.ToyModel(), cfg, None)
    local = {c: S.make_state(S.TOY, seed=2100 + c) for c in range(3)}
    same(agg.aggregate(local), ora.aggregate(local), "round0")
    same(agg.m_vector, ora.m_vector, "m")
    same(agg.v_vector, ora.v_vector, "v")
    local = {c: S.make_state(S.TOY, seed=2200 + c) for c in range(4)}
    agg.compute_steps(local)
    ora.compute_steps({c: {k: v.numpy() for k, v in m.items()} for c, m in local.items()})
    same(agg.step, ora.step, "step")

    class Halved(ShardedFedAvgAggregator):
        def compute_steps(self, local_models):
            super().compute_steps(local_models)
            for t in self.step.values():
                t.mul_(0.5)

    class HalvedOracle(OA.OracleFedAvgM):
        def compute_steps(self, local_models):
            super().compute_steps(local_models)
            for k in self.step:
                self.step[k] = self.step[k] * np.float32(0.5)
    sub = Halved(None, dict(S.HP["FedAvgMAggregator"]), None, algorithm="FedAvgM", pieces=3)
    sub_o = HalvedOracle(None, dict(S.HP["FedAvgMAggregator"]), None)
    for r in range(2):
        local = {c: S.make_state(S.TOY, seed=2300 + 10 * r + c) for c in range(3)}
        same(sub.aggregate(local), sub_o.aggregate(local), f"halved{r}")

    st, st_o = ShardedFedAvgAggregator(None, {}, None, pieces=2), OA.OracleFedAvg(None, {}, None)
    chunks = S.chunk_keys_by_size(S.TOY, 600)
    for r in range(2):
        states = {c: S.make_state(S.TOY, seed=2400 + 10 * r + c) for c in range(3)}
        if r == 1:
            whole = {c: S.make_state(S.TOY, seed=2450 + c) for c in range(2)}
            same(st.aggregate(whole), st_o.aggregate(whole), "streamed_whole")
        for ci, keys in enumerate(chunks):
            kw = dict(_chunk_idx={c: ci for c in states}, _total_chunks={c: len(chunks) for c in states},
                      _chunk_keys={c: list(keys) for c in states})
            part = {c: {k: states[c][k].clone() for k in keys} for c in states}
            same(st.aggregate(part, **kw), st_o.aggregate(part, **kw), f"streamed{r}.{ci}")
    same(st.get_parameters(), st_o.get_parameters(), "streamed_final")

    scfg = {"use_secure_agg": True, "client_weights_mode": "sample_size"}
    sec, sec_o = ShardedFedAvgAggregator(None, scfg, None), OA.OracleFedAvg(None, scfg, None)
    names = [n for n, (_, kind) in S.TOY.items() if kind == "f32"]
    for r in range(2):
        payload = {}
        for c in range(3):
            sd = S.make_state(S.TOY, seed=2500 + 10 * r + c, gradient=True, scale=0.01)
            payload[c] = {"type": "masked_update_flat",
                          "flat": torch.cat([sd[n].reshape(-1) for n in names]),
                          "shapes": [(n, tuple(sd[n].shape)) for n in names], "num_examples": 5 + c}
        same(sec.aggregate(payload), sec_o.aggregate(payload), f"secure{r}")


@pytest.mark.parametrize("opt,K,n", [("none", 48, 21_000_003), ("none", 8, 80_000_004), ("adam", 48, 9_000_001),
                                     ("avgm", 64, 5_000_000)])
def test_rounds_kept_off_the_persistent_kernels_give_the_same_bits(opt, K, n):
    """fa_reduce_opt_ex with FA_ROUTE_NO_PERSISTENT (what the sharded rounds pass beside a running
    gather): shapes the default routing sends to a persistent kernel must come out bit-identical
    from the tile-per-block kernel."""
    from appfl_amd import ops
    assert ops.reduce_kernel_name(K, n, opt) != "reduce_kernel"
    dev = torch.device("cuda")
    g0 = torch.empty(n, device=dev)
    ops.synth_fill(g0, 7, scale=0.05)
    xs = torch.empty((K, n), device=dev)
    for k in range(K):
        ops.synth_fill(xs[k], 1000 + k, scale=0.01, base=g0)
    w = [(k + 1) / (K * (K + 1) / 2) for k in range(K)]
    outs = []
    for persistent in (True, False):
        g, m, v = g0.clone(), torch.zeros(n, device=dev), torch.full((n,), 1e-6, device=dev)
        for _ in range(2):
            ops.reduce_opt(list(xs), w, g, m if opt in ("adam",) else None, v if opt != "none" else None,
                           opt, persistent=persistent, **HP)
        outs.append((g, m, v))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# -- bfloat16 / float16 models (sliced rounds) ----------------------------------------------------

def _same_records(got, want, what):
    import scenarios as S
    a, b = S._snap(got), S._snap(want)
    assert list(a) == list(b), what
    for k in b:
        assert golden_util.same_bits(a[k], b[k]), (what, k)


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("alg,spec_name", [("FedYogi", "toy"), ("FedAvgM", "ragged"), ("FedAvg", "ragged")])
def test_16_bit_state_dict_sharded_single_rank_matches_plain_aggregator_and_oracle(kind, alg, spec_name):
    import scenarios as S
    import appfl_amd.aggregator as A
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
    spec = S.SPECS[f"{spec_name}_{kind}"]
    model = S.MODELS[f"toy_{kind}"] if spec_name == "toy" else (lambda: None)
    cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode="sample_size")
    plain = getattr(A, alg + "Aggregator")(model(), cfg, None)
    shard = ShardedFedAvgAggregator(model(), cfg, None, algorithm=alg, pieces=3)
    ora = OA.ORACLES[alg + "Aggregator"](model(), cfg, None)
    for agg in (plain, shard, ora):
        for c in range(4):
            agg.set_client_sample_size(c, 5 + 2 * c)
    for r, K in enumerate([3, 4, 2]):
        local = {c: S.make_state(spec, seed=950 + 10 * r + c) for c in range(K)}
        if r == 1:
            del local[1][next(n for n, (shp, _) in spec.items() if len(shp) == 2)]
        want = plain.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        truth = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        if r == 1:
            for c, m in local.items():
                shard.stage(c, m)
        got = shard.aggregate(local)
        assert shard._map.itemsize == 2 and shard._gathered.dtype == S.HALF_KINDS[kind]
        assert shard._slab.row_bytes == shard._layout.row_bytes < 2 * shard._full_layout.row_bytes
        _same_records(got, want, (r, "plain"))
        _same_records(got, truth, (r, "oracle"))
    _same_records(shard.get_parameters(), got, "served")
    _same_records({k: v.cpu() for k, v in shard.global_state.items()}, got, "global_state")
    # the step and the moments of a further round
    local = {c: S.make_state(spec, seed=990 + c) for c in range(3)}
    shard.compute_steps(local)
    ora.compute_steps({c: {k: OA.to_np(v) for k, v in m.items()} for c, m in local.items()})
    _same_records({k: v.cpu() for k, v in shard.step.items()}, ora.step, "step")
    if alg != "FedAvg":
        m, v = shard.gather_moments()
        _same_records({k: t.cpu() for k, t in v.items()}, ora.v_vector, "v")
        if alg != "FedAvgM":
            _same_records({k: t.cpu() for k, t in m.items()}, ora.m_vector, "m")


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_flat_sharded_rounds_on_16_bit_vectors_equal_the_unsharded_kernel(kind):
    """Block rows of a 16-bit vector (several per rank, the last one partial), each reduced by
    its own launch and gathered: the bits of one launch over the whole vector."""
    import scenarios as S
    from appfl_amd import half_ops
    from appfl_amd.sharded import ShardedFlatAggregator
    dt = S.HALF_KINDS[kind]
    n, K = 1_000_003, 6
    base = torch.from_numpy(synth.fill(n, 7, scale=0.05)).to(dt).cuda()
    agg = ShardedFlatAggregator(n, "yogi", pieces=5, dtype=dt, **HP)
    agg.load_global(base)
    ops16 = half_ops.HalfOps(dt)
    g, m, v = base.clone(), torch.zeros_like(base), torch.zeros_like(base)
    ops16.fill(v, HP["tau"] ** 2)
    for r in range(3):
        xs = [torch.from_numpy(synth.fill(n, 100 * r + k, scale=0.02)).cuda().add_(base.float()).to(dt)
              for k in range(K)]
        w = [(k + 1) / 21 for k in range(K)]
        ops16.reduce_opt(xs, w, g, m, v, "yogi", **HP)
        full = agg.round_sliced_gather([agg.map.take(x) for x in xs], w)
        assert full.dtype == dt and torch.equal(full[:n].view(torch.int16), g.view(torch.int16))
        assert not full[n:].any()
    with pytest.raises(NotImplementedError, match="whole-update"):
        agg.round_whole([], [])


def test_whole_mode_refuses_a_16_bit_model():
    import scenarios as S
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
    agg = ShardedFedAvgAggregator(None, {}, None, mode="whole")
    with pytest.raises(NotImplementedError, match="whole-update"):
        agg.aggregate({c: S.make_state(S.SPECS["toy_f16"], seed=c) for c in range(2)})


def test_16_bit_sharded_round_with_more_clients_than_one_launch_takes():
    """130 clients: the client list runs as two launches per block row that carry the partial
    step through a 16-bit workspace of the block's length."""
    import scenarios as S
    import appfl_amd.aggregator as A
    from appfl_amd import ops
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
    K = ops.kchunk() + 2
    spec = S.SPECS["ragged_bf16"]
    cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="equal")
    plain = A.FedAdamAggregator(None, cfg, None)
    shard = ShardedFedAvgAggregator(None, cfg, None, algorithm="FedAdam", pieces=2)
    for r in range(2):
        local = {c: S.make_state(spec, seed=4000 + 200 * r + c, huge=False) for c in range(K)}
        want = plain.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        got = shard.aggregate(local)
        assert shard._step_flat is not None and shard._step_flat.dtype == torch.bfloat16
        _same_records(got, want, r)


@pytest.mark.parametrize("seed", range(6))
def test_random_block_maps_give_the_bits_of_the_unsharded_round(seed):
    """Random lengths, piece counts, client counts, optimizers and element types: the pipelined
    sliced round over the block rows against ONE launch over the whole vector."""
    import os
    from appfl_amd import half_ops, ops
    from appfl_amd.sharded import ShardedFlatAggregator
    rng = np.random.default_rng(31 + seed + int(os.environ.get("APPFL_AMD_FUZZ_OFFSET", "0")))
    for it in range(8):
        dt = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
        n = int(rng.choice([1, 5, 63, 1024, 4099, 65537, 300007, 1_000_003])) + int(rng.integers(0, 3))
        pieces = int(rng.choice([1, 2, 3, 5, 8]))
        K = int(rng.choice([1, 2, 7, 16, 33, 130]))
        if n * K > 20_000_000:
            K = max(1, 20_000_000 // n)
        opt = ["none", "avgm", "adam", "yogi", "adagrad"][int(rng.integers(0, 5))]
        hp = dict(lr=float(rng.choice([0.1, 1.0])), beta1=float(rng.choice([0.9, 0.5])),
                  beta2=float(rng.choice([0.99, 0.9])), tau=float(rng.choice([1e-3, 0.1])))
        what = f"seed {seed} case {it}: {dt} n={n} pieces={pieces} K={K} {opt}"
        o = ops if dt == torch.float32 else half_ops.HalfOps(dt)
        base = torch.from_numpy(synth.fill(n, 7 + it, scale=0.05)).cuda().to(dt)
        agg = ShardedFlatAggregator(n, opt, pieces=pieces, dtype=dt, **hp)
        agg.load_global(base)
        g = base.clone()
        m = torch.zeros_like(g) if opt in ("adam", "yogi", "adagrad") else None
        v = None
        if opt != "none":
            v = torch.zeros_like(g)
            if opt != "avgm":
                o.fill(v, hp["tau"] ** 2)
        w = (rng.random(K) + 0.05)
        w = (w / w.sum()).tolist()
        for r in range(2):
            xs = [(torch.from_numpy(synth.fill(n, 100 * r + k + 1000 * it, scale=0.02)).cuda() + base.float()).to(dt)
                  for k in range(K)]
            ws = torch.zeros(n, dtype=dt, device="cuda") if K > ops.kchunk() else None
            o.reduce_opt(xs, w, g, m, v, opt, step_ws=ws, **hp)
            ws_l = torch.zeros(agg.L, dtype=dt, device="cuda") if K > ops.kchunk() else None
            full = agg.round_sliced_gather([agg.map.take(x) for x in xs], w, step_ws=ws_l)
            bits = torch.int32 if dt == torch.float32 else torch.int16
            assert torch.equal(full[:n].view(bits), g.view(bits)), what + f" round {r}"
            assert not full[n:].any(), what
