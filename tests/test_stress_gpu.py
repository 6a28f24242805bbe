"""Randomised rounds through the stage-on-arrival machinery of the aggregators (clients uploaded
on arrival in random order and number, folded while the others are awaited, the last arrival
pipelined with the kernels, clients staged twice, clients lacking an entry, changing client
counts), every result bit for bit against the oracle.  The paths differ in stream and event
ordering only, so what this looks for is a result that depends on timing."""

import os

import numpy as np
import pytest
import torch

import golden_util
import scenarios as S
from oracle import aggregators as OA

pytestmark = pytest.mark.gpu
SEED_OFFSET = int(os.environ.get("APPFL_AMD_FUZZ_OFFSET", "0"))
CLASSES = ["FedAvgAggregator", "FedAvgMAggregator", "FedAdamAggregator", "FedYogiAggregator"]


def _spec(kind, n):
    return {"w": ((n,), kind), "steps": ((3,), "i64"), "b": ((7, 3), kind), "c32": ((5,), "i32"),
            "v": ((n // 3 + 1,), kind)}


def _same(got, want, what):
    a, b = S._snap(got), S._snap(want)
    assert list(a) == list(b), what
    for k in b:
        assert golden_util.same_bits(a[k], b[k]), (what, k)


@pytest.mark.parametrize("seed", range(8))
def test_random_staged_rounds(seed, monkeypatch):
    import appfl_amd.aggregator as A
    from appfl_amd.aggregator.fedavg_aggregator import FedAvgAggregator
    rng = np.random.default_rng(9000 + seed + SEED_OFFSET)
    kind = ["f32", "bf16", "f16"][int(rng.integers(0, 3))]
    cls = CLASSES[int(rng.integers(0, len(CLASSES)))]
    n = int(rng.choice([70_001, 600_003, 4_600_003, 4_600_003, 5_100_001]))
    monkeypatch.setattr(FedAvgAggregator, "STAGE_WINDOW", int(rng.choice([1, 2, 8])) << 20)
    spec = _spec(kind, n)
    cfg = dict(S.HP[cls], client_weights_mode=str(rng.choice(["sample_size", "equal"])))
    agg, ora = getattr(A, cls)(None, cfg, None), OA.ORACLES[cls](None, cfg, None)
    for a in (agg, ora):
        for c in range(8):
            a.set_client_sample_size(c, 3 + 2 * c)
    for r in range(5):
        K = int(rng.integers(1, 8))
        ids = [int(c) for c in rng.permutation(8)[:K]]
        models = {c: S.make_state(spec, seed=100_000 * seed + 100 * r + c, huge=False) for c in ids}
        if K > 1 and r and rng.random() < 0.3:
            del models[ids[-1]]["v"]                     # a client lacking a weighted entry
        # what the scheduler was configured with: the round's size (the folding of early clients
        # relies on it), another size, or nothing
        hint = [K, K, K + 1, 0][int(rng.integers(0, 4))]
        agg.aggregator_configs["num_clients"] = hint
        if rng.random() < 0.6:                           # as a scheduler would: in arrival order
            order = ids[:int(rng.integers(max(K - 1, 0), K + 1))]
        else:                                            # some clients, in any order
            order = [ids[i] for i in rng.permutation(K)[:int(rng.integers(0, K + 1))]]
        what = f"seed {seed} {cls} {kind} n={n} round {r}: clients {ids}, staged {order}, num_clients {hint}"
        for c in order:
            agg.stage(c, models[c])
        if order and rng.random() < 0.25:
            agg.stage(order[0], models[order[0]])        # the same client again
        if order and rng.random() < 0.2:
            # a staged client sends again and the scheduler's dict holds the new message, which
            # was never staged: the round must take the dict's
            c = order[int(rng.integers(0, len(order)))]
            models[c] = S.make_state(spec, seed=100_000 * seed + 100 * r + c + 50, huge=False)
            what += f", client {c} replaced un-staged"
        got = agg.aggregate(models)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        _same(got, want, what)
        if rng.random() < 0.5:
            _same(agg.get_parameters(), want, what + " (served)")


@pytest.mark.parametrize("kind", ["f32", "bf16"])
def test_single_client_rounds_wait_for_their_upload(kind):
    """One client per round, not staged on arrival: the kernel may only start once that client's
    row has landed (with no landed row there is nothing to start on early)."""
    import appfl_amd.aggregator as A
    spec = _spec(kind, 6_000_001)
    agg, ora = A.FedAvgAggregator(None, {}, None), OA.OracleFedAvg(None, {}, None)
    for r in range(4):
        models = {0: S.make_state(spec, seed=77 + r, huge=False)}
        got = agg.aggregate(models)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        _same(got, want, (kind, r))


@pytest.mark.parametrize("seed", range(6))
def test_random_async_arrivals(seed):
    """FedAsync / FedBuff: random arrival orders, downloads in between, model sizes on both
    sides of the pipelined arrival's threshold, every staleness function; each reply against
    the oracle (the pipelined arrival and FedBuff's early download overlap three streams)."""
    import appfl_amd.aggregator as A
    rng = np.random.default_rng(7000 + seed + SEED_OFFSET)
    kind = ["f32", "bf16", "f16"][int(rng.integers(0, 3))]
    cls = ["FedAsyncAggregator", "FedBuffAggregator"][int(rng.integers(0, 2))]
    n = int(rng.choice([70_001, 600_003, 4_600_003, 5_100_001]))
    spec = _spec(kind, n)
    stale = dict(list(S.STALENESS.values())[int(rng.integers(0, len(S.STALENESS)))])
    cfg = dict(stale, alpha=float(rng.choice([0.9, 0.5])), num_clients=4,
               gradient_based=bool(rng.integers(0, 2)), optimize_memory=bool(rng.integers(0, 2)),
               client_weights_mode=str(rng.choice(["sample_size", "equal"])))
    if cls == "FedBuffAggregator":
        cfg["K"] = int(rng.integers(1, 4))
    agg, ora = getattr(A, cls)(None, cfg, None), OA.ORACLES[cls](None, cfg, None)
    for a in (agg, ora):
        for c in range(4):
            a.set_client_sample_size(c, 3 + 2 * c)
    for it in range(10):
        c = int(rng.integers(0, 4))
        what = f"seed {seed} {cls} {kind} n={n} cfg={cfg} arrival {it} of client {c}"
        model = S.make_state(spec, seed=100_000 * seed + 10 * it + c, huge=False, gradient=cfg["gradient_based"])
        got = agg.aggregate(c, model)
        want = ora.aggregate(c, {k: v.clone() for k, v in model.items()})
        _same(got, want, what)
        if rng.random() < 0.4:
            d = int(rng.integers(0, 4))
            _same(agg.get_parameters(client_id=d), ora.get_parameters(client_id=d), what + f" (download by {d})")


@pytest.mark.parametrize("seed", range(6))
def test_random_api_sequences(seed):
    """Whole rounds (some staged on arrival), streamed rounds chunk by chunk, chunked and whole
    `get_parameters`, in random order on one aggregator, with or without a model: every reply
    against the oracle."""
    import appfl_amd.aggregator as A
    rng = np.random.default_rng(8000 + seed + SEED_OFFSET)
    kind = ["f32", "bf16"][int(rng.integers(0, 2))]
    cls = CLASSES[int(rng.integers(0, len(CLASSES)))]
    with_model = bool(rng.integers(0, 2))
    spec_name = "toy" if kind == "f32" else f"toy_{kind}"
    spec = S.SPECS[spec_name]
    cfg = dict(S.HP[cls], client_weights_mode=str(rng.choice(["sample_size", "equal"])))
    mk = S.MODELS[spec_name] if with_model else (lambda: None)
    agg, ora = getattr(A, cls)(mk(), cfg, None), OA.ORACLES[cls](mk(), cfg, None)
    for a in (agg, ora):
        for c in range(6):
            a.set_client_sample_size(c, 3 + 2 * c)
    chunk_bytes = int(rng.choice([200, 600, 1500]))
    chunks = S.chunk_keys_by_size(spec, chunk_bytes)
    for a in (agg, ora):
        a.set_model_chunk_size(chunk_bytes)
    for it in range(12):
        act = int(rng.integers(0, 4))
        what = f"seed {seed} {cls} {kind} model={with_model} step {it} action {act}"
        K = int(rng.integers(1, 6))
        ids = [int(c) for c in rng.permutation(6)[:K]]
        states = {c: S.make_state(spec, seed=100_000 * seed + 100 * it + c) for c in ids}
        if act == 0:                                     # a whole round, some clients staged on arrival
            for c in ids[:int(rng.integers(0, K + 1))]:
                agg.stage(c, states[c])
            _same(agg.aggregate(states),
                  ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in states.items()}), what)
        elif act == 1:                                   # a streamed round
            for ci, keys in enumerate(chunks):
                kw = dict(_chunk_idx={c: ci for c in ids}, _total_chunks={c: len(chunks) for c in ids},
                          _chunk_keys={c: list(keys) for c in ids})
                got = agg.aggregate({c: {k: states[c][k].clone() for k in keys} for c in ids}, **kw)
                want = ora.aggregate({c: {k: states[c][k].clone() for k in keys} for c in ids}, **kw)
                _same(got, want, what + f" chunk {ci}")
        elif act == 2:                                   # a chunked download
            if not with_model and ora.global_state is None:
                continue
            cid = int(rng.integers(0, len(chunks)))
            kw = {"_request_chunked": True} if rng.random() < 0.5 else {"_chunk_id": cid}
            got, want = agg.get_parameters(**kw), ora.get_parameters(**kw)
            if isinstance(want, tuple):
                assert got[1] == want[1], what
                got, want = got[0], want[0]
            _same(got, want, what + f" {kw}")
        else:
            if not with_model and ora.global_state is None:
                continue
            _same(agg.get_parameters(), ora.get_parameters(), what)


@pytest.mark.parametrize("seed", range(6))
def test_random_rounds_of_the_sharded_class_on_one_rank(seed):
    """The sharded state-dict aggregator on one rank (both modes, random piece counts, staged
    arrivals, changing client sets) against the plain aggregator's oracle: with one rank no
    collective reorders anything, so every bit must agree."""
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
    rng = np.random.default_rng(6000 + seed + SEED_OFFSET)
    mode = ["sliced", "whole"][int(rng.integers(0, 2))]
    kind = "f32" if mode == "whole" else ["f32", "bf16", "f16"][int(rng.integers(0, 3))]
    alg = ["FedAvg", "FedAvgM", "FedAdam", "FedYogi", "FedAdagrad"][int(rng.integers(0, 5))]
    n = int(rng.choice([5, 1003, 70_001, 600_003]))
    spec = _spec(kind, n)
    cfg = dict(S.HP[alg + "Aggregator"], client_weights_mode=str(rng.choice(["sample_size", "equal"])))
    agg = ShardedFedAvgAggregator(None, cfg, None, algorithm=alg, pieces=int(rng.choice([1, 2, 3, 7])), mode=mode)
    ora = OA.ORACLES[alg + "Aggregator"](None, cfg, None)
    for a in (agg, ora):
        for c in range(6):
            a.set_client_sample_size(c, 3 + 2 * c)
    for r in range(4):
        K = int(rng.integers(1, 6))
        ids = [int(c) for c in rng.permutation(6)[:K]]
        models = {c: S.make_state(spec, seed=100_000 * seed + 100 * r + c, huge=False) for c in ids}
        if K > 1 and r and rng.random() < 0.3:
            del models[ids[-1]]["v"]
        what = f"seed {seed} {alg} {mode} {kind} n={n} round {r}: clients {ids}"
        for c in ids[:int(rng.integers(0, K + 1))]:
            agg.stage(c, models[c])
        got = agg.aggregate(models)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        _same(got, want, what)
        if rng.random() < 0.5:
            _same(agg.get_parameters(), want, what + " (served)")
    if alg != "FedAvg":
        m, v = agg.gather_moments()
        _same({k: t.cpu() for k, t in v.items()}, ora.v_vector, "v")


@pytest.mark.parametrize("seed", range(6))
def test_random_fedcompass_and_fedqueue_sequences(seed):
    """Single arrivals and groups in random order (FedCompass), groups of changing membership
    (FedQueue), random staleness and sizes, fp32 / bf16 / f16: every reply against the oracle."""
    import appfl_amd.aggregator as A
    rng = np.random.default_rng(5000 + seed + SEED_OFFSET)
    kind = ["f32", "bf16", "f16"][int(rng.integers(0, 3))]
    cls = ["FedCompassAggregator", "FedQueueAggregator"][int(rng.integers(0, 2))]
    n = int(rng.choice([5, 1003, 70_001, 600_003]))
    spec = _spec(kind, n)
    gb = bool(rng.integers(0, 2))
    if cls == "FedCompassAggregator":
        stale = dict(list(S.STALENESS.values())[int(rng.integers(0, len(S.STALENESS)))])
        cfg = dict(stale, alpha=float(rng.choice([0.9, 0.5])), num_clients=5, gradient_based=gb)
    else:
        cfg = [{"staleness_fn": "harmonic", "staleness_fn_kwargs": {"beta": 0.5}, "gradient_based": gb},
               {"gradient_based": gb}][int(rng.integers(0, 2))]
    agg, ora = getattr(A, cls)(None, cfg, None), OA.ORACLES[cls](None, cfg, None)
    if rng.random() < 0.6:
        for a in (agg, ora):
            for c in range(5):
                a.set_client_sample_size(c, 3 + 2 * c)
    for it in range(8):
        K = int(rng.integers(1, 5))
        ids = [int(c) for c in rng.permutation(5)[:K]]
        what = f"seed {seed} {cls} {kind} n={n} cfg={cfg} step {it}: clients {ids}"
        models = {c: S.make_state(spec, seed=100_000 * seed + 100 * it + c, huge=False, gradient=gb) for c in ids}
        stale_of = {c: int(rng.integers(0, 6)) for c in ids}
        copies = {c: {k: v.clone() for k, v in m.items()} for c, m in models.items()}
        if cls == "FedCompassAggregator" and rng.random() < 0.5:
            c = ids[0]
            got = agg.aggregate(client_id=c, local_model=models[c], staleness=stale_of[c])
            want = ora.aggregate(client_id=c, local_model=copies[c], staleness=stale_of[c])
        elif cls == "FedCompassAggregator":
            got = agg.aggregate(local_models=models, staleness=stale_of)
            want = ora.aggregate(local_models=copies, staleness=stale_of)
        else:
            got = agg.aggregate(models, stale_of)
            want = ora.aggregate(copies, stale_of)
        _same(got, want, what)
        if rng.random() < 0.3:
            _same(agg.get_parameters(), ora.get_parameters(), what + " (served)")


@pytest.mark.parametrize("staged", ["all", "all_but_last"])
def test_more_clients_than_one_launch_takes_through_the_staged_paths(staged):
    """130 clients of a model above the pipelines' threshold: the folding on arrival, the
    pipelined last arrival and the plain round all carry the partial step across launches."""
    import appfl_amd.aggregator as A
    from appfl_amd import ops
    K = ops.kchunk() + 2
    spec = {"w": ((4_200_001,), "f32"), "steps": ((), "i64")}
    cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="equal", num_clients=K)
    agg, ora = A.FedAdamAggregator(None, cfg, None), OA.OracleFedAdam(None, cfg, None)
    for r in range(1):       # (the un-staged round of so many clients runs in tests/test_aggregators_gpu.py)
        models = {c: S.make_state(spec, seed=1000 * r + c, huge=False) for c in range(K)}
        n_staged = {"all": K, "all_but_last": K - 1}[staged]
        for c in range(n_staged):
            agg.stage(c, models[c])
        got = agg.aggregate(models)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        _same(got, want, (staged, r))
