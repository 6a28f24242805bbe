"""The CPU oracle against the golden vectors of the reference's own aggregator classes, and the
C oracle against the numpy oracle.  Runs without a GPU."""

import numpy as np
import pytest

import golden_util
import scenarios as S
from oracle import aggregators as OA
from oracle import c_oracle as OC
from oracle import synth


def oracle_factory(cls, model, cfg):
    return OA.ORACLES[cls](model, cfg, None)


@pytest.mark.parametrize("sc", S.all_scenarios(), ids=lambda s: s["id"])
def test_numpy_oracle_matches_reference_golden(sc):
    golden_util.check_records(sc["id"], S.run_scenario(oracle_factory, sc), S.reference_rtol(sc))


@pytest.mark.parametrize("cls", S.SQRT_CLASSES)
def test_first_round_moments_are_bit_exact_even_with_sqrt(cls):
    """m and v do not pass through sqrt, so the reference's round-0 moments are reproduced
    bit for bit; only the step (and what follows from it) carries the sqrt tolerance."""
    sc = next(s for s in S.sync_scenarios() if s["cls"] == cls and not s["model"])
    rec = S.run_scenario(oracle_factory, sc)
    z, index = golden_util.load()
    for i, (label, d) in enumerate(rec):
        if label in ("round0.m_vector", "round0.v_vector"):
            assert golden_util.record_hash(d) == index[sc["id"]][i]["sha256"], label


def test_sqrt_classes_stay_within_the_stated_elementwise_bounds(capsys):
    """The three aggregators with a sqrt in their step against the reference's vectors: within
    1e-6 of each tensor's largest magnitude (north_star's tolerance, read per tensor) and,
    element by element, within 1e-5 of the element's own magnitude plus one ulp of the tensor's
    scale.  Prints the worst element-wise relative error and ulp distance it saw."""
    for k in golden_util.worst_seen:
        golden_util.worst_seen[k] = 0
    for sc in S.all_scenarios():
        if S.reference_rtol(sc) is not None:
            golden_util.check_records(sc["id"], S.run_scenario(oracle_factory, sc), S.reference_rtol(sc))
    w = golden_util.worst_seen
    with capsys.disabled():
        print(f"\nsqrt classes vs reference: worst error {w['scale_rel']:.3g} of the tensor scale, "
              f"worst element-wise relative error {w['elem_rel']:.3g}, worst distance {w['ulp']} ulp, "
              f"{w['bound_frac']:.3g} of the element-wise bound")
    assert w["scale_rel"] <= 1e-6 and w["bound_frac"] <= 1.0 and w["ulp"] <= 4096


def test_golden_covers_every_scenario():
    _, index = golden_util.load()
    assert sorted(index) == sorted(s["id"] for s in S.all_scenarios())


def test_c_synth_equals_numpy_synth():
    base = synth.fill(10007, 7, first=3, scale=0.05)
    assert np.array_equal(base.view(np.uint32), OC.synth_fill(10007, 7, first=3, scale=0.05).view(np.uint32))
    a = synth.fill(10007, 99, first=11, scale=0.01, base=base)
    b = OC.synth_fill(10007, 99, first=11, scale=0.01, base=base)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _flat_case(n, K, seed=5):
    base = synth.fill(n, seed, scale=0.05)
    xs = [synth.fill(n, 100 + k, scale=0.02, base=base) for k in range(K)]
    xs[0][: min(n, 3)] = base[: min(n, 3)]
    return base, xs


@pytest.mark.parametrize("cls,opt", [("FedAvgAggregator", OC.OPT_NONE), ("FedAvgMAggregator", OC.OPT_AVGM),
                                     ("FedAdamAggregator", OC.OPT_ADAM), ("FedYogiAggregator", OC.OPT_YOGI),
                                     ("FedAdagradAggregator", OC.OPT_ADAGRAD)])
def test_c_oracle_equals_numpy_oracle_sync(cls, opt):
    """Three rounds on one flat tensor: the C restatement and the numpy restatement agree bit
    for bit, so the golden pin of the numpy oracle carries over to the C oracle."""
    n = 5003
    cfg = dict(S.HP[cls], client_weights_mode="sample_size")
    agg = OA.ORACLES[cls](None, cfg, None)
    sizes = {0: 5, 1: 11, 2: 2, 3: 40}
    for c, s in sizes.items():
        agg.set_client_sample_size(c, s)
    g = m = v = None
    for r, K in enumerate([2, 4, 3]):
        base, xs = _flat_case(n, K, seed=5 + r)
        cids = list(range(K))
        out = agg.aggregate({c: {"w": xs[c].copy()} for c in cids})
        if g is None:
            g = xs[0].copy()
            m = np.zeros(n, np.float32)
            tau = cfg.get("server_adapt_param", 0.0)
            v = np.full(n, np.float32(0.0 if opt == OC.OPT_AVGM else tau ** 2), np.float32)
        tot = sum(sizes[c] for c in cids)
        w = [sizes[c] / tot for c in cids]
        OC.reduce_opt(xs, w, g, m, v, opt, lr=cfg.get("server_learning_rate", 0.0),
                      beta1=cfg.get("server_momentum_param_1", 0.0),
                      beta2=cfg.get("server_momentum_param_2", 0.0),
                      tau=cfg.get("server_adapt_param", 0.0))
        assert np.array_equal(out["w"].view(np.uint32), g.view(np.uint32)), f"round {r}"


@pytest.mark.parametrize("gradient_based,sub_form", [(False, True), (True, True), (True, False)])
def test_c_oracle_equals_numpy_oracle_async(gradient_based, sub_form):
    n = 4001
    cfg = dict(S.STALENESS["polynomial"], alpha=0.9, num_clients=3, gradient_based=gradient_based,
               optimize_memory=sub_form)
    agg = OA.OracleFedAsync(None, cfg, None)
    g = None
    for i, cid in enumerate([0, 1, 0, 2, 1]):
        x = synth.fill(n, 50 + i, scale=0.02)
        stale = agg.global_step - agg.client_step.get(cid, 0)
        alpha_t = 0.9 * (stale + 1) ** (-0.5) * (1.0 / 3)
        out = agg.aggregate(cid, {"w": x.copy()})
        if g is None:
            g = x.copy()
        OC.async_mix(x, alpha_t, gradient_based, sub_form, g)
        assert np.array_equal(out["w"].view(np.uint32), g.view(np.uint32)), f"arrival {i}"


@pytest.mark.parametrize("cls,alg", [("IIADMMAggregator", "ii"), ("ICEADMMAggregator", "ice")])
def test_c_oracle_equals_numpy_oracle_admm(cls, alg):
    """Three ADMM rounds on a one-layer model: the flat C restatement reproduces the (golden-pinned)
    numpy restatement bit for bit; the residual norms agree to 1e-9 (float64 sums, other order)."""
    import torch
    model = S.LinearModel()
    agg = OA.ORACLES[cls](model, {}, None)
    names = [n for n, _ in model.named_parameters()]
    flat = lambda d: np.concatenate([OA.to_np(d[n]).ravel() for n in names])
    cids = [0, 1, 2]
    duals = [np.zeros(sum(model.state_dict()[n].numel() for n in names), np.float32) for _ in cids]
    prev = None
    for r in range(3):
        g = flat(model.state_dict()).copy()
        local, primal, penalty = {}, [], []
        for c in cids:
            st = S.make_state(S.LINEAR, seed=40 * r + c, huge=False)
            local[c] = {"primal": st, "penalty": 50.0 / (c + 1) + r}
            if alg == "ice":
                d = S.make_state(S.LINEAR, seed=40 * r + c + 9, gradient=True, scale=0.3, huge=False)
                local[c]["dual"] = d
                duals[c] = flat(d).copy()
            primal.append(flat(st).copy())
            penalty.append(local[c]["penalty"])
        out = agg.aggregate(local)
        res = OC.admm(alg, primal, duals, prev, penalty, len(cids), g)
        assert np.array_equal(flat(out).view(np.uint32), g.view(np.uint32)), f"round {r}"
        if alg == "ii":
            for c in cids:
                assert np.array_equal(flat(agg.dual_states[c]).view(np.uint32), duals[c].view(np.uint32))
        assert np.isclose(res[0], agg.prim_res, rtol=1e-9)
        assert np.isclose(res[1], agg.dual_res if r else 0.0, rtol=1e-9)
        prev = primal


@pytest.mark.parametrize("dtype", [np.float32, np.int64, np.int32])
def test_c_oracle_plain_mean(dtype):
    n, K = 101, 4
    if dtype == np.float32:
        xs = [synth.fill(n, 9 + k) for k in range(K)]
    else:
        xs = [synth.ints(9 + k, n, -9, 10, dtype) for k in range(K)]
        xs[0][:3], xs[1][:3], xs[2][:3], xs[3][:3] = [6, 7, -7], [0, 0, 0], [0, 0, 0], [0, 0, 0]
    agg = OA.OracleFedAvg(None, {}, None)
    agg.named_parameters = set()          # everything takes the plain-mean path
    agg.global_state = {"w": xs[0].copy()}
    out = agg.aggregate({k: {"w": xs[k]} for k in range(K)})
    g = np.zeros(n, dtype)
    OC.mean(xs, g)
    assert np.array_equal(out["w"], g)
    if dtype != np.float32:
        assert list(g[:3]) == [1, 1, -1]


class _NaiveAsAggregator:
    """oracle/torch_naive.py (what bench.py times as cpu_baseline) behind the aggregator
    interface of the scenarios: client weights as fedavg_aggregator.py:343-358."""

    def __init__(self, cls, cfg):
        from oracle import torch_naive as TN
        opt = {"FedAvgAggregator": "none", "FedAvgMAggregator": "avgm", "FedAdamAggregator": "adam",
               "FedYogiAggregator": "yogi", "FedAdagradAggregator": "adagrad"}[cls]
        self.naive, self.mode, self.sizes = TN.NaiveFedAvg(opt, cfg), cfg.get("client_weights_mode", "equal"), {}

    def set_client_sample_size(self, cid, n):
        self.sizes[cid] = n

    def aggregate(self, local_models, **kwargs):
        total = sum(self.sizes[c] for c in local_models if c in self.sizes)
        w = {c: (self.sizes[c] / total if self.mode == "sample_size" and c in self.sizes and total > 0
                 else 1.0 / len(local_models)) for c in local_models}
        return self.naive.aggregate(local_models, w)

    def get_parameters(self, **kwargs):
        return {k: v.clone() for k, v in self.naive.global_state.items()}

    @property
    def m_vector(self):
        return self.naive.m_vector

    @property
    def v_vector(self):
        return self.naive.v_vector


@pytest.mark.parametrize("sc", [s for s in S.sync_scenarios() if not s["model"] and not s.get("hash_only")],
                         ids=lambda s: s["id"])
def test_torch_naive_cpu_baseline_matches_reference_golden(sc):
    """The per-tensor torch loops that the benchmark times as `cpu_baseline` reproduce the
    reference's vectors: bit for bit where no sqrt is involved, within the sqrt tolerance else."""
    rec = S.run_scenario(lambda cls, model, cfg: _NaiveAsAggregator(cls, cfg), sc)
    golden_util.check_records(sc["id"], rec, S.reference_rtol(sc))


def test_torch_naive_async_mix_matches_the_pinned_oracle():
    import torch
    from oracle import torch_naive as TN
    cfg = dict(S.STALENESS["polynomial"], alpha=0.9, num_clients=3)
    ora = OA.OracleFedAsync(None, cfg, None)
    g = None
    for i, cid in enumerate([0, 1, 0, 2]):
        st = S.make_state(S.TOY, seed=300 + i)
        stale = ora.global_step - ora.client_step.get(cid, 0)
        alpha_t = 0.9 * (stale + 1) ** (-0.5) * (1.0 / 3)
        want = ora.aggregate(cid, {k: v.clone() for k, v in st.items()})
        if g is None:
            g = {k: v.clone() for k, v in st.items()}
            continue
        got = TN.naive_async_mix(g, {k: v.clone() for k, v in st.items()}, alpha_t)
        for k in want:
            assert golden_util.same_bits(got[k].numpy(), want[k]), (i, k)
