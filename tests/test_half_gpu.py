"""bfloat16 / float16 state dicts on the device: the fa_half_* kernels (16-bit buffers, every
operation rounded to the 16-bit type) against the CPU oracle, whose 16-bit arithmetic is pinned
by golden vectors of the reference's classes (tests/golden/golden_half.npz; the scenario-level
comparisons run in test_aggregators_gpu.py)."""

import io

import numpy as np
import pytest
import torch

import scenarios as S
from oracle import aggregators as OA
from oracle import halfarray as H
from oracle import synth

pytestmark = pytest.mark.gpu

KINDS = {"bf16": torch.bfloat16, "f16": torch.float16}
CLS_OPT = {"FedAvgAggregator": "none", "FedAvgMAggregator": "avgm", "FedAdamAggregator": "adam",
           "FedYogiAggregator": "yogi", "FedAdagradAggregator": "adagrad"}


def _hops(kind):
    from appfl_amd import half_ops
    return half_ops.HalfOps(KINDS[kind])


def _carrier(a: np.ndarray, kind: str) -> np.ndarray:
    return H._ROUND[kind](np.asarray(a, np.float32))


def _bits(a: np.ndarray, kind: str) -> np.ndarray:
    return H.HalfArray(a, kind, True).bits()


def _tbits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _dev16(a: np.ndarray, kind: str, shift: int = 0) -> torch.Tensor:
    """`a` (values of the 16-bit type, held in fp32) as a device tensor of that type, `shift`
    elements past a 16-byte boundary."""
    t = torch.zeros(a.size + 16, dtype=KINDS[kind], device="cuda")
    t[shift:shift + a.size].copy_(torch.from_numpy(a))
    return t[shift:shift + a.size]


@pytest.mark.parametrize("kind", list(KINDS))
def test_device_rounding_equals_torchs_cpu_arithmetic(kind):
    """g + step through fa_half_apply_step against the same sum by torch's CPU kernel for the
    16-bit type: random bit patterns (denormals, infinities, NaNs, exact ties among the sums)."""
    rng = np.random.default_rng(5)
    dt = KINDS[kind]
    n = (1 << 17) + 5
    a = torch.from_numpy(rng.integers(0, 2 ** 16, n, dtype=np.uint64).astype(np.uint16).view(np.int16)).view(dt)
    b = torch.from_numpy(rng.integers(0, 2 ** 16, n, dtype=np.uint64).astype(np.uint16).view(np.int16)).view(dt)
    # half of the pairs within a few binades of each other, so that rounding (and ties) happens
    b[::2] = (a[::2].float() * torch.from_numpy(rng.choice([0.5, 3.0, -0.75, 1.0009765625, 257.0], n)[::2]).float()).to(dt)
    want = a + b
    g = a.cuda()
    _hops(kind).apply_step(g, b.cuda())
    got = g.cpu()
    nan = want.isnan()
    assert torch.equal(got.isnan(), nan)
    ok = nan | (got.view(torch.int16) == want.view(torch.int16))
    assert bool(ok.all()), (a[~ok][:4], b[~ok][:4], got[~ok][:4], want[~ok][:4])


def _flat_clients(n, K, kind, seed=5):
    base = _carrier(synth.fill(n, seed, scale=0.05), kind)
    xs = [_carrier(synth.fill(n, 100 + 3 * seed + k, scale=0.02, base=base), kind) for k in range(K)]
    xs[0][: min(n, 3)] = base[: min(n, 3)]
    return base, xs


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("cls", list(CLS_OPT))
@pytest.mark.parametrize("K,n,shift", [(1, 4099, 0), (7, 4099, 1), (8, 1 << 16, 0), (130, 5003, 0), (19, 3, 0)])
def test_flat_reduce_opt_matches_the_oracle(kind, cls, K, n, shift):
    """Three rounds of fa_half_reduce_opt on one flat vector against the oracle's aggregator on a
    one-entry 16-bit state dict: g, m and v identical in every bit.  `shift` = 1 puts every buffer
    4 bytes past a 16-byte boundary (scalar kernel); K = 130 takes two launches with a carried
    partial step."""
    dt = KINDS[kind]
    hops = _hops(kind)
    cfg = dict(S.HP[cls], client_weights_mode="sample_size")
    ora = OA.ORACLES[cls](None, cfg, None)
    sizes = {c: 3 + 5 * (c % 11) for c in range(K)}
    for c, sz in sizes.items():
        ora.set_client_sample_size(c, sz)
    tot = sum(sizes.values())
    w = [sizes[c] / tot for c in range(K)]
    opt = CLS_OPT[cls]
    hp = {}
    if opt != "none":
        hp["beta1"] = cfg["server_momentum_param_1"]
    if opt in ("adam", "yogi", "adagrad"):
        hp.update(lr=cfg["server_learning_rate"], tau=cfg["server_adapt_param"],
                  beta2=cfg.get("server_momentum_param_2", 0.0))

    def dev(a):
        return _dev16(a, kind, shift)

    g = m = v = None
    for r in range(3):
        base, xs = _flat_clients(n, K, kind, seed=5 + r)
        want = ora.aggregate({c: {"w": torch.from_numpy(xs[c]).to(KINDS[kind])} for c in range(K)})
        if g is None:
            g = dev(xs[0])
            if opt != "none":
                v = dev(np.zeros(n, np.float32))
                if opt != "avgm":
                    m = dev(np.zeros(n, np.float32))
                    hops.fill(v, cfg["server_adapt_param"] ** 2)
        ws = dev(np.zeros(n, np.float32)) if K > hops.kchunk() else None
        hops.reduce_opt([dev(x) for x in xs], w, g, m, v, opt, step_ws=ws, n=n, **hp)
        assert g.dtype == dt and np.array_equal(_tbits(g), want["w"].bits()), f"g, round {r}"
        if opt not in ("none", "avgm"):
            assert np.array_equal(_tbits(m), ora.m_vector["w"].bits()), f"m, round {r}"
        if opt != "none":
            assert np.array_equal(_tbits(v), ora.v_vector["w"].bits()), f"v, round {r}"


@pytest.mark.parametrize("kind", list(KINDS))
def test_step_then_optimizer_then_apply_equals_the_fused_call(kind):
    hops = _hops(kind)
    n, K = 10007, 5
    base, xs = _flat_clients(n, K, kind)
    w = [0.1, 0.3, 0.2, 0.25, 0.15]
    hp = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
    dx = [_dev16(x, kind) for x in xs]
    for opt in ("none", "avgm", "adam", "yogi", "adagrad"):
        g1, g2 = _dev16(base, kind), _dev16(base, kind)
        m1, m2 = torch.zeros_like(g1), torch.zeros_like(g1)
        v1 = torch.empty_like(g1)
        hops.fill(v1, 1e-6)
        v2 = v1.clone()
        step = torch.empty_like(g1)
        for _ in range(2):
            hops.reduce_opt(dx, w, g1, m1, v1, opt, **hp)
            hops.reduce_step(dx, w, g2, step)
            hops.server_opt_step(step, m2, v2, opt, **hp)
            hops.apply_step(g2, step)
        for a, b in ((g1, g2), (m1, m2), (v1, v2)):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), opt


@pytest.mark.parametrize("kind", list(KINDS))
def test_aggregator_hands_back_the_callers_dtype_and_exposes_narrowed_state(kind):
    import appfl_amd.aggregator as A
    dt = KINDS[kind]
    spec = S.SPECS[f"toy_{kind}"]
    agg = A.FedAdamAggregator(None, dict(S.HP["FedAdamAggregator"]), None)
    ora = OA.OracleFedAdam(None, dict(S.HP["FedAdamAggregator"]), None)
    local = {c: S.make_state(spec, seed=40 + c) for c in range(3)}
    keep = {c: {k: v.clone() for k, v in m.items()} for c, m in local.items()}
    out = agg.aggregate(local)
    want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in keep.items()})
    for k, t in out.items():
        assert not t.is_cuda and t.dtype == keep[0][k].dtype and t.shape == keep[0][k].shape
        assert torch.equal(local[0][k], keep[0][k])
    assert out["fc.weight"].dtype == dt
    for name, d in (("m", agg.m_vector), ("v", agg.v_vector), ("g", agg.global_state)):
        assert d["fc.weight"].dtype == dt, name
    assert np.array_equal(S._snap(agg.v_vector)["fc.weight"], ora.v_vector["fc.weight"].bits())
    assert np.array_equal(S._snap(out)["fc.weight"], want["fc.weight"].bits())
    agg.compute_steps(local)
    assert agg.step["fc.weight"].dtype == dt
    # chunked replies are sized by the caller's 2-byte elements
    agg.set_model_chunk_size(300)
    reply, meta = agg.get_parameters(_request_chunked=True)
    assert meta["_total_chunks"] == len(S.chunk_keys_by_size(spec, 300)) and reply[meta["_chunk_keys"][0]].dtype in (dt, torch.int64)


@pytest.mark.parametrize("kind", list(KINDS))
def test_stage_bytes_and_user_override_on_a_16_bit_model(kind):
    """The wire route (torch.save bytes -> stage_bytes) and the un-fused route (a subclass that
    overrides compute_steps) give the round the plain route gives."""
    import appfl_amd.aggregator as A
    spec = S.SPECS[f"toy_{kind}"]
    cfg = dict(S.HP["FedAvgMAggregator"])

    class Mine(A.FedAvgMAggregator):
        def compute_steps(self, local_models):
            super().compute_steps(local_models)
            self.seen = sorted(self.step)

    plain, wired, mine = A.FedAvgMAggregator(None, cfg, None), A.FedAvgMAggregator(None, cfg, None), Mine(None, cfg, None)
    for r in range(2):
        local = {c: S.make_state(spec, seed=70 + 10 * r + c) for c in range(3)}
        a = plain.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        staged = {}
        for c, m in local.items():
            buf = io.BytesIO()
            torch.save(m, buf)
            staged[c] = wired.stage_bytes(c, buf.getvalue())
        b = wired.aggregate(staged)
        c_ = mine.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in local.items()})
        for k in a:
            assert a[k].dtype == b[k].dtype == c_[k].dtype
            assert torch.equal(a[k].view(torch.int16) if a[k].dtype in KINDS.values() else a[k],
                               b[k].view(torch.int16) if b[k].dtype in KINDS.values() else b[k]), (r, k)
            assert torch.equal(a[k].view(torch.int16) if a[k].dtype in KINDS.values() else a[k],
                               c_[k].view(torch.int16) if c_[k].dtype in KINDS.values() else c_[k]), (r, k)
    assert mine.seen


def test_what_is_refused_says_so():
    import appfl_amd.aggregator as A
    st = S.make_state(S.SPECS["toy_bf16"], seed=3)
    with pytest.raises(NotImplementedError, match="sharded"):
        from appfl_amd.layout import Layout, WindowedLayout
        full = Layout([(n, tuple(t.shape), t.dtype) for n, t in st.items()], None)
        WindowedLayout(full, [(0, 8, 0)], 8)
    mixed = dict(st)
    mixed["fc.bias"] = mixed["fc.bias"].to(torch.float32)
    with pytest.raises(NotImplementedError, match="mixes"):
        A.FedAvgAggregator(None, {}, None).aggregate({0: mixed, 1: mixed})
    with pytest.raises(NotImplementedError):
        A.FedAvgAggregator(None, {}, None).aggregate({0: {"w": torch.zeros(4, dtype=torch.int8)}})


@pytest.mark.parametrize("kind", list(KINDS))
def test_large_round_on_16_bit_carriers(kind):
    """ResNet-18-sized vector, 12 clients, FedAdam: the 128-bit kernel over many blocks against the
    oracle (every 97th element and a checksum of all)."""
    hops = _hops(kind)
    n, K = 11_699_112 // 4, 12
    base, xs = _flat_clients(n, K, kind, seed=9)
    cfg = dict(S.HP["FedAdamAggregator"])
    ora = OA.OracleFedAdam(None, cfg, None)
    want = ora.aggregate({c: {"w": torch.from_numpy(xs[c]).to(KINDS[kind])} for c in range(K)})
    g = _dev16(xs[0], kind)
    m, v = torch.zeros_like(g), torch.empty_like(g)
    hops.fill(v, cfg["server_adapt_param"] ** 2)
    hops.reduce_opt([_dev16(x, kind) for x in xs], [1.0 / K] * K, g, m, v, "adam",
                    lr=cfg["server_learning_rate"], beta1=cfg["server_momentum_param_1"],
                    beta2=cfg["server_momentum_param_2"], tau=cfg["server_adapt_param"])
    assert np.array_equal(_tbits(g), want["w"].bits())
