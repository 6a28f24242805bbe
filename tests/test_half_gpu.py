"""bfloat16 / float16 state dicts on the device: the fa_half_* kernels (fp32 carriers, every
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


@pytest.mark.parametrize("kind", list(KINDS))
def test_device_rounding_equals_the_conversion_torch_does(kind):
    """g + 0 through fa_half_apply_step rounds every fp32 pattern the way torch's CPU conversion
    to the 16-bit type does: random bit patterns (denormals, infinities, NaNs among them), exact
    ties, the largest finite values."""
    rng = np.random.default_rng(5)
    pat = rng.integers(0, 2 ** 32, 1 << 16, dtype=np.uint64).astype(np.uint32)
    ties = (rng.integers(0, 2 ** 16, 4096, dtype=np.uint64).astype(np.uint32) << np.uint32(16)) | np.uint32(0x8000)
    edge = np.array([0x7F7FFFFF, 0xFF7FFFFF, 0x477FE000, 0x477FF000, 0x33800000, 0x33000000, 0x00000001,
                     0x80000001, 0x00800000, 0x7F800000, 0xFF800000, 0x7FC00000, 0x38800000, 0x387FC000],
                    np.uint32)
    x = np.concatenate([pat, ties, edge]).view(np.float32)
    g = torch.from_numpy(x.copy()).cuda()
    _hops(kind).apply_step(g, torch.zeros_like(g))
    got = g.cpu()
    xt = torch.from_numpy(x.copy())
    want = (xt + torch.zeros_like(xt)).to(KINDS[kind]).to(torch.float32)      # the fp32 sum, then the conversion
    gn, wn = got.numpy(), want.numpy()
    assert np.array_equal(np.isnan(gn), np.isnan(wn))
    ok = np.isnan(wn) | (gn.view(np.uint32) == wn.view(np.uint32))
    assert ok.all(), (x[~ok][:4], gn[~ok][:4], wn[~ok][:4])


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
        t = torch.zeros(a.size + 8, dtype=torch.float32, device="cuda")
        t[shift:shift + a.size].copy_(torch.from_numpy(a))
        return t[shift:shift + a.size]

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
        assert np.array_equal(_bits(g.cpu().numpy(), kind), want["w"].bits()), f"g, round {r}"
        if opt not in ("none", "avgm"):
            assert np.array_equal(_bits(m.cpu().numpy(), kind), ora.m_vector["w"].bits()), f"m, round {r}"
        if opt != "none":
            assert np.array_equal(_bits(v.cpu().numpy(), kind), ora.v_vector["w"].bits()), f"v, round {r}"


@pytest.mark.parametrize("kind", list(KINDS))
def test_step_then_optimizer_then_apply_equals_the_fused_call(kind):
    hops = _hops(kind)
    n, K = 10007, 5
    base, xs = _flat_clients(n, K, kind)
    w = [0.1, 0.3, 0.2, 0.25, 0.15]
    hp = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
    dx = [torch.from_numpy(x).cuda() for x in xs]
    for opt in ("none", "avgm", "adam", "yogi", "adagrad"):
        g1, g2 = torch.from_numpy(base).cuda(), torch.from_numpy(base).cuda()
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
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), opt


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
    with pytest.raises(NotImplementedError, match="FedCompass"):
        A.FedCompassAggregator(None, {"alpha": 0.9, "num_clients": 2}, None).aggregate(
            client_id=0, local_model=st, staleness=0)
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
    g = torch.from_numpy(xs[0]).cuda()
    m, v = torch.zeros_like(g), torch.empty_like(g)
    hops.fill(v, cfg["server_adapt_param"] ** 2)
    hops.reduce_opt([torch.from_numpy(x).cuda() for x in xs], [1.0 / K] * K, g, m, v, "adam",
                    lr=cfg["server_learning_rate"], beta1=cfg["server_momentum_param_1"],
                    beta2=cfg["server_momentum_param_2"], tau=cfg["server_adapt_param"])
    assert np.array_equal(_bits(g.cpu().numpy(), kind), want["w"].bits())
