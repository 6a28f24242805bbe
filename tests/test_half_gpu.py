"""bfloat16 / float16 state dicts on the device: the fa_half_* kernels (16-bit buffers, every
operation rounded to the 16-bit type) against the CPU oracle, whose 16-bit arithmetic is pinned
by golden vectors of the reference's classes (tests/golden/golden_half.npz; the scenario-level
comparisons run in test_aggregators_gpu.py)."""

import io

import os

import numpy as np
import pytest
import torch

import scenarios as S
from oracle import aggregators as OA
from oracle import halfarray as H
from oracle import synth

pytestmark = pytest.mark.gpu
# a one-off longer sweep: APPFL_AMD_FUZZ_OFFSET=1000 (2000, ...) moves every seeded case to fresh seeds
SEED_OFFSET = int(os.environ.get("APPFL_AMD_FUZZ_OFFSET", "0"))

KINDS = {"bf16": torch.bfloat16, "f16": torch.float16}
CLS_OPT = {"FedAvgAggregator": "none", "FedAvgMAggregator": "avgm", "FedAdamAggregator": "adam",
           "FedYogiAggregator": "yogi", "FedAdagradAggregator": "adagrad"}


def _hops(kind):
    from appfl_amd import half_ops
    return half_ops.HalfOps(KINDS[kind])


def _rounded(a: np.ndarray, kind: str) -> np.ndarray:
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
    base = _rounded(synth.fill(n, seed, scale=0.05), kind)
    xs = [_rounded(synth.fill(n, 100 + 3 * seed + k, scale=0.02, base=base), kind) for k in range(K)]
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
def test_aggregator_hands_back_the_callers_dtype_and_exposes_16_bit_state(kind):
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
    from appfl_amd.layout import Layout, WindowedLayout
    full = Layout([(n, tuple(t.shape), t.dtype) for n, t in st.items()], None)
    with pytest.raises(ValueError, match="16 bytes"):      # 2-byte elements: blocks of 8
        WindowedLayout(full, [(0, 4, 0)], 4)
    mixed = dict(st)
    mixed["fc.bias"] = mixed["fc.bias"].to(torch.float32)
    with pytest.raises(NotImplementedError, match="mixes"):
        A.FedAvgAggregator(None, {}, None).aggregate({0: mixed, 1: mixed})
    with pytest.raises(NotImplementedError):
        A.FedAvgAggregator(None, {}, None).aggregate({0: {"w": torch.zeros(4, dtype=torch.int8)}})


@pytest.mark.parametrize("kind", list(KINDS))
def test_large_round_on_16_bit_buffers(kind):
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


# -- seeded random sweeps: the flat entry points against PyTorch's own CPU arithmetic on 16-bit
# tensors (the engine the reference's aggregators run on), one torch operation per reference operation

LENGTHS = [1, 2, 7, 8, 9, 63, 64, 255, 2047, 2048, 2049, 4099, 65537]


def _rand16(rng, n, kind, scale=0.05):
    return torch.from_numpy((rng.standard_normal(n) * scale).astype(np.float32)).to(KINDS[kind])


def _shifted(t, shift):
    buf = torch.zeros(t.numel() + 16, dtype=t.dtype, device="cuda")
    buf[shift:shift + t.numel()].copy_(t)
    return buf[shift:shift + t.numel()]


def _same16(a, b, inputs=()):
    """Bit equality of two 16-bit tensors (all NaNs alike); on failure the first differing
    elements are printed with the matching elements of `inputs`."""
    a, b = a.detach().cpu(), b.detach().cpu()
    nan = b.isnan()
    ok = (nan & a.isnan()) | (~nan & ~a.isnan() & (a.view(torch.int16) == b.view(torch.int16)))
    if bool(ok.all()):
        return True
    bad = (~ok).nonzero().flatten()[:5]
    print("differing elements", bad.tolist(), "got", a[bad].float().tolist(), "want", b[bad].float().tolist(),
          "inputs", [t.detach().cpu()[bad].float().tolist() for t in inputs])
    return False


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("seed", range(6))
def test_random_arrivals_means_and_mixes(kind, seed):
    hops = _hops(kind)
    rng = np.random.default_rng(77 + seed + SEED_OFFSET)
    for it in range(12):
        n = int(rng.choice(LENGTHS)) + int(rng.integers(0, 3))
        shift = int(rng.integers(0, 3)) * int(rng.integers(0, 2))      # 0, or 1-2 elements off alignment
        what = f"{kind} seed {seed} case {it}: n={n} shift={shift}"
        g = _rand16(rng, n, kind)
        x = _rand16(rng, n, kind, 0.06)
        if n > 4:
            x[1] = g[1]
            x[2] = 3.0e30 if it % 3 == 0 else -0.0
        alpha = float(rng.random() * 0.5 + 0.01)
        # FedAsync (fedasync_aggregator.py:243-264, :156-160)
        for gb, sub in ((False, True), (False, False), (True, True), (True, False)):
            if sub:
                if gb:
                    step = torch.zeros_like(x).sub_(x, alpha=alpha)
                else:
                    step = torch.zeros_like(x).add_(x.clone().sub_(g, alpha=1), alpha=alpha)
            else:
                step = alpha * (-x) if gb else alpha * (x - g)
            want = g.clone().add_(step, alpha=1)
            gd, sd = _shifted(g, shift), _shifted(torch.zeros_like(g), shift)
            hops.async_mix(_shifted(x, shift), alpha, gd, gradient_based=gb, sub_form=sub, step_out=sd, n=n)
            if sub:
                # add_ / sub_(..., alpha=) is the one operation where PyTorch's own CPU result depends
                # on WHERE in the tensor an element sits: its vectorised body evaluates 0 +/- alpha*x in
                # fp32 and rounds once (a product that underflows keeps its sign: -0), its scalar tail
                # rounds the product first and then adds it to zero (+0).  The kernels follow the
                # body; the step is compared with the two zeros taken as equal.
                step = step + torch.zeros_like(step)
                sd.add_(torch.zeros_like(sd))
            assert _same16(gd, want, (x, g)) and _same16(sd, step, (x, g)), what + f" async gb={gb} sub={sub} alpha={alpha!r}"
        # FedBuff (fedbuff_aggregator.py:145-158, :75-88): three arrivals, the last one flushes
        for gb in (False, True):
            buf_ref = torch.zeros_like(g)
            gd, bd = _shifted(g, shift), _shifted(torch.full_like(g, 7.0), shift)
            g_ref = g.clone()
            for a_i in range(3):
                xa = _rand16(rng, n, kind, 0.06)
                buf_ref += (alpha * (-xa) if gb else alpha * (xa - g_ref))
                hops.buff_accum(_shifted(xa, shift), alpha, gd, bd, a_i == 0, gradient_based=gb, flush=a_i == 2, n=n)
            g_ref = g_ref + buf_ref
            assert _same16(bd, buf_ref) and _same16(gd, g_ref), what + f" buff gb={gb}"
        # plain mean and FedBuff's raw sums (fedavg_aggregator.py:285-295, fedbuff_aggregator.py:146-152, :84-86)
        K = int(rng.choice([1, 3, 8, 17, 130]))
        xs = [_rand16(rng, n, kind, 0.5) for _ in range(K)]
        s = torch.zeros_like(g)
        for t in xs:
            s.add_(t)
        md = _shifted(torch.zeros_like(g), shift)
        hops.mean([_shifted(t, shift) for t in xs], md, n=n)
        assert _same16(md, s.clone().div_(K)), what + f" mean K={K}"
        rd = _shifted(torch.full_like(g, 3.0), shift)
        for i, t in enumerate(xs[:3]):
            hops.buff_accum_raw(_shifted(t, shift), rd, i == 0, n=n)
        raw = torch.zeros_like(g)
        for t in xs[:3]:
            raw += t
        assert _same16(rd, raw), what + " raw sum"
        fd = _shifted(torch.zeros_like(g), shift)
        hops.buff_flush_mean(fd, rd, 3, n=n)
        assert _same16(fd, torch.div(raw, 3)), what + " flush mean"
        # FedCompass / FedQueue mixes (fedcompass_aggregator.py:152-338, fedqueue_aggregator.py:75-131)
        w = (rng.random(K) * 0.2 + 0.01).tolist()
        coeff = 1.0 - sum(w)
        acc = torch.zeros_like(g)
        for t, wk in zip(xs, w):
            acc = acc + t * wk
        sub_ref = g.clone()
        for t, wk in zip(xs, w):
            sub_ref = sub_ref - t * wk
        # the flat-payload sums start from the first term, not from a zero tensor
        # (fedavg_aggregator.py:180-217): a first product that underflows to -0 stays -0
        acc1 = xs[0] * w[0]
        for t, wk in zip(xs[1:], w[1:]):
            acc1 = acc1 + t * wk
        cases = {"replace": acc, "convex": g * coeff + acc, "subtract": sub_ref, "sum_add": g + acc1, "sum_set": acc1}
        if K == 1:
            cases["convex_one"] = g * coeff + xs[0] * w[0]
        for mode, want in cases.items():
            gd = _shifted(g, shift)
            ws = _shifted(torch.zeros_like(g), shift) if K > hops.kchunk() and mode != "subtract" else None
            hops.weighted_mix([_shifted(t, shift) for t in xs], w, gd, mode, coeff, acc_ws=ws, n=n)
            assert _same16(gd, want), what + f" mix {mode} K={K}"


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("seed", range(6))
def test_random_reduce_rounds(kind, seed):
    """Random lengths around the 8-element vector and tile boundaries, client counts around the
    batch and launch limits, misaligned buffers, every optimizer, random hyper-parameters and
    weights: two rounds of fa_half_reduce_opt against the oracle's aggregator, every bit of g, m, v
    and of the step handed out."""
    hops = _hops(kind)
    rng = np.random.default_rng(4321 + seed + SEED_OFFSET)
    classes = list(CLS_OPT)
    for it in range(8):
        n = int(rng.choice(LENGTHS)) + int(rng.integers(0, 3))
        K = int(rng.choice([1, 2, 7, 8, 9, 16, 33, 128, 129]))
        if n * K > 3_000_000:
            K = max(1, 3_000_000 // n)
        cls = classes[int(rng.integers(0, len(classes)))]
        opt = CLS_OPT[cls]
        shift = int(rng.integers(0, 3)) * int(rng.integers(0, 2))
        cfg = {"client_weights_mode": "sample_size",
               "server_learning_rate": float(rng.choice([0.1, 0.01, 1.0])),
               "server_momentum_param_1": float(rng.choice([0.9, 0.5, 0.0])),
               "server_momentum_param_2": float(rng.choice([0.99, 0.9])),
               "server_adapt_param": float(rng.choice([1e-3, 1e-8, 0.1]))}
        ora = OA.ORACLES[cls](None, cfg, None)
        sizes = {c: int(rng.integers(1, 50)) for c in range(K)}
        for c, sz in sizes.items():
            ora.set_client_sample_size(c, sz)
        tot = sum(sizes.values())
        w = [sizes[c] / tot for c in range(K)]
        hp = {}
        if opt != "none":
            hp["beta1"] = cfg["server_momentum_param_1"]
        if opt in ("adam", "yogi", "adagrad"):
            hp.update(lr=cfg["server_learning_rate"], tau=cfg["server_adapt_param"],
                      beta2=cfg["server_momentum_param_2"])
        what = f"{kind} seed {seed} case {it}: n={n} K={K} opt={opt} shift={shift}"
        g = m = v = None
        for r in range(2):
            base, xs = _flat_clients(n, K, kind, seed=11 * seed + 3 * it + r)
            if n > 4:
                xs[K - 1][2] = np.float32(3.0e30 if r == 0 else -3.0e30)     # inf in float16
                xs[K - 1] = _rounded(xs[K - 1], kind)
            want = ora.aggregate({c: {"w": torch.from_numpy(xs[c]).to(KINDS[kind])} for c in range(K)})
            if g is None:
                g = _dev16(xs[0], kind, shift)
                if opt != "none":
                    v = _dev16(np.zeros(n, np.float32), kind, shift)
                    if opt != "avgm":
                        m = _dev16(np.zeros(n, np.float32), kind, shift)
                        hops.fill(v, cfg["server_adapt_param"] ** 2)
            step = _dev16(np.zeros(n, np.float32), kind, shift)
            hops.reduce_opt([_dev16(x, kind, shift) for x in xs], w, g, m, v, opt, step_out=step, n=n, **hp)
            assert _same16(g, want["w"].to_torch()), what + f" (g, round {r})"
            assert _same16(step, ora.last_step["w"].to_torch()), what + f" (step, round {r})"
            if opt not in ("none", "avgm"):
                assert _same16(m, ora.m_vector["w"].to_torch()), what + f" (m, round {r})"
            if opt != "none":
                assert _same16(v, ora.v_vector["w"].to_torch()), what + f" (v, round {r})"


def test_a_vector_longer_than_one_launch_segment():
    """2^30 + 4099 bfloat16 elements: the reduce call is cut into launches of at most 2^30 elements
    (write-through stores address their buffer through a 32-bit byte offset).  One client of weight
    1, plain FedAvg: g + (x - g), each operation rounded -- checked against the same two operations
    by torch on the device, over the cut and at both ends."""
    hops = _hops("bf16")
    n = (1 << 30) + 4099
    gen = torch.Generator(device="cuda").manual_seed(5)
    g = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_(0, 0.05, generator=gen)
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_(0, 0.05, generator=gen)
    spots = [slice(0, 70000), slice((1 << 30) - 70000, (1 << 30) + 4099)]
    want = [g[s] + (x[s] - g[s]) for s in spots]
    hops.reduce_opt([x], [1.0], g, n=n)
    torch.cuda.synchronize()
    for s, w_ in zip(spots, want):
        assert torch.equal(g[s].view(torch.int16), w_.view(torch.int16))


# -- stage-on-arrival for 16-bit models: the folding of early clients, the pipelined last arrival
# and the start on the rows that have landed take the same routes as for fp32

def _spec16(kind, n):
    return {"w": ((n,), kind), "steps": ((3,), "i64"), "b": ((7, 3), kind), "c32": ((5,), "i32")}


def _same_records(got, want, what):
    a, b = S._snap(got), S._snap(want)
    assert list(a) == list(b), what
    import golden_util
    for k in b:
        assert golden_util.same_bits(a[k], b[k]), (what, k)


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("cls", ["FedAvgAggregator", "FedYogiAggregator"])
def test_16_bit_round_after_the_last_arrival_runs_as_a_pipeline(kind, cls, monkeypatch):
    import appfl_amd.aggregator as A
    from appfl_amd.aggregator.fedavg_aggregator import FedAvgAggregator
    spec = _spec16(kind, 4_600_003)
    cfg = dict(S.HP[cls], client_weights_mode="sample_size")
    agg, ora = getattr(A, cls)(None, cfg, None), OA.ORACLES[cls](None, cfg, None)
    for a in (agg, ora):
        for c in range(4):
            a.set_client_sample_size(c, 3 + 2 * c)
    taken = []
    plain = FedAvgAggregator._round_pipelined

    def spy(self, K, weights):
        out = plain(self, K, weights)
        taken.append(out is not None)
        return out
    monkeypatch.setattr(FedAvgAggregator, "_round_pipelined", spy)
    monkeypatch.setattr(FedAvgAggregator, "STAGE_WINDOW", 2 << 20)      # several windows per 9 MB row
    for r, K in enumerate([3, 4, 2]):
        models = {c: S.make_state(spec, seed=60 * r + c) for c in range(K)}
        for c, m in models.items():
            agg.stage(c, m)
        got = agg.aggregate(models)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        _same_records(got, want, (kind, cls, r))
    assert taken[1:] == [True, True], taken


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("cls", ["FedAvgMAggregator", "FedAdamAggregator"])
def test_16_bit_clients_staged_on_arrival_are_folded_while_the_others_are_awaited(kind, cls, monkeypatch):
    import appfl_amd.aggregator as A
    from appfl_amd import half_ops
    from appfl_amd.aggregator.fedavg_aggregator import FedAvgAggregator
    spec = _spec16(kind, 4_600_003)
    K = 5
    cfg = dict(S.HP[cls], client_weights_mode="sample_size", num_clients=K)
    agg, ora = getattr(A, cls)(None, cfg, None), OA.ORACLES[cls](None, cfg, None)
    for a in (agg, ora):
        for c in range(K):
            a.set_client_sample_size(c, 3 + 2 * c)
    resumed = []
    plain = half_ops.HalfOps.round_sync

    def spy(self, *args, **kw):
        resumed.append(kw.get("k_done", 0))
        return plain(self, *args, **kw)
    monkeypatch.setattr(half_ops.HalfOps, "round_sync", spy)
    monkeypatch.setattr(FedAvgAggregator, "STAGE_WINDOW", 2 << 20)

    def one_round(r, ids):
        models = {c: S.make_state(spec, seed=70 * r + c) for c in ids}
        for c, m in models.items():
            agg.stage(c, m)
            torch.cuda.synchronize()
        del resumed[:]
        got = agg.aggregate(models)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        _same_records(got, want, (kind, cls, r))
        return list(resumed)

    for r in range(2):
        calls = one_round(r, range(K))
        assert calls and all(k == K - 1 for k in calls), (r, calls)
    calls = one_round(2, range(K - 1))          # not the configured round: the guess is dropped
    assert calls and all(k == 0 for k in calls), calls


@pytest.mark.parametrize("landed", ["all_but_last", "half", "one"])
def test_16_bit_round_starts_on_the_rows_that_have_landed(landed, monkeypatch):
    import appfl_amd.aggregator as A
    from appfl_amd.staging import ClientSlab

    def ready_prefix(self, rows):
        K = len(list(rows))
        return {"all_but_last": K - 1, "half": (K + 1) // 2, "one": 1}[landed]

    monkeypatch.setattr(ClientSlab, "ready_prefix", ready_prefix)
    spec = S.SPECS["toy_bf16"]
    cfg = dict(S.HP["FedAdamAggregator"], client_weights_mode="sample_size")
    model = S.MODELS["toy_bf16"]
    agg, ora = A.FedAdamAggregator(model(), cfg, None), OA.OracleFedAdam(model(), cfg, None)
    for a in (agg, ora):
        for c in range(6):
            a.set_client_sample_size(c, 3 + 2 * c)
    for r, K in enumerate([4, 6, 2]):
        models = {c: S.make_state(spec, seed=300 + 10 * r + c) for c in range(K)}
        truth = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        for c in list(models)[: K - 1]:
            agg.stage(c, models[c])
        _same_records(agg.aggregate(models), truth, (landed, r))


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("gradient_based", [False, True])
def test_16_bit_fedasync_pipelined_arrival_and_fedbuff_early_download(kind, gradient_based, monkeypatch):
    """The windowed upload / mix / download pipeline of large FedAsync arrivals and FedBuff's early
    download, forced onto a small 16-bit model with tiny windows: the oracle's records."""
    import appfl_amd.aggregator as A
    from appfl_amd import staging
    spec = S.SPECS[f"ragged_{kind}"]
    cfg = dict(S.STALENESS["polynomial"], alpha=0.9, num_clients=3, gradient_based=gradient_based)
    piped, ora = A.FedAsyncAggregator(None, cfg, None), OA.OracleFedAsync(None, cfg, None)
    piped.PIPELINE_MIN_ELEMENTS = 0
    taken = []
    plain_pipe = A.FedAsyncAggregator._arrival_pipelined

    def spy(self, local_model, alpha_t):
        out = plain_pipe(self, local_model, alpha_t)
        taken.append(out is not None)
        return out
    monkeypatch.setattr(A.FedAsyncAggregator, "_arrival_pipelined", spy)
    monkeypatch.setattr(staging, "UPLOAD_WINDOW", 64 << 10)
    for i in range(5):
        st = S.make_state(spec, seed=800 + 13 * i, gradient=gradient_based, scale=0.01 if gradient_based else 0.02)
        got = piped.aggregate(i % 3, {k: v.clone() for k, v in st.items()})
        want = ora.aggregate(i % 3, {k: v.clone() for k, v in st.items()})
        _same_records(got, want, (kind, "async", i))
    assert all(taken), taken
    cfgb = dict(cfg, K=3)
    early, orab = A.FedBuffAggregator(None, cfgb, None), OA.OracleFedBuff(None, cfgb, None)
    early.PIPELINE_MIN_ELEMENTS = 0
    for i in range(7):
        st = S.make_state(spec, seed=900 + 11 * i, gradient=gradient_based, scale=0.01 if gradient_based else 0.02)
        got = early.aggregate(i % 3, {k: v.clone() for k, v in st.items()})
        want = orab.aggregate(i % 3, {k: v.clone() for k, v in st.items()})
        _same_records(got, want, (kind, "buff", i))


def test_the_16_bit_entry_points_refuse_bad_arguments_with_their_error_codes():
    """Straight through the C ABI: NULL buffers, unknown type / optimizer codes, no clients, odd
    addresses, a client list longer than one launch without a workspace -- each answered with
    the documented code before anything is launched (the buffers keep their contents)."""
    import ctypes as C

    from appfl_amd import _native as N
    lib = N.load()
    INVALID, ALIGN, WORKSPACE = -1, -2, -3
    n, K = 64, 3
    g = torch.full((n,), 1.5, dtype=torch.bfloat16, device="cuda")
    m, v, step = torch.zeros_like(g), torch.zeros_like(g), torch.zeros_like(g)
    xs = [torch.full((n,), float(k), dtype=torch.bfloat16, device="cuda") for k in range(K)]
    x, w = N.ptr_array([t.data_ptr() for t in xs]), N.double_array([1.0 / K] * K)
    P = lambda t: C.c_void_p(t.data_ptr())                      # noqa: E731
    BF = N.FA_HALF_BF16
    adam = (0.1, 0.9, 0.99, 1e-3)

    def reduce_opt(x_=x, w_=w, K_=K, g_=P(g), m_=P(m), v_=P(v), opt=2, step_out=None, ws=None, half=BF):
        return lib.fa_half_reduce_opt(x_, w_, K_, g_, m_, v_, n, opt, *adam, step_out, ws, half, None)

    assert reduce_opt() == 0
    torch.cuda.synchronize()
    before = g.clone()
    assert reduce_opt(x_=None) == INVALID and reduce_opt(w_=None) == INVALID and reduce_opt(g_=None) == INVALID
    assert reduce_opt(K_=0) == INVALID and reduce_opt(half=7) == INVALID and reduce_opt(opt=9) == INVALID
    assert reduce_opt(v_=None) == INVALID and reduce_opt(m_=None) == INVALID         # the moments Adam needs
    assert reduce_opt(g_=C.c_void_p(g.data_ptr() + 1)) == ALIGN
    odd = N.ptr_array([xs[0].data_ptr(), xs[1].data_ptr() + 1, xs[2].data_ptr()])
    assert reduce_opt(x_=odd) == ALIGN
    holes = N.ptr_array([xs[0].data_ptr(), 0, xs[2].data_ptr()])
    assert reduce_opt(x_=holes) == INVALID
    from appfl_amd import ops
    many = ops.kchunk() + 1
    xm, wm = N.ptr_array([xs[0].data_ptr()] * many), N.double_array([1.0 / many] * many)
    assert reduce_opt(x_=xm, w_=wm, K_=many) == WORKSPACE
    assert reduce_opt(x_=xm, w_=wm, K_=many, ws=P(step)) == 0
    assert lib.fa_half_reduce_opt_ex(x, w, K, P(g), P(m), P(v), n, 2, *adam, None, None, BF, 8, None) == INVALID   # unknown flag
    assert lib.fa_half_reduce_opt_ex(x, w, K, P(g), P(m), P(v), n, 2, *adam, None, None, BF, 1, None) == 0
    torch.cuda.synchronize()
    g.copy_(before)
    # the other entry points
    assert lib.fa_half_reduce_step_resume(x, w, K, K, P(g), P(step), n, BF, None) == INVALID    # no client left
    assert lib.fa_half_reduce_step_resume(x, w, K, -1, P(g), P(step), n, BF, None) == INVALID
    assert lib.fa_half_reduce_opt_resume(x, w, K, 1, P(g), P(m), P(v), n, 2, *adam, None, BF, None) == WORKSPACE
    assert lib.fa_half_server_opt_step(P(step), None, P(v), n, 2, *adam, BF, None) == INVALID
    assert lib.fa_half_server_opt_step(P(step), P(m), P(v), n, 0, *adam, BF, None) == 0          # nothing to do
    assert lib.fa_half_apply_step(None, P(step), n, BF, None) == INVALID
    assert lib.fa_half_mean(x, 0, P(g), n, BF, None) == INVALID
    assert lib.fa_half_weighted_mix(x, w, K, P(g), 0.5, 9, None, n, BF, None) == INVALID         # unknown mode
    assert lib.fa_half_weighted_mix(x, w, K, P(g), 0.5, 3, None, n, BF, None) == INVALID         # convex_one: K = 1
    assert lib.fa_half_weighted_mix(xm, wm, many, P(g), 0.5, 1, None, n, BF, None) == WORKSPACE
    assert lib.fa_half_async_mix(None, 0.5, 0, 0, P(g), None, n, BF, None) == INVALID
    assert lib.fa_half_async_mix(P(xs[0]), 0.5, 0, 0, C.c_void_p(g.data_ptr() + 1), None, n, BF, None) == ALIGN
    assert lib.fa_half_buff_accum(P(xs[0]), 0.5, 0, P(g), None, 1, 0, n, BF, None) == INVALID
    assert lib.fa_half_buff_flush_mean(P(g), P(step), 0, n, BF, None) == INVALID
    assert lib.fa_half_fill(P(g), 1.0, n, 0, None) == INVALID
    torch.cuda.synchronize()
    assert torch.equal(g.view(torch.int16), before.view(torch.int16))
    # n = 0 is a valid empty call everywhere
    assert lib.fa_half_mean(x, K, P(g), 0, BF, None) == 0 and lib.fa_half_fill(P(g), 1.0, 0, BF, None) == 0


# -- the persistent kernel with the late epilogue ------------------------------------------------------

def _late_rounds(o, dt, opt, K, n, shift, rounds=2, persistent=True):
    gen = torch.Generator(device="cuda").manual_seed(n + K)
    buf = lambda: torch.zeros(n + 16, dtype=dt, device="cuda")[shift:shift + n]     # noqa: E731
    g = buf().copy_((torch.randn(n, device="cuda", generator=gen) * 0.05).to(dt))
    m, v = buf(), buf()
    if opt not in ("avgm", "none"):
        o.fill(v, 1e-6)
    xs = [buf().copy_((g.float() + torch.randn(n, device="cuda", generator=gen) * 0.02).to(dt)) for _ in range(K)]
    w = [(k + 1) / (K * (K + 1) / 2) for k in range(K)]
    for _ in range(rounds):
        o.reduce_opt(xs, w, g, None if opt in ("avgm", "none") else m, None if opt == "none" else v, opt,
                     lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3, persistent=persistent)
    torch.cuda.synchronize()
    return [t.clone().view(torch.int16) for t in (g, m, v)]


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("opt,K,n,shift", [("adam", 49, 3_000_011, 0), ("avgm", 33, 2_200_003, 8), ("yogi", 7, 700_001, 0),
                                           ("adagrad", 128, 300_007, 0), ("adam", 3, 1_048_576, 0),
                                           ("none", 37, 2_600_003, 0), ("none", 2, 40_009, 8)])
def test_late_epilogue_kernel_gives_the_bits_of_the_tile_kernel(kind, opt, K, n, shift, monkeypatch):
    """The persistent kernel forced on shapes of every kind (client counts that leave a remainder
    of the double-buffered batches, ragged lengths, a tile count that leaves CUs idle, buffers on
    other 16-byte boundaries) against the tile-per-block kernel: g, m and v bit for bit."""
    dt, o = KINDS[kind], _hops(kind)
    monkeypatch.setenv("APPFL_AMD_HALF_LATE", "0")
    want = _late_rounds(o, dt, opt, K, n, shift)
    monkeypatch.setenv("APPFL_AMD_HALF_LATE", "4")
    assert o.reduce_kernel_name(K, n, opt) == "reduce_late_half_kernel"
    got = _late_rounds(o, dt, opt, K, n, shift)
    for a, b, what in zip(got, want, "gmv"):
        assert torch.equal(a, b), (kind, opt, K, n, what)


@pytest.mark.parametrize("mode", ["1", "2", "3"])
def test_late_epilogue_kernel_in_phases_on_a_long_vector(mode, monkeypatch):
    """ResNet-50 length x 48 clients: more tiles per CU than a phase holds, so the blocks meet
    before their bursts (two staggered groups, no meetings, one chip-wide group): same bits,
    and the default routing takes this kernel here."""
    from appfl_amd import half_ops
    dt, o = torch.bfloat16, _hops("bf16")
    n, K = 25_557_032, 48
    monkeypatch.delenv("APPFL_AMD_HALF_LATE", raising=False)
    assert o.reduce_kernel_name(K, n, "adam") == "reduce_late_half_kernel"
    assert o.reduce_kernel_name(16, n, "adam") == "reduce_half_kernel"
    assert o.reduce_kernel_name(K, n, "none") == "reduce_late_half_kernel"
    assert o.reduce_kernel_name(12, n, "none") == "reduce_half_kernel"
    assert o.reduce_kernel_name(32, 8_000_000, "avgm") == "reduce_late_half_kernel"
    assert _hops("f16").reduce_kernel_name(K, n, "adam") == "reduce_late_half_kernel"
    assert _hops("f16").reduce_kernel_name(32, n, "adam") == "reduce_half_kernel"    # float16 breaks even later
    monkeypatch.setenv("APPFL_AMD_HALF_LATE", "0")
    assert o.reduce_kernel_name(K, n, "adam") == "reduce_half_kernel"
    want = _late_rounds(o, dt, "adam", K, n, 0)
    monkeypatch.setenv("APPFL_AMD_HALF_LATE", mode)
    got = _late_rounds(o, dt, "adam", K, n, 0)
    for a, b, what in zip(got, want, "gmv"):
        assert torch.equal(a, b), (mode, what)
    if mode == "1":      # float16, whose client terms run as native packed instructions there
        o16 = _hops("f16")
        for opt in ("yogi", "none"):
            monkeypatch.setenv("APPFL_AMD_HALF_LATE", "0")
            want16 = _late_rounds(o16, torch.float16, opt, 64, n, 0)
            monkeypatch.setenv("APPFL_AMD_HALF_LATE", "1")
            assert o16.reduce_kernel_name(64, n, opt) == "reduce_late_half_kernel"
            got16 = _late_rounds(o16, torch.float16, opt, 64, n, 0)
            for a, b, what in zip(got16, want16, "gmv"):
                assert torch.equal(a, b), ("f16", opt, what)
        monkeypatch.setenv("APPFL_AMD_HALF_LATE", mode)
    # the per-call routing word keeps a round off it (the sharded rounds beside a gather)
    kept_off = _late_rounds(o, dt, "adam", K, n, 0, persistent=False)
    for a, b, what in zip(kept_off, want, "gmv"):
        assert torch.equal(a, b), (mode, what, "kept off")
