"""ADMM server updates on the GPU: the flat kernels (fa_admm_ii / fa_admm_ice, through the C ABI)
against the C oracle -- tensors bit for bit, the two residual norms to 1e-9 (both sum the same
fp32 squares in float64, in different orders) -- and the drop-in classes' behaviour around them.
The classes against the reference's golden vectors: tests/test_aggregators_gpu.py."""

import numpy as np
import pytest
import torch

import scenarios as S
from oracle import aggregators as OA
from oracle import c_oracle as O
from test_kernels_gpu import assert_bits, dev

pytestmark = pytest.mark.gpu


def _rows(n, K, seed):
    base = O.synth_fill(n, seed, scale=0.05)
    primal = [O.synth_fill(n, seed + 10 + k, scale=0.02, base=base) for k in range(K)]
    dual = [O.synth_fill(n, seed + 500 + k, scale=0.5) for k in range(K)]
    prev = [O.synth_fill(n, seed + 900 + k, scale=0.02, base=base) for k in range(K)]
    if n > 8:
        primal[0][3] = base[3]                 # zero difference
        primal[K - 1][5] = -0.0
        dual[0][6] = 1e-41                     # denormal
        prev[0][7] = primal[0][7]              # zero dual-residual term
    return base, primal, dual, prev


@pytest.mark.parametrize("alg", ["ii", "ice"])
@pytest.mark.parametrize("n", [1, 3, 4, 257, 4099, 1000003, 9000011])   # the last: the long-vector launch shapes
@pytest.mark.parametrize("K,with_prev", [(1, True), (3, False), (7, True), (64, True), (65, True), (150, False)])
def test_admm_round_matches_oracle(alg, n, K, with_prev):
    from appfl_amd import _native as N
    from appfl_amd import ops
    if n == 1000003 and K > 7:
        pytest.skip("large vectors are covered with few clients")
    if n == 9000011 and K not in (3, 7):
        pytest.skip("the longest vector is covered with few clients")
    base, primal, dual, prev = _rows(n, K, seed=31 + K)
    penalty = [500.0 / (1 + k % 5) + 0.37 * k for k in range(K)]
    g_ref = base.copy()
    dual_ref = [d.copy() for d in dual]
    want = O.admm(alg, primal, dual_ref, prev if with_prev else None, penalty, K + 2, g_ref)
    g = dev(base)
    d_dev = [dev(d) for d in dual]
    norms = torch.zeros(2, dtype=torch.float64, device="cuda")
    res_ws = torch.empty(N.FA_ADMM_RES_WS, dtype=torch.float64, device="cuda")
    carry = torch.empty(ops.admm_carry_floats(n), dtype=torch.float32, device="cuda") if K > N.FA_ADMM_KCHUNK else None
    ops.admm_round(alg, [dev(x) for x in primal], d_dev, [dev(p) for p in prev] if with_prev else None,
                   penalty, g, norms, res_ws, num_clients=K + 2, carry_ws=carry)
    assert_bits(g, g_ref, "g")
    for k in range(K):
        assert_bits(d_dev[k], dual_ref[k], f"dual row {k}")      # ICEADMM: untouched
    got = norms.cpu().numpy()
    assert np.allclose(got, want, rtol=1e-9, atol=0.0), (got, want)
    if not with_prev:
        assert got[1] == 0.0


def test_admm_round_on_unaligned_rows_and_argument_errors():
    from appfl_amd import _native as N
    from appfl_amd import ops
    n, K = 1001, 3
    base, primal, dual, prev = _rows(n + 1, K, seed=77)
    g_ref = base[1:].copy()
    dual_ref = [d[1:].copy() for d in dual]
    penalty = [3.0, 0.5, 11.0]
    want = O.admm("ii", [x[1:].copy() for x in primal], dual_ref, [p[1:].copy() for p in prev], penalty, K, g_ref)
    g = dev(base)[1:]                            # 4-byte aligned only: the scalar kernel takes it
    d_dev = [dev(d)[1:] for d in dual]
    norms = torch.zeros(2, dtype=torch.float64, device="cuda")
    res_ws = torch.empty(N.FA_ADMM_RES_WS, dtype=torch.float64, device="cuda")
    ops.admm_round("ii", [dev(x)[1:] for x in primal], d_dev, [dev(p)[1:] for p in prev], penalty, g,
                   norms, res_ws, num_clients=K)
    assert_bits(g, g_ref, "g")
    assert_bits(d_dev[2], dual_ref[2], "dual row 2")
    assert np.allclose(norms.cpu().numpy(), want, rtol=1e-9)
    with pytest.raises(N.NativeLibraryError, match="workspace"):
        ops.admm_round("ice", [g] * 65, [g] * 65, None, [1.0] * 65, g, norms, res_ws)
    with pytest.raises(ValueError):
        ops.admm_round("ii", [g], [g, g], None, [1.0], g, norms, res_ws)


@pytest.mark.parametrize("cls", ["IIADMMAggregator", "ICEADMMAggregator"])
def test_admm_classes_keep_the_model_as_the_global_state(cls):
    """The model's parameters are re-read every round (a caller may have changed them), the other
    state-dict entries pass through, the output is fresh CPU tensors and inputs stay untouched."""
    import appfl_amd.aggregator as A
    model, ref_model = S.ToyModel(), S.ToyModel()
    agg, ora = getattr(A, cls)(model, {}, None), OA.ORACLES[cls](ref_model, {}, None)
    named = [n for n, _ in model.named_parameters()]

    def payload(seed):
        p = {"primal": S.make_state(S.TOY, seed=seed, huge=False), "penalty": 20.0 + seed % 7}
        if cls == "ICEADMMAggregator":
            d = S.make_state(S.TOY, seed=seed + 1, gradient=True, scale=0.3, huge=False)
            p["dual"] = {n: d[n] for n in named}
        return p

    for r in range(3):
        local = {c: payload(100 * r + i) for i, c in enumerate(("a", "b"))}
        keep = {c: {k: v.clone() for k, v in local[c]["primal"].items()} for c in local}
        if r == 2:                               # somebody edits the model between rounds
            for m in (model, ref_model):
                with torch.no_grad():
                    m.fc.bias.add_(0.25)
                    m.bn.running_mean.fill_(0.125)
        got, want = agg.aggregate(local), ora.aggregate(local)
        assert list(got) == list(model.state_dict())
        for k in want:
            assert not got[k].is_cuda
            assert got[k].numpy().tobytes() == want[k].tobytes(), (r, k)
            assert model.state_dict()[k].numpy().tobytes() == want[k].tobytes(), (r, k)
        assert np.isclose(agg.prim_res, ora.prim_res, rtol=1e-9) and np.isclose(agg.dual_res, ora.dual_res, rtol=1e-9)
        for c in local:
            for k, v in keep[c].items():
                assert torch.equal(local[c]["primal"][k], v)
    assert float(got["bn.running_mean"][0]) == 0.125
    got["fc.bias"].add_(1.0)                     # the caller may do what it likes with the result
    assert agg.get_parameters()["fc.bias"].numpy().tobytes() == want["fc.bias"].tobytes()
    assert set(agg.dual_states) == {"a", "b"} and agg.dual_states["a"]["fc.bias"].is_cuda
    for c in ("a", "b"):
        for k in named:
            assert agg.primal_states_curr[c][k].cpu().numpy().tobytes() == ora.primal_states_curr[c][k].tobytes()
            assert agg.primal_states_prev[c][k].cpu().numpy().tobytes() == ora.primal_states_prev[c][k].tobytes()
    # a client that sends nothing new: its previous primal is its current one afterwards
    got, want = agg.aggregate({"a": payload(900), "b": None}), ora.aggregate({"a": payload(900), "b": None})
    for k in want:
        assert got[k].numpy().tobytes() == want[k].tobytes(), k
    for c in ("a", "b"):
        for k in named:
            assert agg.primal_states_prev[c][k].cpu().numpy().tobytes() == ora.primal_states_prev[c][k].tobytes()
    assert agg.penalty == ora.penalty


def test_admm_errors():
    import appfl_amd.aggregator as A
    agg = A.IIADMMAggregator(S.ToyModel(), {}, None)
    good = {"primal": S.make_state(S.TOY, seed=1), "penalty": 5.0}
    with pytest.raises(KeyError):
        agg.aggregate({0: good, 1: None})        # nothing to stand by in the first round
    agg = A.IIADMMAggregator(S.ToyModel(), {}, None)
    agg.aggregate({0: good, 1: dict(good)})
    with pytest.raises(KeyError):
        agg.aggregate({0: good, 7: good})        # a client the first round did not have
    bad = {"primal": {k: v for k, v in good["primal"].items() if k != "fc.bias"}, "penalty": 5.0}
    with pytest.raises(KeyError, match="fc.bias"):
        agg.aggregate({0: bad})
    with pytest.raises(AttributeError):
        A.ICEADMMAggregator(None, {}, None)
    with pytest.raises(NotImplementedError):
        A.IIADMMAggregator(S.Mixed64Model(), {}, None)


def test_admm_many_clients_carry_the_sums_between_launches():
    import appfl_amd.aggregator as A
    K = 70
    model, ref_model = S.LinearModel(), S.LinearModel()
    agg, ora = A.IIADMMAggregator(model, {}, None), OA.OracleIIADMM(ref_model, {}, None)
    for r in range(2):
        local = {c: {"primal": S.make_state(S.LINEAR, seed=300 * r + c, huge=False), "penalty": 10.0 + c}
                 for c in range(K)}
        got, want = agg.aggregate(local), ora.aggregate(local)
        for k in want:
            assert got[k].numpy().tobytes() == want[k].tobytes(), (r, k)
        assert np.isclose(agg.prim_res, ora.prim_res, rtol=1e-9) and np.isclose(agg.dual_res, ora.dual_res, rtol=1e-9)


@pytest.mark.parametrize("cls", ["IIADMMAggregator", "ICEADMMAggregator"])
def test_admm_messages_staged_on_arrival_give_the_same_rounds(cls):
    import appfl_amd.aggregator as A
    model, ref_model = S.LinearModel(), S.LinearModel()
    agg, ora = getattr(A, cls)(model, {}, None), OA.ORACLES[cls](ref_model, {}, None)
    named = [n for n, _ in model.named_parameters()]

    def payload(seed):
        p = {"primal": S.make_state(S.LINEAR, seed=seed, huge=False), "penalty": 30.0 + seed % 5}
        if cls == "ICEADMMAggregator":
            d = S.make_state(S.LINEAR, seed=seed + 3, gradient=True, scale=0.3, huge=False)
            p["dual"] = {n: d[n] for n in named}
        return p

    for r in range(4):
        local = {c: payload(50 * r + c) for c in range(3)}
        if r == 3:
            local[1] = None                      # stands by its last message
        for c, m in local.items():
            if r != 2 or c != 0:                 # round 2: client 0 is NOT staged, the others are
                agg.stage(c, m)                  # (no-op in the first round)
        got, want = agg.aggregate(local), ora.aggregate(local)
        for k in want:
            assert got[k].numpy().tobytes() == want[k].tobytes(), (r, k)
        assert np.isclose(agg.prim_res, ora.prim_res, rtol=1e-9) and np.isclose(agg.dual_res, ora.dual_res, rtol=1e-9)


def test_iceadmm_message_staged_but_not_taken_leaves_the_clients_state_alone():
    """A message uploaded on arrival that the round then does not use (the scheduler hands `None`
    for that client, or another message object) must not reach the round: the reference keeps the
    client's previous primal AND dual (iceadmm_aggregator.py:66-70)."""
    import appfl_amd.aggregator as A
    model, ref_model = S.LinearModel(), S.LinearModel()
    agg, ora = A.ICEADMMAggregator(model, {}, None), OA.ORACLES["ICEADMMAggregator"](ref_model, {}, None)
    named = [n for n, _ in model.named_parameters()]

    def payload(seed):
        d = S.make_state(S.LINEAR, seed=seed + 3, gradient=True, scale=0.3, huge=False)
        return {"primal": S.make_state(S.LINEAR, seed=seed, huge=False), "penalty": 30.0 + seed % 5,
                "dual": {n: d[n] for n in named}}

    for r in range(4):
        local = {c: payload(60 * r + c) for c in range(3)}
        if r == 1:
            agg.stage(1, payload(999))           # staged, then the round stands by the old message
            local[1] = None
        if r == 2:
            agg.stage(2, payload(998))           # staged, then another message object arrives
        got, want = agg.aggregate(local), ora.aggregate(local)
        for k in want:
            assert got[k].numpy().tobytes() == want[k].tobytes(), (r, k)
        assert np.isclose(agg.prim_res, ora.prim_res, rtol=1e-9) and np.isclose(agg.dual_res, ora.dual_res, rtol=1e-9)
