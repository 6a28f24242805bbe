"""State dicts in the shapes real models hand over: transposed (non-contiguous) tensors, tensors
that require grad, two names sharing one storage (tied weights), empty and zero-dimensional
entries -- the device aggregators against the oracle, which takes them as the reference does."""

import numpy as np
import pytest
import torch

import golden_util
import scenarios as S
from oracle import aggregators as OA
from oracle import synth

pytestmark = pytest.mark.gpu


def _client(seed, dtype=torch.float32):
    base = torch.from_numpy(synth.fill(6 * 5, 7, scale=0.05).reshape(6, 5))
    w = torch.from_numpy(synth.fill(6 * 5, seed, scale=0.02).reshape(6, 5)) + base
    emb = torch.from_numpy(synth.fill(12, seed + 1, scale=0.02).reshape(4, 3))
    st = {
        "w.t": w.t(),                                             # non-contiguous view
        "emb": emb.clone().requires_grad_(True),                  # a parameter as the model holds it
        "head": emb,                                              # tied to nothing here, tied below
        "empty": torch.zeros(0, 3),                               # an empty buffer
        "scale": torch.tensor(0.5 + 0.001 * seed),                # zero-dimensional float
        "steps": torch.tensor(seed % 5, dtype=torch.int64),
    }
    st["head_tied"] = st["head"]                                   # two names, one storage
    if dtype != torch.float32:
        st = {k: (v.detach().to(dtype) if v.dtype == torch.float32 else v) for k, v in st.items()}
        st["head_tied"] = st["head"]
    return st


@pytest.mark.parametrize("cls", ["FedAvgAggregator", "FedAdamAggregator", "FedAsyncAggregator", "FedBuffAggregator"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_odd_state_dicts(cls, dtype):
    import appfl_amd.aggregator as A
    cfg = dict(S.HP.get(cls, {}), alpha=0.9, num_clients=3, K=2, staleness_fn="polynomial",
               staleness_fn_kwargs={"a": 0.5})
    agg, ora = getattr(A, cls)(None, dict(cfg), None), OA.ORACLES[cls](None, dict(cfg), None)
    sync = cls in S.HP
    for r in range(3):
        if sync:
            local = {c: _client(10 * r + c, dtype) for c in range(3)}
            got = agg.aggregate(local)
            want = ora.aggregate({c: {k: v.detach() for k, v in m.items()} for c, m in local.items()})
        else:
            m = _client(10 * r, dtype)
            got = agg.aggregate(r % 3, m)
            want = ora.aggregate(r % 3, {k: v.detach() for k, v in m.items()})
        assert list(got) == list(want)
        rec_g, rec_w = S._snap(got), S._snap(want)
        for k in rec_w:
            assert rec_g[k].shape == rec_w[k].shape, (r, k)
            assert golden_util.same_bits(rec_g[k], rec_w[k]), (r, k)
        assert not got["emb"].requires_grad and got["w.t"].shape == (5, 6) and got["empty"].shape == (0, 3)
