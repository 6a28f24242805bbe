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


def test_state_dict_sharded_aggregator_single_rank_matches_plain_aggregator_and_oracle():
    import scenarios as S
    import appfl_amd.aggregator as A
    from appfl_amd.aggregator.sharded_aggregator import ShardedFedAvgAggregator
    cfg = dict(S.HP["FedYogiAggregator"], client_weights_mode="sample_size")
    plain = A.FedYogiAggregator(S.ToyModel(), cfg, None)
    shard = ShardedFedAvgAggregator(S.ToyModel(), cfg, None, algorithm="FedYogi", pieces=3)
    ora = OA.OracleFedYogi(S.ToyModel(), cfg, None)
    for agg in (plain, shard, ora):
        for c in range(4):
            agg.set_client_sample_size(c, 5 + 2 * c)
    first = shard.get_parameters()
    for k, t in ora.get_parameters().items():
        assert golden_util.same_bits(first[k].numpy(), np.asarray(t)), k
    for r, K in enumerate([3, 4, 2]):
        local = {c: S.make_state(S.TOY, seed=900 + 10 * r + c) for c in range(K)}
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
    assert golden_util.same_bits(gs["fc.weight"].cpu().numpy(), got["fc.weight"].numpy())
