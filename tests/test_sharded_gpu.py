"""The sharded aggregator on one GPU with the HIP backend (world size 1: both modes must equal
the plain fused round), complementing the 2-rank gloo rehearsal."""

import numpy as np
import pytest
import torch

from oracle import c_oracle as OC
from oracle import synth

pytestmark = pytest.mark.gpu
HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)


@pytest.mark.parametrize("mode", ["sliced", "whole"])
def test_single_rank_modes_match_oracle(mode):
    from appfl_amd.sharded import ShardedFlatAggregator
    n, K = 100003, 6
    base = synth.fill(n, 7, scale=0.05)
    g, m, v = base.copy(), np.zeros(n, np.float32), np.full(n, np.float32(HP["tau"] ** 2))
    agg = ShardedFlatAggregator(n, "adam", **HP)
    agg.load_global(torch.from_numpy(base).cuda())
    for r in range(3):
        xs = [synth.fill(n, 100 * r + k, scale=0.02, base=base) for k in range(K)]
        w = [1.0 / K] * K
        OC.reduce_opt(xs, w, g, m, v, OC.OPT_ADAM, **HP)
        dx = [torch.from_numpy(x).cuda() for x in xs]
        if mode == "sliced":
            agg.round_sliced(dx, w)
        else:
            agg.round_whole(dx, w)
    got = agg.all_gather_global().cpu().numpy()[:n]
    assert np.array_equal(got.view(np.uint32), g.view(np.uint32))
