"""Seeded random sweep of the C ABI against the C oracle: lengths around tile and vector
boundaries, client counts around the unroll / chunk limits, misaligned sub-buffers (the scalar
path), every optimizer, optional outputs, every reduce kernel.  Bit-exact throughout."""

import os

import numpy as np
import pytest
import torch

import golden_util
from appfl_amd import ops
from oracle import c_oracle as O

pytestmark = pytest.mark.gpu
# a one-off longer sweep: APPFL_AMD_FUZZ_OFFSET=1000 (2000, ...) moves every seeded case to fresh seeds
SEED_OFFSET = int(os.environ.get("APPFL_AMD_FUZZ_OFFSET", "0"))

OPTS = [("none", O.OPT_NONE), ("avgm", O.OPT_AVGM), ("adam", O.OPT_ADAM), ("yogi", O.OPT_YOGI),
        ("adagrad", O.OPT_ADAGRAD)]
LENGTHS = [1, 2, 3, 4, 5, 63, 64, 255, 256, 1023, 1024, 1025, 4095, 4099, 16384, 65537, 262147]
CLIENTS = [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 47, 49, 127, 128, 129, 140]


def _case(rng):
    n = int(rng.choice(LENGTHS)) + int(rng.integers(0, 3))
    K = int(rng.choice(CLIENTS))
    if n * K > 12_000_000:
        K = max(1, 12_000_000 // n)
    opt, code = OPTS[int(rng.integers(0, len(OPTS)))]
    shift = int(rng.integers(0, 4))          # element offset of every buffer: 0 keeps 16-byte alignment
    hp = dict(lr=float(rng.choice([0.1, 0.01, 1.0])), beta1=float(rng.choice([0.9, 0.5, 0.0])),
              beta2=float(rng.choice([0.99, 0.9])), tau=float(rng.choice([1e-3, 1e-8, 0.1])))
    w = rng.random(K) + 0.05
    w = (w / w.sum()).tolist() if rng.random() < 0.7 else w.tolist()
    return n, K, opt, code, shift, hp, w


@pytest.mark.parametrize("seed", range(40))
def test_random_reduce_rounds(seed):
    rng = np.random.default_rng(1234 + seed + SEED_OFFSET)
    try:
        for it in range(10):
            n, K, opt, code, shift, hp, w = _case(rng)
            # 0: default routing, 1: persistent kernel, 2: tile kernel, 3: wide-batch persistent
            # kernel where it applies (plain FedAvg), else the persistent kernel, 4: the persistent
            # kernel with the late optimizer epilogue where it applies (final form of an optimizer
            # round on 16-byte aligned buffers), else the tile kernel
            forced = int(rng.integers(0, 5))
            ops.tune(stash={0: 2, 1: -1, 2: 0, 3: -2, 4: -3}[forced])
            with_step = bool(rng.integers(0, 2))
            rounds = int(rng.integers(1, 3))
            base = O.synth_fill(n, 7 + seed, scale=0.05)
            g_ref, m_ref = base.copy(), np.zeros_like(base)
            v_ref = np.full_like(base, np.float32(0.0 if opt in ("none", "avgm") else hp["tau"] ** 2))
            step_ref = np.zeros_like(base)

            def shifted(a):
                buf = torch.zeros(a.size + 8, dtype=torch.float32, device="cuda")
                buf[shift:shift + a.size].copy_(torch.from_numpy(a))
                return buf[shift:shift + a.size]

            g, m, v = shifted(g_ref), shifted(m_ref), shifted(v_ref)
            step = shifted(step_ref) if with_step else None
            ws = shifted(step_ref) if (K > ops.kchunk() and not with_step) else None
            for r in range(rounds):
                xs = [O.synth_fill(n, 100 * it + 10 * r + k + 1, scale=0.02, base=base) for k in range(K)]
                if n > 4:
                    xs[0][1] = g_ref[1]
                    xs[K - 1][2] = np.float32(3e30 if r == 0 else -3e30)
                O.reduce_opt(xs, w, g_ref, m_ref, v_ref, code,
                             step_out=step_ref if with_step else None, **hp)
                ops.reduce_opt([shifted(x) for x in xs], w, g,
                               m if opt in ("adam", "yogi", "adagrad") else None,
                               v if opt != "none" else None, opt, step_out=step, step_ws=ws, **hp)
            what = f"seed {seed} case {it}: n={n} K={K} opt={opt} shift={shift} forced={forced}"
            assert golden_util.same_bits(g.cpu().numpy(), g_ref), what + " (g)"
            if opt != "none":
                assert golden_util.same_bits(v.cpu().numpy(), v_ref), what + " (v)"
            if opt in ("adam", "yogi", "adagrad"):
                assert golden_util.same_bits(m.cpu().numpy(), m_ref), what + " (m)"
            if with_step:
                assert golden_util.same_bits(step.cpu().numpy(), step_ref), what + " (step)"
    finally:
        ops.tune()


@pytest.mark.parametrize("seed", range(12))
def test_random_mixes_means_and_arrivals(seed):
    rng = np.random.default_rng(99 + seed + SEED_OFFSET)
    modes = [("replace", 0), ("convex", 1), ("subtract", 2), ("sum_add", 4), ("sum_set", 5)]
    for it in range(10):
        n = int(rng.choice(LENGTHS)) + int(rng.integers(0, 3))
        K = int(rng.choice(CLIENTS))
        if n * K > 8_000_000:
            K = max(1, 8_000_000 // n)
        base = O.synth_fill(n, 3 + seed, scale=0.05)
        xs = [O.synth_fill(n, 50 * it + k + 1, scale=0.02, base=base) for k in range(K)]
        dx = [torch.from_numpy(x).cuda() for x in xs]
        w = (rng.random(K) * 0.3).tolist()
        what = f"seed {seed} case {it}: n={n} K={K}"
        name, code = modes[int(rng.integers(0, len(modes)))]
        want = base.copy()
        O.weighted_mix(xs, w, want, code, 0.4)
        g = torch.from_numpy(base).cuda()
        ops.weighted_mix(dx, w, g, name, coeff=0.4, acc_ws=torch.empty_like(g))
        assert golden_util.same_bits(g.cpu().numpy(), want), what + f" (mix {name})"
        want = np.empty_like(base)
        O.mean(xs, want)
        g = torch.empty(n, device="cuda")
        ops.mean(dx, g)
        assert golden_util.same_bits(g.cpu().numpy(), want), what + " (mean)"
        alpha, grad = float(rng.random() * 0.5), bool(rng.integers(0, 2))
        want = base.copy()
        O.async_mix(xs[0], alpha, grad, True, want)
        g = torch.from_numpy(base).cuda()
        ops.async_mix(dx[0], alpha, g, gradient_based=grad)
        assert golden_util.same_bits(g.cpu().numpy(), want), what + " (async)"
        gw, sw = base.copy(), np.empty_like(base)
        g, s = torch.from_numpy(base).cuda(), torch.empty(n, device="cuda")
        B = min(K, 4)
        for j in range(B):
            O.buff_accum(xs[j], alpha, grad, gw, sw, j == 0, j == B - 1)
            ops.buff_accum(dx[j], alpha, g, s, first=(j == 0), gradient_based=grad, flush=(j == B - 1))
        assert golden_util.same_bits(g.cpu().numpy(), gw), what + " (buff g)"
        assert golden_util.same_bits(s.cpu().numpy(), sw), what + " (buff step)"


@pytest.mark.parametrize("seed", range(16))
def test_random_state_dict_rounds(seed):
    """Drop-in classes on random state-dict layouts (tensor counts, shapes, integer entries,
    client counts and sample sizes changing per round) against the numpy oracle."""
    import appfl_amd.aggregator as A
    import scenarios as S
    from oracle import aggregators as OA
    rng = np.random.default_rng(777 + seed + SEED_OFFSET)
    cls = ["FedAvgAggregator", "FedAvgMAggregator", "FedAdamAggregator", "FedYogiAggregator",
           "FedAdagradAggregator"][seed % 5]
    spec = {}
    for t in range(int(rng.integers(1, 9))):
        kind = "f32" if rng.random() < 0.75 else ("i64" if rng.random() < 0.6 else "i32")
        dims = int(rng.integers(0, 4))
        shape = tuple(int(rng.integers(1, 40)) for _ in range(dims))
        spec[f"layer{t}.{'weight' if kind == 'f32' else 'count'}"] = (shape, kind)
    spec["tail.weight"] = ((int(rng.choice([1, 5, 1027, 70001])),), "f32")
    mode = "sample_size" if seed % 2 else "equal"
    cfg = dict(S.HP[cls], client_weights_mode=mode)
    dev, ora = getattr(A, cls)(None, cfg, None), OA.ORACLES[cls](None, cfg, None)
    for agg in (dev, ora):
        for c in range(12):
            agg.set_client_sample_size(c, int(3 + 7 * c))
    for r in range(4):
        K = int(rng.integers(1, 12))
        ids = [int(c) for c in rng.permutation(12)[:K]]
        models = {c: S.make_state(spec, seed=1000 * seed + 20 * r + c) for c in ids}
        got = dev.aggregate(models)
        want = ora.aggregate({c: {k: v.clone() for k, v in m.items()} for c, m in models.items()})
        assert list(got) == list(want)
        for k in want:
            g = got[k].numpy()
            assert g.dtype == want[k].dtype and g.shape == want[k].shape, (seed, r, k)
            assert golden_util.same_bits(g, want[k]), f"seed {seed} round {r} {cls} entry {k}"
