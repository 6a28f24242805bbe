"""Parity at the full sizes of the benchmark configurations (GPU).

The synthetic generator is counter based, so the CPU oracle can produce any window
[first, first+len) of a client update without the rest: configurations 2 and 3 (ResNet-18 K=100,
ResNet-50 K=64 x 5 rounds) are compared over the WHOLE vector window by window, configurations 4
and 5 (ViT-B/16 arrivals, 162.5 M x K=32) over evenly spaced windows that include both ends,
plus size-independent properties on the whole vector (fixed point, independence of how the
parameter axis is cut into launches).  All comparisons are bit for bit.
"""

import numpy as np
import pytest
import torch

import golden_util
from appfl_amd import ops, workloads as W
from oracle import c_oracle as O

pytestmark = pytest.mark.gpu

BASE_SEED = 7
HP_ADAM = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)      # mnist/server_fedadam.yaml:53-59
HP_YOGI = dict(lr=0.01, beta1=0.9, beta2=0.99, tau=1e-3)     # mnist/server_fedyogi.yaml:53-59


def _windows(n, count, length):
    """`count` windows of `length` elements spread evenly over [0, n), first and last included."""
    if count * length >= n:
        return [(a, min(a + length, n)) for a in range(0, n, length)]
    starts = np.linspace(0, n - length, count).astype(np.int64)
    starts[1:-1] = starts[1:-1] // 4 * 4 + 1          # odd starts: windows are not tile aligned
    return [(int(a), int(a) + length) for a in starts]


def _device_rows(K, n, seeds, base, slab=None):
    if slab is None:
        slab = torch.empty((K, (n + 63) // 64 * 64), dtype=torch.float32, device="cuda")
    for k in range(K):
        ops.synth_fill(slab[k], seeds[k], scale=0.01, base=base, n=n)
    return slab


def _host_rows(a, b, seeds, base_w):
    return [O.synth_fill(b - a, s, first=a, scale=0.01, base=base_w) for s in seeds]


def _same(dev, a, b, want, what):
    got = dev[a:b].cpu().numpy()
    assert golden_util.same_bits(got, want), f"{what}: window [{a},{b}) differs from the oracle"


@pytest.mark.parametrize("mode", ["equal", "sample_size"])
def test_config2_resnet18_k100_whole_vector(mode):
    n, K = W.numel(W.resnet(18)), 100
    sizes = [100 + 37 * k for k in range(K)]
    w = [1.0 / K] * K if mode == "equal" else [s / sum(sizes) for s in sizes]
    seeds = [1000 + k for k in range(K)]
    g = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(g, BASE_SEED, scale=0.05)
    slab = _device_rows(K, n, seeds, g)
    ops.reduce_opt([slab[k] for k in range(K)], w, g, n=n)
    torch.cuda.synchronize()
    for a, b in _windows(n, 1 << 30, 1 << 21):
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        O.reduce_opt(_host_rows(a, b, seeds, base_w), w, base_w)
        _same(g, a, b, base_w, "g")


@pytest.mark.parametrize("opt,code,hp,full", [("adam", O.OPT_ADAM, HP_ADAM, True),
                                              ("yogi", O.OPT_YOGI, HP_YOGI, True)])
def test_config3_resnet50_k64_five_rounds(opt, code, hp, full):
    n, K, rounds = W.numel(W.resnet(50)), 64, 5
    w = [1.0 / K] * K
    base = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, BASE_SEED, scale=0.05)
    g, m = base.clone(), torch.zeros_like(base)
    v = torch.full_like(base, float(np.float32(hp["tau"] ** 2)))
    slab = None
    for r in range(rounds):
        slab = _device_rows(K, n, [2000 + 64 * r + k for k in range(K)], base, slab)
        ops.reduce_opt([slab[k] for k in range(K)], w, g, m, v, opt, n=n, **hp)
    torch.cuda.synchronize()
    wins = _windows(n, 1 << 30, 1 << 21) if full else _windows(n, 12, 1 << 18)
    for a, b in wins:
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        gw, mw = base_w.copy(), np.zeros_like(base_w)
        vw = np.full_like(base_w, np.float32(hp["tau"] ** 2))
        for r in range(rounds):
            xs = _host_rows(a, b, [2000 + 64 * r + k for k in range(K)], base_w)
            O.reduce_opt(xs, w, gw, mw, vw, code, **hp)
        _same(g, a, b, gw, "g")
        _same(m, a, b, mw, "m")
        _same(v, a, b, vw, "v")


def _alpha(i):
    # scripted staleness 0..15, polynomial a = 0.5, alpha = 0.9, 16 clients (mnist/server_fedasync.yaml:55-61)
    return 0.9 * ((i % 16) + 1) ** (-0.5) * (1.0 / 16)


@pytest.mark.parametrize("gradient_based", [False, True])
def test_config4_vitb16_fedasync_64_arrivals(gradient_based):
    n, arrivals = W.numel(W.vit_b_16()), 64
    base = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, BASE_SEED, scale=0.05)
    g, x = base.clone(), torch.empty_like(base)
    for i in range(arrivals):
        ops.synth_fill(x, 3000 + i, scale=0.01, base=base)
        ops.async_mix(x, _alpha(i), g, gradient_based=gradient_based)
    torch.cuda.synchronize()
    for a, b in _windows(n, 64, 1 << 17):
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        gw = base_w.copy()
        for i in range(arrivals):
            xw = O.synth_fill(b - a, 3000 + i, first=a, scale=0.01, base=base_w)
            O.async_mix(xw, _alpha(i), gradient_based, True, gw)
        _same(g, a, b, gw, "g")


def test_config4_vitb16_fedbuff_k16():
    n, arrivals, K = W.numel(W.vit_b_16()), 64, 16
    base = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, BASE_SEED, scale=0.05)
    g, x, step = base.clone(), torch.empty_like(base), torch.empty_like(base)
    for i in range(arrivals):
        ops.synth_fill(x, 4000 + i, scale=0.01, base=base)
        ops.buff_accum(x, _alpha(i), g, step, first=(i % K == 0), flush=(i % K == K - 1))
    torch.cuda.synchronize()
    for a, b in _windows(n, 64, 1 << 17):
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        gw, sw = base_w.copy(), np.empty_like(base_w)
        for i in range(arrivals):
            xw = O.synth_fill(b - a, 4000 + i, first=a, scale=0.01, base=base_w)
            O.buff_accum(xw, _alpha(i), False, gw, sw, i % K == 0, i % K == K - 1)
        _same(g, a, b, gw, "g")


def test_config5_flat_162m_k32_shard():
    """One GPU's share of the 1.3 B-parameter case: 162.5 M parameters x 32 clients (20.8 GB)."""
    n, K = 162_500_000, 32
    free, _ = torch.cuda.mem_get_info()
    if free < (K + 3) * n * 4 + (1 << 30):
        pytest.skip("not enough free device memory for the 20.8 GB slab")
    sizes = [100 + 37 * k for k in range(K)]
    w = [s / sum(sizes) for s in sizes]
    seeds = [5000 + k for k in range(K)]
    base = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, BASE_SEED, scale=0.05)
    slab = _device_rows(K, n, seeds, base)
    rows = [slab[k] for k in range(K)]
    g = base.clone()
    ops.reduce_opt(rows, w, g, n=n)
    # the same round cut into three launches at unaligned-to-tile (but 16-byte aligned) places
    g2 = base.clone()
    cuts = [0, 40_000_004, 40_000_004 + 1028, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        ops.reduce_opt([r[a:] for r in rows], w, g2[a:], n=b - a)
    assert torch.equal(g.view(torch.int32), g2.view(torch.int32)), "result depends on the launch partition"
    # clients equal to the global model leave it untouched, bit for bit
    g3 = base.clone()
    ops.reduce_opt([base] * K, w, g3, n=n)
    assert torch.equal(g3.view(torch.int32), base.view(torch.int32))
    torch.cuda.synchronize()
    for a, b in _windows(n, 96, 1 << 17):
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        O.reduce_opt(_host_rows(a, b, seeds, base_w), w, base_w)
        _same(g, a, b, base_w, "g")


@pytest.mark.parametrize("K", [2, 17], ids=["tile_kernel", "persistent_kernel"])
def test_vector_longer_than_one_launch_segment(K):
    """More than 2^29 elements: the library cuts the axis into segments of 2 GiB per launch;
    the seam and both ends must come out like everything else."""
    n = (1 << 29) + 70_003
    free, _ = torch.cuda.mem_get_info()
    if free < (K + 4) * n * 4 + (1 << 30):
        pytest.skip("not enough free device memory")
    w = [(1.0 + k) / sum(1.0 + j for j in range(K)) for k in range(K)]
    seeds = [6000 + k for k in range(K)]
    base = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, BASE_SEED, scale=0.05)
    slab = _device_rows(K, n, seeds, base)
    g, m = base.clone(), torch.zeros_like(base)
    v = torch.full_like(base, float(np.float32(HP_ADAM["tau"] ** 2)))
    ops.reduce_opt([slab[k] for k in range(K)], w, g, m, v, "adam", n=n, **HP_ADAM)
    torch.cuda.synchronize()
    seam = 1 << 29
    L = 1 << 15
    wins = _windows(n, 8, L) + [(seam - L, seam), (seam - 5, seam + L), (seam, seam + L)]
    for a, b in wins:
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        gw, mw = base_w.copy(), np.zeros_like(base_w)
        vw = np.full_like(base_w, np.float32(HP_ADAM["tau"] ** 2))
        O.reduce_opt(_host_rows(a, b, seeds, base_w), w, gw, mw, vw, O.OPT_ADAM, **HP_ADAM)
        _same(g, a, b, gw, "g")
        _same(m, a, b, mw, "m")
        _same(v, a, b, vw, "v")


def test_other_entry_points_across_the_segment_seam():
    n = (1 << 29) + 70_003
    free, _ = torch.cuda.mem_get_info()
    if free < 8 * n * 4:
        pytest.skip("not enough free device memory")
    K, seeds = 3, [6100, 6101, 6102]
    w = [0.2, 0.3, 0.1]
    base = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, BASE_SEED, scale=0.05)
    slab = _device_rows(K, n, seeds, base)
    rows = [slab[k] for k in range(K)]
    g_async = base.clone()
    ops.async_mix(rows[0], 0.05, g_async)
    g_mix = base.clone()
    ops.weighted_mix(rows, w, g_mix, "convex", coeff=0.4)
    g_mean = torch.empty_like(base)
    ops.mean(rows, g_mean)
    torch.cuda.synchronize()
    seam, L = 1 << 29, 1 << 14
    for a, b in [(0, L), (seam - L, seam + L), (n - L - 3, n)]:
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        xs = _host_rows(a, b, seeds, base_w)
        want = base_w.copy()
        O.async_mix(xs[0], 0.05, False, True, want)
        _same(g_async, a, b, want, "async_mix")
        want = base_w.copy()
        O.weighted_mix(xs, w, want, 1, 0.4)
        _same(g_mix, a, b, want, "weighted_mix")
        want = np.empty_like(base_w)
        O.mean(xs, want)
        _same(g_mean, a, b, want, "mean")


def test_config5_whole_1p3b_vector_on_one_gpu():
    """The 1.3 B-parameter, K = 32 case un-sharded (166 GB of client rows in one GPU's HBM):
    three launch segments; sampled windows including both seams and the tail."""
    n, K = 1_300_000_000, 32
    free, _ = torch.cuda.mem_get_info()
    if free < (K + 2) * n * 4 + (4 << 30):
        pytest.skip("not enough free device memory for 166 GB of client rows")
    w = [1.0 / K] * K
    seeds = [7000 + k for k in range(K)]
    base = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.synth_fill(base, BASE_SEED, scale=0.05)
    slab = _device_rows(K, n, seeds, base)
    g = base.clone()
    ops.reduce_opt([slab[k] for k in range(K)], w, g, n=n)
    torch.cuda.synchronize()
    L = 1 << 15
    wins = _windows(n, 40, L) + [(s - L, s + L) for s in (1 << 29, 1 << 30)]
    for a, b in wins:
        base_w = O.synth_fill(b - a, BASE_SEED, first=a, scale=0.05)
        O.reduce_opt(_host_rows(a, b, seeds, base_w), w, base_w)
        _same(g, a, b, base_w, "g")
    del slab, g, base
    torch.cuda.empty_cache()
