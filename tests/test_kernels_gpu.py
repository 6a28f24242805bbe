"""HIP kernels (through the C ABI) against the C oracle on the same seeded inputs: bit-exact."""

import numpy as np
import pytest
import torch

from oracle import c_oracle as O

pytestmark = pytest.mark.gpu

HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
HP_YOGI = dict(lr=0.01, beta1=0.9, beta2=0.99, tau=1e-3)


def dev(a):
    return torch.from_numpy(a).cuda()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else a.dtype)


def assert_bits(got, want, what):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    bad = np.nonzero(bits(got) != bits(want))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {want.size} differ, first at {bad[:5]}: " \
                          f"{got[bad[:5]]} vs {want[bad[:5]]}"


def make_clients(n, K, seed=7, spike=True):
    base = O.synth_fill(n, seed, scale=0.05)
    xs = [O.synth_fill(n, 1000 + k, scale=0.01, base=base) for k in range(K)]
    if spike and n > 16:
        xs[0][3] = base[3]          # zero difference
        xs[K - 1][5] = 1e30         # huge
        xs[0][7] = -0.0
        base[9] = 0.0
        for x in xs:
            x[9] = 0.0              # all-zero column: Yogi sign(0), step 0
        xs[0][11] = 1e-41           # denormal input
    return base, xs


@pytest.mark.parametrize("n", [1, 3, 4, 257, 1000003])
@pytest.mark.parametrize("K", [1, 2, 7, 9, 100])
def test_reduce_apply_matches_oracle(n, K):
    from appfl_amd import ops
    base, xs = make_clients(n, K)
    w = [(3 + 2 * k) / sum(3 + 2 * j for j in range(K)) for k in range(K)]
    g_ref = base.copy()
    step_ref = np.empty_like(base)
    O.reduce_opt(xs, w, g_ref, step_out=step_ref)
    g = dev(base)
    step = torch.empty_like(g)
    ops.reduce_opt([dev(x) for x in xs], w, g, step_out=step)
    assert_bits(step, step_ref, "step")
    assert_bits(g, g_ref, "g")


@pytest.mark.parametrize("opt,hp", [("avgm", HP), ("adam", HP), ("yogi", HP_YOGI), ("adagrad", HP)])
@pytest.mark.parametrize("n", [5, 4099])
def test_optimizer_rounds_match_oracle(opt, hp, n):
    """Three consecutive rounds so m / v state carries over; K changes between rounds."""
    from appfl_amd import ops
    code = {"avgm": O.OPT_AVGM, "adam": O.OPT_ADAM, "yogi": O.OPT_YOGI, "adagrad": O.OPT_ADAGRAD}[opt]
    base, _ = make_clients(n, 1)
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v0 = np.float32(0.0) if opt == "avgm" else np.float32(hp["tau"] ** 2)
    v_ref = np.full_like(base, v0)
    g, m, v = dev(base), dev(m_ref), dev(v_ref)
    for rnd, K in enumerate([4, 7, 2]):
        xs = [O.synth_fill(n, 50 * rnd + k, scale=0.02, base=g_ref) for k in range(K)]
        if n > 4:
            for x in xs:
                x[2] = g_ref[2]     # exact zero pseudo-gradient
        w = [1.0 / K] * K
        O.reduce_opt(xs, w, g_ref, m_ref, v_ref, code, **hp)
        ops.reduce_opt([dev(x) for x in xs], w, g, m, v, opt, **hp)
        assert_bits(v, v_ref, f"v round {rnd}")
        if opt != "avgm":
            assert_bits(m, m_ref, f"m round {rnd}")
        assert_bits(g, g_ref, f"g round {rnd}")


def test_many_clients_use_workspace():
    """K > FA_KCHUNK runs as several launches carrying the partial step; same bits."""
    from appfl_amd import ops
    n, K = 1027, 2 * ops.kchunk() + 3
    base, xs = make_clients(n, K, spike=False)
    w = [1.0 / K] * K
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v_ref = np.full_like(base, np.float32(1e-6))
    O.reduce_opt(xs, w, g_ref, m_ref, v_ref, O.OPT_ADAM, **HP)
    g, m, v = dev(base), dev(np.zeros_like(base)), dev(np.full_like(base, np.float32(1e-6)))
    dx = [dev(x) for x in xs]
    with pytest.raises(Exception):
        ops.reduce_opt(dx, w, g, m, v, "adam", **HP)
    ops.reduce_opt(dx, w, g, m, v, "adam", step_ws=torch.empty_like(g), **HP)
    assert_bits(g, g_ref, "g")
    assert_bits(m, m_ref, "m")
    assert_bits(v, v_ref, "v")


def test_unaligned_buffers_take_scalar_path():
    from appfl_amd import ops
    n, K = 1001, 5
    base, xs = make_clients(n, K)
    w = [1.0 / K] * K
    g_ref = base.copy()
    O.reduce_opt(xs, w, g_ref)
    gbuf = torch.zeros(n + 1, device="cuda")
    g = gbuf[1:]
    g.copy_(dev(base))
    dx = []
    for k, x in enumerate(xs):
        b = torch.zeros(n + 3, device="cuda")
        v = b[k % 3:k % 3 + n]
        v.copy_(dev(x))
        dx.append(v)
    ops.reduce_opt(dx, w, g)
    assert_bits(g, g_ref, "g")


def test_reduce_step_and_apply():
    from appfl_amd import ops
    n, K = 70001, 6
    base, xs = make_clients(n, K)
    w = [1.0 / K] * K
    step_ref = np.empty_like(base)
    O.reduce_step(xs, w, base, step_ref)
    g = dev(base)
    step = torch.empty_like(g)
    ops.reduce_step([dev(x) for x in xs], w, g, step)
    assert_bits(step, step_ref, "step")
    assert_bits(g, base, "g untouched")
    g_ref = base.copy()
    O.apply_step(g_ref, step_ref)
    ops.apply_step(g, step)
    assert_bits(g, g_ref, "g")


def test_slab_entry_point():
    from appfl_amd import ops
    n, K, stride = 5003, 4, 5008
    base, xs = make_clients(n, K)
    w = [0.1, 0.2, 0.3, 0.4]
    g_ref = base.copy()
    O.reduce_opt(xs, w, g_ref)
    slab = torch.zeros(K * stride, device="cuda")
    for k, x in enumerate(xs):
        slab[k * stride:k * stride + n].copy_(dev(x))
    g = dev(base)
    ops.reduce_opt_slab(slab, stride, w, g)
    assert_bits(g, g_ref, "g")


@pytest.mark.parametrize("dtype", [np.float32, np.int64, np.int32])
@pytest.mark.parametrize("K", [1, 3, 4, 7, 130])
def test_plain_mean(dtype, K):
    from appfl_amd import ops
    n = 1001
    rng = np.random.default_rng(K)
    if dtype == np.float32:
        xs = [O.synth_fill(n, 10 + k, scale=1.0) for k in range(K)]
    else:
        xs = [rng.integers(-1000, 1000, n).astype(dtype) for _ in range(K)]
        xs[0][:3] = [6, 7, -7]
        for x in xs[1:]:
            x[:3] = 0
    g_ref = np.zeros(n, dtype)
    O.mean(xs, g_ref)
    g = torch.zeros(n, dtype=getattr(torch, np.dtype(dtype).name), device="cuda")
    ops.mean([dev(x) for x in xs], g)
    assert_bits(g, g_ref, "mean")


@pytest.mark.parametrize("gradient_based,sub_form", [(False, True), (True, True), (True, False)])
def test_async_mix(gradient_based, sub_form):
    from appfl_amd import ops
    n = 100003
    base, xs = make_clients(n, 3)
    g_ref = base.copy()
    g = dev(base)
    for k, alpha in enumerate([0.9 * 0.25, 0.9 * 2 ** -0.5 * 0.25, 0.9 / 31.0]):
        step_ref = np.empty_like(base)
        O.async_mix(xs[k], alpha, gradient_based, sub_form, g_ref, step_ref)
        step = torch.empty_like(g)
        ops.async_mix(dev(xs[k]), alpha, g, gradient_based, sub_form, step_out=step)
        assert_bits(step, step_ref, f"step {k}")
        assert_bits(g, g_ref, f"g {k}")


@pytest.mark.parametrize("gradient_based", [False, True])
def test_buff_accumulate_and_flush(gradient_based):
    from appfl_amd import ops
    n, K = 50001, 3
    base, xs = make_clients(n, K)
    g_ref, step_ref = base.copy(), np.full_like(base, 123.0)
    g, step = dev(base), torch.full((n,), 123.0, device="cuda")
    alphas = [0.3, 0.9 * 3 ** -0.5 / 4, 0.225]
    for k in range(K):
        last = k == K - 1
        O.buff_accum(xs[k], alphas[k], gradient_based, g_ref, step_ref, k == 0, last)
        ops.buff_accum(dev(xs[k]), alphas[k], g, step, k == 0, gradient_based, flush=last)
        assert_bits(step, step_ref, f"step {k}")
        assert_bits(g, g_ref, f"g {k}")
    # unfused flush gives the same bits
    g2_ref, s2_ref = base.copy(), np.zeros_like(base)
    g2, s2 = dev(base), torch.zeros(n, device="cuda")
    for k in range(K):
        O.buff_accum(xs[k], alphas[k], gradient_based, g2_ref, s2_ref, k == 0, False)
        ops.buff_accum(dev(xs[k]), alphas[k], g2, s2, k == 0, gradient_based)
    O.apply_step(g2_ref, s2_ref)
    ops.apply_step(g2, s2)
    assert_bits(g2, g_ref, "unfused flush")
    assert_bits(g2, g2_ref, "unfused flush vs oracle")


@pytest.mark.parametrize("dtype", [np.float32, np.int64, np.int32])
def test_buff_raw_entries(dtype):
    from appfl_amd import ops
    n, K = 1001, 3
    rng = np.random.default_rng(3)
    xs = [(rng.integers(-50, 50, n)).astype(dtype) for _ in range(K)]
    xs[0][:2], xs[1][:2], xs[2][:2] = [4, -4], [7, -7], [0, 0]
    tdt = getattr(torch, np.dtype(dtype).name)
    step_ref = np.full(n, 9, dtype)
    step = torch.full((n,), 9, dtype=tdt, device="cuda")
    for k in range(K):
        O.buff_accum_raw(xs[k], step_ref, k == 0)
        ops.buff_accum_raw(dev(xs[k]), step, k == 0)
    assert_bits(step, step_ref, "raw step")
    g_ref = np.zeros(n, dtype)
    g = torch.zeros(n, dtype=tdt, device="cuda")
    O.buff_flush_mean(g_ref, step_ref, 2)
    ops.buff_flush_mean(g, step, 2)
    assert_bits(g, g_ref, "flush mean")


def test_synth_fill_matches_host_generator():
    from appfl_amd import ops
    n = 100001
    base = O.synth_fill(n, 7, first=5, scale=0.05)
    want = O.synth_fill(n, 1234, first=99, scale=0.01, base=base)
    b = torch.empty(n, device="cuda")
    ops.synth_fill(b, 7, first=5, scale=0.05)
    assert_bits(b, base, "base")
    p = torch.empty(n, device="cuda")
    ops.synth_fill(p, 1234, first=99, scale=0.01, base=b)
    assert_bits(p, want, "update")
    assert abs(float(np.std(O.synth_fill(n, 1))) - 1.0) < 0.02


def test_every_launch_shape_gives_the_same_bits():
    from appfl_amd import ops
    n, K = 300007, 19
    base, xs = make_clients(n, K)
    w = [1.0 / K] * K
    g_ref, m_ref, v_ref = base.copy(), np.zeros_like(base), np.full_like(base, np.float32(1e-6))
    O.reduce_opt(xs, w, g_ref, m_ref, v_ref, O.OPT_ADAM, **HP)
    dx = [dev(x) for x in xs]
    try:
        for vpt in (1, 2, 4):
            for unroll in (4, 8, 16):
                for nt in (0, 1):
                    ops.tune(vpt, unroll, nt, 64, 0)
                    g, m, v = dev(base), dev(np.zeros_like(base)), dev(np.full_like(base, np.float32(1e-6)))
                    ops.reduce_opt(dx, w, g, m, v, "adam", **HP)
                    assert_bits(g, g_ref, f"g vpt={vpt} u={unroll} nt={nt}")
                    assert_bits(v, v_ref, f"v vpt={vpt} u={unroll} nt={nt}")
    finally:
        ops.tune()


@pytest.mark.parametrize("opt,hp", [("none", {}), ("avgm", HP), ("yogi", HP_YOGI)])
@pytest.mark.parametrize("K", [3, 10, 15, 17, 23, 31, 47])
def test_client_counts_that_are_no_multiple_of_the_batch(opt, hp, K):
    """The client loop folds whole batches of `unroll` clients and then the rest in batches of
    8 / 4 / 2 / 1: every remainder path, under the widest and the narrowest launch shape, gives
    the oracle's bits."""
    from appfl_amd import ops
    n = 70001
    code = {"none": O.OPT_NONE, "avgm": O.OPT_AVGM, "yogi": O.OPT_YOGI}[opt]
    base, xs = make_clients(n, K)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v_ref = np.full_like(base, np.float32(0.0 if opt in ("none", "avgm") else hp["tau"] ** 2))
    O.reduce_opt(xs, w, g_ref, m_ref, v_ref, code, **hp)
    dx = [dev(x) for x in xs]
    try:
        for vpt, unroll in ((4, 16), (1, 4), (2, 8)):
            ops.tune(vpt, unroll, 1, 0, 0)
            g, m, v = dev(base), dev(np.zeros_like(base)), dev(np.full_like(base, v_ref.flat[0] * 0 + (
                np.float32(0.0 if opt in ("none", "avgm") else hp["tau"] ** 2))))
            ops.reduce_opt(dx, w, g, m if opt == "yogi" else None, v if opt != "none" else None, opt, **hp)
            assert_bits(g, g_ref, f"g vpt={vpt} unroll={unroll}")
            if opt != "none":
                assert_bits(v, v_ref, f"v vpt={vpt} unroll={unroll}")
    finally:
        ops.tune()


@pytest.mark.parametrize("opt,hp", [("none", {}), ("avgm", HP), ("adam", HP), ("yogi", HP_YOGI), ("adagrad", HP)])
@pytest.mark.parametrize("n", [7, 1027, 300007, 4500003])
@pytest.mark.parametrize("nt", [1, 5], ids=["round_robin", "contiguous"])
def test_persistent_stash_kernel_matches_oracle(opt, hp, n, nt):
    """The persistent LDS-stash kernel forced on (also for vectors far smaller than its grid, so
    most blocks own no or a partial tile) and at its natural size, with both ways of dealing
    tiles to blocks: same bits as the oracle for the final state, the optional step output and
    the step-only mode."""
    from appfl_amd import ops
    K = 49 if n > 1000000 else 11
    if n > 1000000:
        assert ops.reduce_kernel_name(K, n, "none") == "reduce_park_kernel"
        assert ops.reduce_kernel_name(33, 12_000_000, "none") == "reduce_kernel"
        assert ops.reduce_kernel_name(K, n, "avgm") == "reduce_late_kernel"
        assert ops.reduce_kernel_name(K, 1_500_000, "avgm") == "reduce_kernel"     # under two tiles per CU
        assert ops.reduce_kernel_name(K, n, "adam") == "reduce_late_kernel"
    code = {"none": O.OPT_NONE, "avgm": O.OPT_AVGM, "adam": O.OPT_ADAM, "yogi": O.OPT_YOGI,
            "adagrad": O.OPT_ADAGRAD}[opt]
    base, xs = make_clients(n, K)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v_ref = np.full_like(base, np.float32(0.0 if opt in ("none", "avgm") else hp["tau"] ** 2))
    step_ref = np.empty_like(base)
    pre_ref = np.empty_like(base)
    O.reduce_step(xs, w, base, pre_ref)
    O.reduce_opt(xs, w, g_ref, m_ref, v_ref, code, step_out=step_ref, **hp)
    dx = [dev(x) for x in xs]
    try:
        ops.tune(nt=nt, stash=-1)
        for with_step in (False, True):
            g, m, v = dev(base), dev(np.zeros_like(base)), dev(np.full_like(base, v_ref.flat[0] * 0 + (
                np.float32(0.0 if opt in ("none", "avgm") else hp["tau"] ** 2))))
            step = torch.full((n,), 7.0, device="cuda") if with_step else None
            ops.reduce_opt(dx, w, g, m if opt not in ("none", "avgm") else None,
                           v if opt != "none" else None, opt, step_out=step, **hp)
            assert_bits(g, g_ref, f"g step_out={with_step}")
            if opt != "none":
                assert_bits(v, v_ref, "v")
            if opt not in ("none", "avgm"):
                assert_bits(m, m_ref, "m")
            if with_step:
                assert_bits(step, step_ref, "step")
        g = dev(base)
        pre = torch.empty_like(g)
        ops.reduce_step(dx, w, g, pre)
        assert_bits(pre, pre_ref, "step-only")
        assert_bits(g, base, "g untouched")
    finally:
        ops.tune()


@pytest.mark.parametrize("opt,hp", [("none", {}), ("adam", HP)])
def test_phased_persistent_kernel_beside_other_work_on_the_chip(opt, hp):
    """The persistent kernel's meetings are bounded waits that carry no data: with another
    stream's kernels holding CUs (so that some of its blocks start late or not together) a
    phased launch still ends and gives the oracle's bits; the meeting counters are back at zero
    for the launches that follow."""
    from appfl_amd import ops
    # tiles per CU beyond what the kernel parks (18 under plain FedAvg, 3 under FedAdam): phased
    n, K = (20_000_004 if opt == "none" else 6_000_004), 49
    assert ops.reduce_kernel_name(K, n, "none") == "reduce_park_kernel"     # the default; forced below
    code = O.OPT_NONE if opt == "none" else O.OPT_ADAM
    base, xs = make_clients(n, K)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v_ref = np.full_like(base, np.float32(hp["tau"] ** 2 if hp else 0.0))
    dx = [dev(x) for x in xs]
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.empty_like(a)
    g, m, v = dev(base), dev(m_ref.copy()), dev(v_ref.copy())
    try:
        ops.tune(stash=-1)
        for rnd in range(3):
            O.reduce_opt(xs, w, g_ref, m_ref, v_ref, code, **hp)
            with torch.cuda.stream(side):
                for _ in range(30):
                    torch.mm(a, a, out=b)
            ops.reduce_opt(dx, w, g, m if hp else None, v if hp else None, opt, **hp)
            torch.cuda.synchronize()
            assert_bits(g, g_ref, f"g round {rnd}")
            if hp:
                assert_bits(v, v_ref, f"v round {rnd}")
    finally:
        ops.tune()


@pytest.mark.parametrize("n", [7, 1027, 300007, 4500003, 12000004])
@pytest.mark.parametrize("K", [1, 2, 6, 16, 17, 49, 128])
@pytest.mark.parametrize("nt", [1, 65, 1025], ids=["default", "one_group", "phases_of_4"])
def test_wide_persistent_kernel_matches_oracle(n, K, nt):
    """The wide-batch persistent kernel (plain FedAvg on very long vectors) forced on at every
    length -- no tile, partial tiles, more tiles per CU than one phase parks -- and with other
    phase groupings and lengths: the oracle's bits."""
    from appfl_amd import ops
    if n > 10_000_000 and K > 17:
        pytest.skip("host memory")
    assert ops.reduce_kernel_name(32, 162_500_000, "none") == "reduce_park_kernel"
    assert ops.reduce_kernel_name(32, 162_500_000, "avgm") == "reduce_late_kernel"
    assert ops.reduce_kernel_name(32, 162_500_000, "adam") == "reduce_late_kernel"
    assert ops.reduce_kernel_name(6, 162_500_000, "none") == "reduce_park_kernel"
    assert ops.reduce_kernel_name(5, 162_500_000, "none") == "reduce_kernel"
    assert ops.reduce_kernel_name(32, 65_000_000, "none") == "reduce_kernel"
    assert ops.reduce_kernel_name(40, 11_689_512, "none") == "reduce_park_kernel"
    assert ops.reduce_kernel_name(24, 4_000_000, "none") == "reduce_park_kernel"
    assert ops.reduce_kernel_name(24, 11_689_512, "none") == "reduce_kernel"
    assert ops.reduce_kernel_name(100, 200_000, "none") == "reduce_kernel"
    base, xs = make_clients(n, K)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref = base.copy()
    O.reduce_opt(xs, w, g_ref, np.zeros_like(base), np.zeros_like(base), O.OPT_NONE)
    dx = [dev(x) for x in xs]
    try:
        ops.tune(nt=nt, stash=-2)
        assert ops.reduce_kernel_name(K, n, "none") == "reduce_park_kernel"
        g = dev(base)
        for _ in range(2):
            g.copy_(dev(base))
            ops.reduce_opt(dx, w, g, None, None, "none")
            assert_bits(g, g_ref, "g")
        # the step-only form parks and writes the step instead
        pre_ref = np.empty_like(base)
        O.reduce_step(xs, w, base, pre_ref)
        g.copy_(dev(base))
        pre = torch.full((n,), 9.0, device="cuda")
        ops.reduce_step(dx, w, g, pre)
        assert_bits(pre, pre_ref, "step-only")
        assert_bits(g, base, "g untouched")
    finally:
        ops.tune()


def test_wide_persistent_kernel_at_its_default_phases_on_a_long_client_list():
    """Default routing and phase length at 49 clients on a vector of 12 tiles per CU (the
    wide-batch persistent kernel in short chip-wide phases, the last one partial): the oracle's
    bits, two rounds in a row."""
    from appfl_amd import ops
    n, K = 12_000_004, 49
    assert ops.reduce_kernel_name(K, n, "none") == "reduce_park_kernel"
    base, xs = make_clients(n, K, spike=False)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref = base.copy()
    g = dev(base)
    dx = [dev(x) for x in xs]
    for rnd in range(2):
        O.reduce_opt(xs, w, g_ref, np.zeros_like(base), np.zeros_like(base), O.OPT_NONE)
        ops.reduce_opt(dx, w, g, None, None, "none")
        assert_bits(g, g_ref, f"g round {rnd}")


def test_wide_persistent_kernel_carries_the_step_of_the_first_client_chunks():
    """K > FA_KCHUNK under FedAdam with the wide-batch kernel forced on: the first chunk's
    launch (plain FedAvg arithmetic, step only) parks and writes the partial step."""
    from appfl_amd import ops
    n, K = 300007, ops.kchunk() + 5
    base, xs = make_clients(n, K, spike=False)
    w = [1.0 / K] * K
    g_ref, m_ref, v_ref = base.copy(), np.zeros_like(base), np.full_like(base, np.float32(1e-6))
    O.reduce_opt(xs, w, g_ref, m_ref, v_ref, O.OPT_ADAM, **HP)
    g, m, v = dev(base), dev(np.zeros_like(base)), dev(np.full_like(base, np.float32(1e-6)))
    try:
        ops.tune(stash=-2)
        ops.reduce_opt([dev(x) for x in xs], w, g, m, v, "adam", step_ws=torch.empty_like(g), **HP)
    finally:
        ops.tune()
    assert_bits(g, g_ref, "g")
    assert_bits(m, m_ref, "m")
    assert_bits(v, v_ref, "v")


def test_wide_persistent_kernel_where_the_library_routes_to_it():
    """A vector long enough for the default routing to pick the wide-batch persistent kernel
    (several phases per group, a scalar tail after the 128-bit body): the oracle's bits."""
    from appfl_amd import ops
    n, K = 76_000_003, 6
    assert ops.reduce_kernel_name(K, n, "none") == "reduce_park_kernel"
    base, xs = make_clients(n, K, spike=False)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref = base.copy()
    O.reduce_opt(xs, w, g_ref, None, None, O.OPT_NONE)
    g = dev(base)
    ops.reduce_opt([dev(x) for x in xs], w, g, None, None, "none")
    assert_bits(g, g_ref, "g")


def test_wide_persistent_kernel_after_client_chunks_and_beside_other_work():
    """K > FA_KCHUNK: the last chunk's launch starts from the carried step.  And with another
    stream's kernels holding CUs the phased launch still ends with the oracle's bits."""
    from appfl_amd import ops
    n, K = 12_000_004, ops.kchunk() + 16
    base, xs = make_clients(n, 24, spike=False)
    xs = [xs[k % 24] for k in range(K)]
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref = base.copy()
    dx24 = [dev(x) for x in xs[:24]]
    dx = [dx24[k % 24] for k in range(K)]
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.empty_like(a)
    g = dev(base)
    ws = torch.empty_like(g)
    try:
        ops.tune(stash=-2)
        for rnd in range(2):
            O.reduce_opt(xs, w, g_ref, np.zeros_like(base), np.zeros_like(base), O.OPT_NONE)
            with torch.cuda.stream(side):
                for _ in range(30):
                    torch.mm(a, a, out=b)
            ops.reduce_opt(dx, w, g, None, None, "none", step_ws=ws)
            torch.cuda.synchronize()
            assert_bits(g, g_ref, f"g round {rnd}")
    finally:
        ops.tune()


def test_persistent_stash_kernel_with_client_chunks():
    """K > FA_KCHUNK through the persistent kernel: the carry pass parks the partial step."""
    from appfl_amd import ops
    n, K = 70003, ops.kchunk() + 5
    base, xs = make_clients(n, K, spike=False)
    w = [1.0 / K] * K
    g_ref, m_ref, v_ref = base.copy(), np.zeros_like(base), np.full_like(base, np.float32(1e-6))
    O.reduce_opt(xs, w, g_ref, m_ref, v_ref, O.OPT_ADAM, **HP)
    g, m, v = dev(base), dev(np.zeros_like(base)), dev(np.full_like(base, np.float32(1e-6)))
    try:
        ops.tune(stash=-1)
        ops.reduce_opt([dev(x) for x in xs], w, g, m, v, "adam", step_ws=torch.empty_like(g), **HP)
    finally:
        ops.tune()
    assert_bits(g, g_ref, "g")
    assert_bits(m, m_ref, "m")
    assert_bits(v, v_ref, "v")


@pytest.mark.parametrize("mode,code", [("replace", 0), ("convex", 1), ("subtract", 2), ("convex_one", 3)])
@pytest.mark.parametrize("n,K", [(5, 1), (4099, 3), (300007, 9), (2051, 131), (70001, 23), (70001, 31), (4099, 16)])
def test_weighted_mix(mode, code, n, K):
    from appfl_amd import ops
    if mode == "convex_one":
        K = 1
    base, xs = make_clients(n, K)
    w = [0.9 * (k + 1) ** -0.5 / 7 for k in range(K)]
    coeff = 1 - sum(w)
    g_ref = base.copy()
    O.weighted_mix(xs, w, g_ref, code, coeff)
    g = dev(base)
    ws = torch.empty_like(g) if K > ops.kchunk() else None
    ops.weighted_mix([dev(x) for x in xs], w, g, mode, coeff, acc_ws=ws)
    assert_bits(g, g_ref, "g")


@pytest.mark.parametrize("mode", ["replace", "convex", "subtract", "convex_one", "sum_add", "sum_set"])
@pytest.mark.parametrize("n,K", [(5, 1), (4099, 3), (2051, 131)])
def test_weighted_mix_float64(mode, n, K):
    """The f64 mixes against the same op sequence in numpy doubles (each numpy op is one
    correctly rounded IEEE operation); values and weights are not representable in fp32."""
    from appfl_amd import ops
    if mode == "convex_one":
        K = 1
    rng = np.random.default_rng(n + K)
    base = rng.standard_normal(n) * 0.05
    xs = [base + rng.standard_normal(n) * 0.01 for _ in range(K)]
    if n > 16:
        xs[0][3] = 0.0
        xs[K - 1][5] = 1e300
        xs[0][7] = -0.0
        xs[0][11] = 1e-310          # denormal
    w = [0.9 * (k + 1) ** -0.5 / 7 for k in range(K)]
    coeff = 1 - sum(w)
    want = base.copy()
    if mode == "subtract":
        for x, wk in zip(xs, w):
            want = want - x * np.float64(wk)
    else:
        acc = xs[0] * np.float64(w[0]) if mode in ("convex_one", "sum_add", "sum_set") \
            else np.zeros(n) + xs[0] * np.float64(w[0])
        for x, wk in zip(xs[1:], w[1:]):
            acc = acc + x * np.float64(wk)
        if mode in ("convex", "convex_one"):
            want = base * np.float64(coeff) + acc
        elif mode == "sum_add":
            want = base + acc
        else:
            want = acc
    g = dev(base)
    ws = torch.empty_like(g) if K > ops.kchunk() else None
    ops.weighted_mix([dev(x) for x in xs], w, g, mode, coeff, acc_ws=ws)
    got = g.cpu().numpy()
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, f"{bad.size} of {n} differ, first at {bad[:5]}: {got[bad[:5]]} vs {want[bad[:5]]}"


def _round_rows(n_step, counts, K, seed):
    """K client rows [step f32 | mean f32 | pad16 | i64 | i32] as numpy pieces and as one device
    slab; returns (base, per-client pieces, slab, row stride in bytes, byte offsets)."""
    n_f32, n_i64, n_i32 = counts
    rng = np.random.default_rng(seed)
    base, xs = make_clients(n_step, K, seed=seed, spike=n_step > 16)
    off_f32 = 4 * n_step
    off_i64 = (off_f32 + 4 * n_f32 + 15) // 16 * 16
    off_i32 = off_i64 + 8 * n_i64
    stride = (off_i32 + 4 * n_i32 + 255) // 256 * 256
    pieces = []
    slab = np.zeros((K, stride), np.uint8)
    for k in range(K):
        f = O.synth_fill(max(n_f32, 1), 50 + k, scale=1.0)[:n_f32]
        a = rng.integers(-2 ** 40, 2 ** 40, n_i64).astype(np.int64)
        b = rng.integers(-10 ** 6, 10 ** 6, n_i32).astype(np.int32)
        if n_i64 >= 3:
            a[:3] = [6, 7, -7] if k == 0 else 0
        pieces.append((xs[k], f, a, b))
        slab[k, :off_f32] = xs[k].view(np.uint8)
        slab[k, off_f32:off_f32 + 4 * n_f32] = f.view(np.uint8)
        slab[k, off_i64:off_i64 + 8 * n_i64] = a.view(np.uint8)
        slab[k, off_i32:off_i32 + 4 * n_i32] = b.view(np.uint8)
    return base, pieces, torch.from_numpy(slab).cuda(), stride, (off_f32, off_i64, off_i32)


@pytest.mark.parametrize("opt,hp", [("none", {}), ("adam", HP)])
@pytest.mark.parametrize("n_step,K,counts", [
    (4099, 3, (5, 3, 2)),            # one wave per element
    (4099, 100, (0, 20, 0)),         # the usual case: a few counters, many clients
    (1027, 128, (7, 0, 1)),
    (300007, 10, (300, 20, 9)),      # one thread per element, on leading blocks of the grid
    (300007, 70, (3000, 0, 0)),
    (2500003, 40, (0, 20, 0)),       # persistent kernel, by wave
    (2500003, 100, (9000, 53, 5)),   # persistent kernel, by thread
    (2500003, 33, (20000, 3, 0)),    # too many elements to fold: separate mean launches
    (4099, 130, (5, 3, 2)),          # more clients than one launch folds: separate launches
    (3, 5, (2, 1, 1)),               # no 128-bit body: the scalar kernel carries the tail
    (0, 4, (5, 3, 2)),               # no weighted entries at all
])
def test_round_folds_plain_means_into_the_reduce_launch(opt, hp, n_step, K, counts):
    if n_step == 0 and opt != "none":
        pytest.skip("nothing to optimise")
    _check_round(opt, hp, n_step, K, counts)


@pytest.mark.parametrize("n_step,K,counts", [
    (300007, 17, (300, 20, 9)), (2500003, 40, (0, 20, 0)), (2500003, 100, (9000, 53, 5)),
    (11000003, 16, (0, 20, 0)),      # more tiles per CU than one phase parks
])
def test_round_on_the_wide_persistent_kernel(n_step, K, counts):
    """fa_round with the wide-batch persistent kernel forced on: the plain means ride in its
    launch as they do in the other kernels'."""
    from appfl_amd import ops
    try:
        ops.tune(stash=-2)
        assert ops.reduce_kernel_name(K, n_step, "none") == "reduce_park_kernel"
        _check_round("none", {}, n_step, K, counts)
    finally:
        ops.tune()


def _check_round(opt, hp, n_step, K, counts):
    from appfl_amd import ops
    code = {"none": O.OPT_NONE, "adam": O.OPT_ADAM}[opt]
    base, pieces, slab, stride, offs = _round_rows(n_step, counts, K, seed=11)
    w = [(3 + 2 * k) / sum(3 + 2 * j for j in range(K)) for k in range(K)]
    g_ref = base.copy()
    m_ref = np.zeros_like(base)
    v_ref = np.full_like(base, np.float32(HP["tau"] ** 2))
    if n_step:
        O.reduce_opt([p[0] for p in pieces], w, g_ref, m_ref, v_ref, code, **hp)
    wants = []
    for j, dt in enumerate((np.float32, np.int64, np.int32)):
        want = np.zeros(counts[j], dt)
        if counts[j]:
            O.mean([p[1 + j] for p in pieces], want)
        wants.append(want)
    g = dev(base) if n_step else torch.zeros(4, device="cuda")
    m = torch.zeros_like(g)
    v = torch.full_like(g, HP["tau"] ** 2)
    outs = [torch.full((max(c, 1),), -77, dtype=t, device="cuda")
            for c, t in zip(counts, (torch.float32, torch.int64, torch.int32))]
    rows = [slab.data_ptr() + k * stride for k in range(K)]
    ws = torch.empty(max(n_step, 1), device="cuda") if K > ops.kchunk() else None
    means = [(offs[j], outs[j], counts[j]) for j in range(3) if counts[j]]
    for _ in range(2):       # twice: the second round starts from the first one's state (same client rows)
        ops.round_sync(rows, w, g, m if opt != "none" else None, v if opt != "none" else None,
                       opt, step_ws=ws, n=n_step, means=means, **hp)
        if _ == 0 and n_step:
            O.reduce_opt([p[0] for p in pieces], w, g_ref, m_ref, v_ref, code, **hp)
    torch.cuda.synchronize()
    if n_step:
        assert_bits(g[:n_step], g_ref, "g")
        if opt != "none":
            assert_bits(m, m_ref, "m")
            assert_bits(v, v_ref, "v")
    for j, name in enumerate(("mean f32", "mean i64", "mean i32")):
        if counts[j]:
            assert_bits(outs[j][:counts[j]], wants[j], name)


@pytest.mark.parametrize("opt,hp", [("none", {}), ("yogi", HP_YOGI)])
@pytest.mark.parametrize("n_step,K,k_ready", [
    (4099, 3, 2), (4099, 100, 99), (4099, 100, 50), (1027, 128, 1), (300007, 70, 69),
    (2500003, 40, 39),               # the first launch takes the persistent kernel
    (4099, 200, 199), (4099, 200, 60), (4099, 200, 128),   # beside the cuts every FA_KCHUNK clients
])
def test_round_split_where_the_uploads_have_landed(opt, hp, n_step, K, k_ready):
    """fa_round with k_ready: the landed clients in one launch, the rest after the event; the
    partial step travels through the workspace, so the bits are those of one launch."""
    from appfl_amd import ops
    code = {"none": O.OPT_NONE, "yogi": O.OPT_YOGI}[opt]
    counts = (5, 3, 2)
    base, pieces, slab, stride, offs = _round_rows(n_step, counts, K, seed=23)
    w = [(3 + 2 * k) / sum(3 + 2 * j for j in range(K)) for k in range(K)]
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v_ref = np.full_like(base, np.float32(HP["tau"] ** 2))
    O.reduce_opt([p[0] for p in pieces], w, g_ref, m_ref, v_ref, code, **hp)
    wants = []
    for j, dt in enumerate((np.float32, np.int64, np.int32)):
        want = np.zeros(counts[j], dt)
        O.mean([p[1 + j] for p in pieces], want)
        wants.append(want)
    g = dev(base)
    m, v = torch.zeros_like(g), torch.full_like(g, HP["tau"] ** 2)
    outs = [torch.full((c,), -77, dtype=t, device="cuda")
            for c, t in zip(counts, (torch.float32, torch.int64, torch.int32))]
    rows = [slab.data_ptr() + k * stride for k in range(K)]
    means = [(offs[j], outs[j], counts[j]) for j in range(3)]
    ws = torch.empty(n_step, device="cuda")
    with pytest.raises(Exception):       # a split needs the workspace
        ops.round_sync(rows[:100], w[:100], g, n=n_step, k_ready=min(k_ready, 99))
    # the remaining rows "arrive" on a side stream: the event is recorded behind a copy that
    # rewrites them in place
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tail = slab[k_ready:].clone()
        slab[k_ready:].copy_(tail)
        landed = torch.cuda.Event()
        landed.record(side)
    ops.round_sync(rows, w, g, m if opt != "none" else None, v if opt != "none" else None, opt,
                   step_ws=ws, n=n_step, means=means, k_ready=k_ready, rest_event=landed, **hp)
    torch.cuda.synchronize()
    assert_bits(g, g_ref, "g")
    if opt != "none":
        assert_bits(m, m_ref, "m")
        assert_bits(v, v_ref, "v")
    for j, name in enumerate(("mean f32", "mean i64", "mean i32")):
        assert_bits(outs[j], wants[j], name)


@pytest.mark.parametrize("opt,hp", [("none", {}), ("avgm", HP), ("adam", HP)])
@pytest.mark.parametrize("n_step,K,cuts", [
    (4099, 3, (1, 2)), (4099, 100, (1, 2, 3, 50, 99)), (1027, 128, (127,)), (300007, 70, (10, 69)),
    (2500003, 40, (1, 39)),
    (4099, 200, (1, 128, 199)), (4099, 200, (60, 130)),   # beside the cuts every FA_KCHUNK clients
    (3, 5, (4,)),                                         # scalar kernel
])
def test_round_resumed_from_a_partial_step(opt, hp, n_step, K, cuts):
    """fa_reduce_step_resume + fa_round_resume: the clients folded a few at a time (as they
    arrive), the round finished from the partial step -- the bits of one launch over all K."""
    from appfl_amd import ops
    code = {"none": O.OPT_NONE, "avgm": O.OPT_AVGM, "adam": O.OPT_ADAM}[opt]
    counts = (5, 3, 2)
    base, pieces, slab, stride, offs = _round_rows(n_step, counts, K, seed=31)
    w = [(3 + 2 * k) / sum(3 + 2 * j for j in range(K)) for k in range(K)]
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v0 = np.float32(HP["tau"] ** 2 if opt == "adam" else 0.0)
    v_ref = np.full_like(base, v0)
    O.reduce_opt([p[0] for p in pieces], w, g_ref, m_ref, v_ref, code, **hp)
    wants = []
    for j, dt in enumerate((np.float32, np.int64, np.int32)):
        want = np.zeros(counts[j], dt)
        O.mean([p[1 + j] for p in pieces], want)
        wants.append(want)
    g = dev(base)
    m, v = torch.zeros_like(g), torch.full_like(g, float(v0))
    outs = [torch.full((c,), -77, dtype=t, device="cuda")
            for c, t in zip(counts, (torch.float32, torch.int64, torch.int32))]
    rows = [slab.data_ptr() + k * stride for k in range(K)]
    means = [(offs[j], outs[j], counts[j]) for j in range(3)]
    step = torch.full((n_step,), 123.0, device="cuda")     # stale contents: the first fold overwrites
    with pytest.raises(Exception):       # resuming needs the partial step
        ops.round_sync(rows, w, g, n=n_step, k_done=cuts[0])
    with pytest.raises(Exception):       # at least one client is left to fold
        ops.round_sync(rows, w, g, step_ws=step, n=n_step, k_done=K)
    done = 0
    for cut in cuts:
        # later rows may not have landed yet: only rows [done, cut) are read
        ops.reduce_step(rows[:cut], w[:cut], g, step, n=n_step, k_done=done) if done else \
            ops.reduce_step(rows[:cut], w[:cut], g, step, n=n_step)
        done = cut
    ops.round_sync(rows, w, g, m if opt == "adam" else None, v if opt != "none" else None, opt,
                   step_ws=step, n=n_step, means=means, k_done=done, **hp)
    torch.cuda.synchronize()
    assert_bits(g, g_ref, "g")
    if opt != "none":
        assert_bits(v, v_ref, "v")
    if opt == "adam":
        assert_bits(m, m_ref, "m")
    for j, name in enumerate(("mean f32", "mean i64", "mean i32")):
        assert_bits(outs[j], wants[j], name)


def test_meeting_counters_heal_and_later_rounds_keep_their_speed():
    """The persistent kernel's meeting counters are timing-only state that nothing clears between
    launches.  Garbage left in them (as by an aborted launch), more than 64 phased rounds, and
    rounds overlapping on two streams must neither change a bit nor slow down the rounds that
    follow (an unhealed counter costs every block a time-out at every meeting)."""
    from appfl_amd import ops
    K, n = 48, 30_000_000                       # 28 tiles per CU: written in phases, with meetings
    assert ops.reduce_kernel_name(K, n, "none") == "reduce_park_kernel"
    d = torch.device("cuda")
    g0 = torch.empty(n, device=d)
    ops.synth_fill(g0, 7, scale=0.05)
    xs = torch.empty((K, n), device=d)
    for k in range(K):
        ops.synth_fill(xs[k], 1000 + k, scale=0.01, base=g0)
    rows, w = list(xs), [1.0 / K] * K

    def timed(g, reps=7):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.reduce_opt(rows, w, g)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    ref = g0.clone()
    ops.reduce_opt(rows, w, ref)                # one round from g0: the bits every variant must give
    base_ms = timed(g0.clone())
    ops.scribble_meeting_counters()
    g = g0.clone()
    ops.reduce_opt(rows, w, g)
    assert torch.equal(g, ref)
    healed_ms = timed(g0.clone())
    assert healed_ms <= 1.15 * base_ms, (base_ms, healed_ms)
    # > 64 phased rounds, overlapping on two streams (each stream has a counter slot of its own)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ga, gb = g0.clone(), g0.clone()
    torch.cuda.synchronize()
    for i in range(80):
        with torch.cuda.stream(s1):
            ops.reduce_opt(rows, w, ga)
        with torch.cuda.stream(s2):
            ops.reduce_opt(rows, w, gb)
    torch.cuda.synchronize()
    assert torch.equal(ga, gb)
    after_ms = timed(g0.clone())
    assert after_ms <= 1.15 * base_ms, (base_ms, after_ms)
    # garbage again, this time found by launches on the side streams
    ops.scribble_meeting_counters()
    with torch.cuda.stream(s1):
        g1 = g0.clone()
        ops.reduce_opt(rows, w, g1)
    torch.cuda.synchronize()
    assert torch.equal(g1, ref)


_CODES = {"avgm": O.OPT_AVGM, "adam": O.OPT_ADAM, "yogi": O.OPT_YOGI, "adagrad": O.OPT_ADAGRAD}


@pytest.mark.parametrize("opt,hp", [("avgm", HP), ("adam", HP), ("yogi", HP_YOGI), ("adagrad", HP)])
@pytest.mark.parametrize("n", [7, 1027, 300007, 4500003, 12000004])
@pytest.mark.parametrize("K", [1, 6, 17, 49])
@pytest.mark.parametrize("nt", [1, 65, 1025, 2561], ids=["default", "one_group", "phases_of_4", "phases_of_10"])
def test_late_epilogue_persistent_kernel_matches_oracle(opt, hp, n, K, nt):
    """The persistent kernel that parks the averaged step and runs the optimizer epilogue at the
    end of a phase, forced on at every length -- no tile, partial tiles, more tiles per CU than
    a phase parks -- and with other phase groupings and lengths: the oracle's bits for g, m and v,
    two rounds in a row (the moments feed back)."""
    from appfl_amd import ops
    if n > 10_000_000 and K > 17:
        pytest.skip("host memory")
    base, xs = make_clients(n, K)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref, m_ref = base.copy(), np.zeros_like(base)
    v_ref = np.full_like(base, np.float32(0.0 if opt == "avgm" else hp["tau"] ** 2))
    g, m, v = dev(base), dev(m_ref.copy()), dev(v_ref.copy())
    dx = [dev(x) for x in xs]
    try:
        ops.tune(nt=nt, stash=-3)
        assert ops.reduce_kernel_name(K, n, opt) == "reduce_late_kernel"
        for rnd in range(2):
            O.reduce_opt(xs, w, g_ref, m_ref, v_ref, _CODES[opt], **hp)
            ops.reduce_opt(dx, w, g, m if opt != "avgm" else None, v, opt, **hp)
            assert_bits(g, g_ref, f"g round {rnd}")
            assert_bits(v, v_ref, f"v round {rnd}")
            if opt != "avgm":
                assert_bits(m, m_ref, f"m round {rnd}")
    finally:
        ops.tune()


def test_late_epilogue_kernel_where_the_library_routes_to_it():
    """Optimizer rounds take the late-epilogue kernel by default from 32 clients on vectors of
    >= 4 tiles per CU, from fewer clients on longer vectors, and FedAvgM from 48 clients (several
    phases per group, a scalar tail after the 128-bit body); shorter client lists, shorter
    vectors and plain FedAvg do not.  The oracle's bits, moments included, three rounds."""
    from appfl_amd import ops
    name = ops.reduce_kernel_name
    assert name(32, 25_557_032, "adam") == name(24, 18_000_000, "yogi") == "reduce_late_kernel"
    assert name(20, 35_000_000, "adagrad") == name(16, 60_000_000, "adam") == "reduce_late_kernel"
    assert name(64, 25_557_032, "adam") == name(32, 11_689_512, "adam") == "reduce_late_kernel"
    assert name(64, 25_557_032, "avgm") == name(32, 47_000_000, "avgm") == "reduce_late_kernel"
    assert name(16, 25_557_032, "adam") == name(20, 11_689_512, "adam") == "reduce_kernel"
    assert name(32, 1_500_000, "adam") == name(32, 25_557_032, "avgm") == "reduce_kernel"
    assert name(32, 25_557_032, "none") == "reduce_kernel"
    assert name(64, 25_557_032, "none") == "reduce_park_kernel"
    assert name(64, 1_500_000, "avgm") == "reduce_kernel"
    assert name(64, 3_000_000, "avgm") == name(40, 11_689_512, "avgm") == name(16, 35_000_000, "adam") == "reduce_late_kernel"
    n, K = 17_000_003, 24
    base, xs = make_clients(n, K, spike=False)
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref, m_ref, v_ref = base.copy(), np.zeros_like(base), np.full_like(base, np.float32(1e-6))
    g, m, v = dev(base), dev(m_ref.copy()), dev(v_ref.copy())
    dx = [dev(x) for x in xs]
    assert name(K, n, "yogi") == "reduce_late_kernel"
    for rnd in range(3):
        O.reduce_opt(xs, w, g_ref, m_ref, v_ref, O.OPT_YOGI, **HP_YOGI)
        ops.reduce_opt(dx, w, g, m, v, "yogi", **HP_YOGI)
        assert_bits(g, g_ref, f"g round {rnd}")
        assert_bits(m, m_ref, f"m round {rnd}")
        assert_bits(v, v_ref, f"v round {rnd}")


def test_late_epilogue_kernel_after_client_chunks_and_beside_other_work():
    """K > FA_KCHUNK: the last chunk's launch starts from the carried step and runs the late
    epilogue.  With another stream's kernels holding CUs the phased launch still ends with the
    oracle's bits."""
    from appfl_amd import ops
    n, K = 6_000_004, ops.kchunk() + 16
    base, xs24 = make_clients(n, 24, spike=False)
    xs = [xs24[k % 24] for k in range(K)]
    w = [(2 + k) / sum(2 + j for j in range(K)) for k in range(K)]
    g_ref, m_ref, v_ref = base.copy(), np.zeros_like(base), np.full_like(base, np.float32(1e-6))
    dx24 = [dev(x) for x in xs24]
    dx = [dx24[k % 24] for k in range(K)]
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.empty_like(a)
    g, m, v = dev(base), dev(m_ref.copy()), dev(v_ref.copy())
    ws = torch.empty_like(g)
    try:
        ops.tune(stash=-3)
        for rnd in range(2):
            O.reduce_opt(xs, w, g_ref, m_ref, v_ref, O.OPT_ADAM, **HP)
            with torch.cuda.stream(side):
                for _ in range(30):
                    torch.mm(a, a, out=b)
            ops.reduce_opt(dx, w, g, m, v, "adam", step_ws=ws, **HP)
            torch.cuda.synchronize()
            assert_bits(g, g_ref, f"g round {rnd}")
            assert_bits(m, m_ref, f"m round {rnd}")
            assert_bits(v, v_ref, f"v round {rnd}")
    finally:
        ops.tune()


@pytest.mark.parametrize("n_step,K,counts", [
    (300007, 17, (300, 20, 9)), (2500003, 40, (0, 20, 0)), (2500003, 100, (9000, 53, 5)),
    (11000003, 16, (0, 20, 0)),
])
def test_round_on_the_late_epilogue_kernel(n_step, K, counts):
    """fa_round with the late-epilogue kernel forced on: the plain means ride in its launch."""
    from appfl_amd import ops
    try:
        ops.tune(stash=-3)
        assert ops.reduce_kernel_name(K, n_step, "adam") == "reduce_late_kernel"
        _check_round("adam", HP, n_step, K, counts)
    finally:
        ops.tune()


def test_round_plan_launches_equal_marshalled_rounds():
    """fa_round_plan_create / fa_round_plan_launch: the kept arguments give the bits of fa_round
    called afresh, round after round, and the plan outlives the Python lists it was built from."""
    from appfl_amd import ops
    n, K, tail = 70000, 5, 9
    base = O.synth_fill(n + tail, 3, scale=0.05)
    rows = torch.stack([torch.from_numpy(O.synth_fill(n + tail, 40 + k, scale=0.02, base=base)) for k in range(K)]).cuda()
    cnt = torch.arange(K * 4, dtype=torch.int64, device="cuda").reshape(K, 4).contiguous()
    w = [0.1, 0.3, 0.2, 0.25, 0.15]
    hp = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)

    def state():
        g = torch.from_numpy(base.copy()).cuda()
        return g, torch.zeros(n, device="cuda"), torch.full((n,), 1e-6, device="cuda")

    g1, m1, v1 = state()
    g2, m2, v2 = state()
    ptrs = [rows[k].data_ptr() for k in range(K)]
    means1 = [(4 * n, g1[n:], tail)]
    means2 = [(4 * n, g2[n:], tail)]
    plan = ops.RoundPlan(list(ptrs), list(w), g1, m1, v1, "adam", n=n, means=means1, **hp)
    for _ in range(3):
        plan.launch()
        ops.round_sync(ptrs, w, g2, m2, v2, "adam", n=n, means=means2, **hp)
    torch.cuda.synchronize()
    for a, b in ((g1, g2), (m1, m2), (v1, v2)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    del plan


def test_the_entry_points_refuse_bad_arguments_with_their_error_codes():
    """Straight through the C ABI (fp32 entry points): NULL buffers, unknown optimizer / mode /
    flag codes, no clients, misaligned addresses, a client list longer than one launch without
    a workspace, a split without one -- each answered with the documented code before anything
    is launched (the buffers keep their contents)."""
    import ctypes as C

    from appfl_amd import _native as N
    from appfl_amd import ops
    lib = N.load()
    INVALID, ALIGN, WORKSPACE = -1, -2, -3
    n, K = 64, 3
    g = torch.full((n,), 1.5, dtype=torch.float32, device="cuda")
    m, v, step = torch.zeros_like(g), torch.zeros_like(g), torch.zeros_like(g)
    xs = [torch.full((n,), float(k), dtype=torch.float32, device="cuda") for k in range(K)]
    x, w = N.ptr_array([t.data_ptr() for t in xs]), N.double_array([1.0 / K] * K)
    P = lambda t: C.c_void_p(t.data_ptr())                      # noqa: E731
    adam = (0.1, 0.9, 0.99, 1e-3)

    def reduce_opt(x_=x, w_=w, K_=K, g_=P(g), m_=P(m), v_=P(v), opt=2, step_out=None, ws=None):
        return lib.fa_reduce_opt(x_, w_, K_, g_, m_, v_, n, opt, *adam, step_out, ws, None)

    assert reduce_opt() == 0
    torch.cuda.synchronize()
    before = g.clone()
    assert reduce_opt(x_=None) == INVALID and reduce_opt(w_=None) == INVALID and reduce_opt(g_=None) == INVALID
    assert reduce_opt(K_=0) == INVALID and reduce_opt(opt=9) == INVALID and reduce_opt(opt=-1) == INVALID
    assert reduce_opt(v_=None) == INVALID and reduce_opt(m_=None) == INVALID
    assert reduce_opt(g_=C.c_void_p(g.data_ptr() + 2)) == ALIGN
    assert reduce_opt(x_=N.ptr_array([xs[0].data_ptr(), xs[1].data_ptr() + 1, xs[2].data_ptr()])) == ALIGN
    assert reduce_opt(x_=N.ptr_array([xs[0].data_ptr(), 0, xs[2].data_ptr()])) == INVALID
    many = ops.kchunk() + 1
    xm, wm = N.ptr_array([xs[0].data_ptr()] * many), N.double_array([1.0 / many] * many)
    assert reduce_opt(x_=xm, w_=wm, K_=many) == WORKSPACE
    assert lib.fa_reduce_opt_ex(x, w, K, P(g), P(m), P(v), n, 2, *adam, None, None, 8, None) == INVALID   # unknown flag
    assert lib.fa_round(x, w, K, P(g), P(m), P(v), n, 2, *adam, None, None, None, 0, 1, None, None) == WORKSPACE  # a split
    assert lib.fa_round(x, w, K, P(g), P(m), P(v), n, 2, *adam, None, None, None, 0, K + 1, None, None) == INVALID
    assert lib.fa_reduce_step_resume(x, w, K, K, P(g), P(step), n, None) == INVALID                   # no client left
    assert lib.fa_round_resume(x, w, K, 1, P(g), P(m), P(v), n, 2, *adam, None, None, 0, None) == WORKSPACE
    assert lib.fa_server_opt_step(P(step), None, P(v), n, 2, *adam, None) == INVALID
    assert lib.fa_apply_step(None, P(step), n, None) == INVALID
    assert lib.fa_mean_f32(x, 0, P(g), n, None) == INVALID
    assert lib.fa_weighted_mix(x, w, K, P(g), 0.5, 9, None, n, None) == INVALID                        # unknown mode
    assert lib.fa_weighted_mix(x, w, K, P(g), 0.5, 3, None, n, None) == INVALID                        # convex_one: K = 1
    assert lib.fa_weighted_mix(xm, wm, many, P(g), 0.5, 1, None, n, None) == WORKSPACE
    assert lib.fa_async_mix(None, 0.5, 0, 0, P(g), None, n, None) == INVALID
    assert lib.fa_async_mix(P(xs[0]), 0.5, 0, 0, C.c_void_p(g.data_ptr() + 2), None, n, None) == ALIGN
    assert lib.fa_buff_accum(P(xs[0]), 0.5, 0, P(g), None, 1, n, None) == INVALID
    torch.cuda.synchronize()
    assert torch.equal(g, before)
    assert lib.fa_mean_f32(x, K, P(g), 0, None) == 0          # n = 0: a valid empty call
    assert b"workspace" in lib.fa_error_string(WORKSPACE).lower() or lib.fa_error_string(WORKSPACE)
