"""Host-side logic that needs no GPU: the C ABI surface, layout / row packing, chunking rule,
workload tables, the serial harness (plumbing of benchmark configuration 1) driven by the oracle."""

import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import scenarios as S
from appfl_amd import _native as N
from appfl_amd import layout as L
from appfl_amd import workloads as W
from oracle import aggregators as OA

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "appfl_amd.h")).read()
    declared = set(re.findall(r"^(?:int|float|void|const char\*)\s+(fa_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 25
    lib = N.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    assert lib.fa_version() >= 100 and lib.fa_kchunk() == 128
    assert lib.fa_error_string(-2) == b"buffer not aligned to its element size"


def test_argument_checks_without_touching_the_gpu():
    lib = N.load()
    assert lib.fa_reduce_opt(None, None, 0, None, None, None, 0, 0, 0, 0, 0, 0, None, None, None) == -1
    assert lib.fa_tune(3, 8, 1, 0, 4) == -1 and lib.fa_tune(4, 8, 1, 0, 4) == 0
    assert lib.fa_tune(0, 8, 1, 0, 2) == 0          # back to the defaults
    assert lib.fa_admm_ii(None, None, None, None, 0, 1, None, 0, None, None, None, None) == -1
    assert lib.fa_admm_ice(None, None, None, None, 3, None, 8, None, None, None, None) == -1
    assert lib.fa_pack_host(None, None, None, None, 0, 1) == -1
    assert lib.fa_pack_host_window(None, None, None, None, 0, 0, 1, 1) == -1
    assert lib.fa_host_copy_fresh(None, None, 8, 1) == -1


def test_pack_host_matches_numpy():
    rng = np.random.default_rng(0)
    sizes = [5, 0, 4099, 3 * 1024 * 1024 + 7, 17]
    srcs = [rng.integers(0, 255, s, dtype=np.uint8) for s in sizes]
    offs, pos = [], 3
    for s in sizes:
        offs.append(pos)
        pos += s + 5
    for threads in (1, 4, 7):
        dst = np.zeros(pos, np.uint8)
        T = len(sizes)
        rc = N.load().fa_pack_host(
            C.c_void_p(dst.ctypes.data), (C.c_void_p * T)(*[a.ctypes.data if a.size else None for a in srcs]),
            (C.c_size_t * T)(*offs), (C.c_size_t * T)(*sizes), T, threads)
        assert rc == 0
        want = np.zeros(pos, np.uint8)
        for a, o in zip(srcs, offs):
            want[o:o + a.size] = a
        assert np.array_equal(dst, want), threads


def test_pack_host_worker_pool_many_jobs_callers_and_a_forked_child():
    """The pack jobs run on the library's sleeping workers: job after job, more parts than
    workers (threads of its own), several callers at once (the late ones start their own
    threads), and a forked child (which has none of the parent's workers) all copy the same
    bytes."""
    import os
    import threading
    rng = np.random.default_rng(1)
    sizes = [2 * 1024 * 1024 + 3, 1, 0, 1024 * 1024 + 11, 5 * 1024 * 1024]
    srcs = [rng.integers(0, 255, s, dtype=np.uint8) for s in sizes]
    offs, pos = [], 0
    for s in sizes:
        offs.append(pos)
        pos += s + 9
    want = np.zeros(pos, np.uint8)
    for a, o in zip(srcs, offs):
        want[o:o + a.size] = a
    T = len(sizes)

    def run(threads):
        dst = np.zeros(pos, np.uint8)
        rc = N.load().fa_pack_host(
            C.c_void_p(dst.ctypes.data), (C.c_void_p * T)(*[a.ctypes.data if a.size else None for a in srcs]),
            (C.c_size_t * T)(*offs), (C.c_size_t * T)(*sizes), T, threads)
        return rc == 0 and np.array_equal(dst, want)

    for threads in (8, 2, 16, 17, 64, 8, 8):
        assert run(threads), threads
    bad = []
    callers = [threading.Thread(target=lambda: bad.extend(t for t in (8, 3, 8, 16) if not run(t)))
               for _ in range(4)]
    for t in callers:
        t.start()
    for t in callers:
        t.join()
    assert not bad
    pid = os.fork()
    if pid == 0:
        os._exit(0 if run(8) and run(8) else 1)
    assert os.waitpid(pid, 0)[1] == 0
    assert run(8)


def test_layout_classes_offsets_and_rows():
    model = S.ToyModel()
    state = model.state_dict()
    named = {n for n, _ in model.named_parameters()}
    lay = L.Layout([(n, tuple(t.shape), t.dtype) for n, t in state.items()], named)
    assert lay.count[L.STEP] == 108 + 4 + 4 + 640 + 10 and lay.count[L.MEAN_F32] == 8
    assert lay.count[L.MEAN_I64] == 1 and lay.count[L.MEAN_I32] == 0
    assert lay.names == list(state)
    assert lay.seg_byte[L.MEAN_I64] % 16 == 0 and lay.row_bytes % 256 == 0
    spans = sorted((e.byte_offset, e.byte_offset + e.numel * e.dtype.itemsize) for e in lay.entries)
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b <= c
    flats = {"f32": torch.arange(lay.n_f32, dtype=torch.float32), "i64": torch.tensor([5]),
             "i32": torch.zeros(1, dtype=torch.int32)}
    v = lay.views(flats)
    assert v["bn.num_batches_tracked"].shape == () and int(v["bn.num_batches_tracked"]) == 5
    assert v["bn.running_mean"].data_ptr() == flats["f32"][lay.count[L.STEP]:].data_ptr()
    nomodel = L.Layout([(n, tuple(t.shape), t.dtype) for n, t in state.items()], None)
    assert nomodel.count[L.MEAN_F32] == 0
    with pytest.raises(NotImplementedError):
        L.Layout([("h", (3,), torch.int8)], None)
    # 16-bit floats stay 2 bytes wide: same classes and element offsets as the fp32 layout, half the
    # bytes in a row, a 16-bit flat vector; a state dict that mixes float kinds is refused
    half = L.Layout([(n, tuple(t.shape), torch.bfloat16 if t.dtype == torch.float32 else t.dtype)
                     for n, t in state.items()], named)
    assert half.half_dtype == half.float_dtype == torch.bfloat16
    assert lay.half_dtype is None and lay.float_dtype == torch.float32
    assert [(e.name, e.cls, e.offset) for e in half.entries] == [(e.name, e.cls, e.offset) for e in lay.entries]
    assert half.seg_byte[L.MEAN_F32] == 2 * half.count[L.STEP] and half.seg_byte[L.MEAN_I64] % 16 == 0
    assert half.row_bytes % 256 == 0 and half.row_bytes < lay.row_bytes
    e = half.by_name["fc.weight"]
    assert e.dtype == torch.bfloat16 and e.byte_offset == 2 * e.offset
    hflats = half.alloc_state("cpu")
    assert hflats["f32"].dtype == torch.bfloat16 and hflats["i64"].dtype == torch.int64
    assert half.views(hflats)["fc.weight"].dtype == torch.bfloat16
    assert half.matches({n: (t.to(torch.bfloat16) if t.dtype == torch.float32 else t) for n, t in state.items()})
    assert not half.matches(state) and lay.matches(state)
    for kinds in ((torch.float32, torch.bfloat16), (torch.float16, torch.bfloat16)):
        with pytest.raises(NotImplementedError):
            L.Layout([("a", (3,), kinds[0]), ("b", (3,), kinds[1])], None)
    L.Layout([("a", (3,), torch.float16), ("b", (3,), torch.float64), ("c", (), torch.int64)], None)


def test_chunking_rule():
    sizes = {"b": 400, "a": 300, "huge": 5000, "c": 700, "d": 100}
    assert L.split_names_by_size(sizes, 1000) == [["a", "b"], ["c", "d"], ["huge"]]
    assert S.chunk_keys_by_size(S.TOY, 600) == [
        ["bn.bias", "bn.num_batches_tracked", "bn.running_mean", "bn.running_var", "bn.weight",
         "conv.weight", "fc.bias"], ["fc.weight"]]


def test_workload_tables_have_the_published_parameter_counts():
    assert W.numel(W.resnet(18), True) == 11689512 and len(W.resnet(18)) == 122
    assert W.numel(W.resnet(50), True) == 25557032 and len(W.resnet(50)) == 320
    assert W.numel(W.vit_b_16()) == 86567656 and len(W.vit_b_16()) == 152
    assert W.numel(W.mnist_cnn()) == 582026 and len(W.mnist_cnn()) == 8


def test_serial_harness_runs_config1_plumbing_with_the_oracle():
    """MNIST CNN, FedAvg, 2 clients, serial simulation: Future semantics and the round result."""
    from appfl_amd.harness import SerialSyncHarness
    rows = W.mnist_cnn()
    spec = {n: (shape, "f32") for n, shape, _, _ in rows}
    h = SerialSyncHarness(OA.OracleFedAvg(None, {"client_weights_mode": "sample_size"}, None), 2)
    h.set_sample_size("c0", 100)
    h.set_sample_size("c1", 300)
    with pytest.raises(ValueError):
        h.get_parameters(serial_run=True)
    a, b = S.make_state(spec, seed=1), S.make_state(spec, seed=2)
    f0 = h.global_update("c0", a, blocking=False, val_accuracy=0.1)
    assert not f0.done() and h.get_num_global_epochs() == 0
    f1 = h.global_update("c1", b, blocking=False, val_accuracy=0.2)
    assert f0.done() and f1.done() and f0.result() is f1.result()
    assert h.get_num_global_epochs() == 1
    out = f0.result()
    # first round without a model: g = a, step = 0.25 (a - a) + 0.75 (b - a)
    want = a["fc2.bias"].numpy() + (np.float32(0.75) * (b["fc2.bias"].numpy() - a["fc2.bias"].numpy())
                                    + np.float32(0.25) * np.float32(0.0))
    assert np.allclose(out["fc2.bias"], want, rtol=0, atol=1e-7)


def test_pack_host_window_and_fresh_copy():
    rng = np.random.default_rng(1)
    sizes = [300, 70000, 0, 5 * 1024 * 1024 + 3]
    srcs = [rng.integers(0, 255, s, dtype=np.uint8) for s in sizes]
    offs, pos = [], 16
    for s in sizes:
        offs.append(pos)
        pos += s + 9
    want = np.zeros(pos, np.uint8)
    for a, o in zip(srcs, offs):
        want[o:o + a.size] = a
    dst = np.zeros(pos, np.uint8)
    T = len(sizes)
    args = ((C.c_void_p * T)(*[a.ctypes.data if a.size else None for a in srcs]),
            (C.c_size_t * T)(*offs), (C.c_size_t * T)(*sizes), T)
    step = 1 << 20
    for lo in range(0, pos, step):       # windows cut through the pieces
        assert N.load().fa_pack_host_window(C.c_void_p(dst.ctypes.data), *args, lo, min(pos, lo + step), 5) == 0
        done = min(pos, lo + step)
        assert np.array_equal(dst[:done], want[:done]) and not dst[done:].any()
    fresh = np.empty(9 * 1024 * 1024 + 11, np.uint8)
    src = rng.integers(0, 255, fresh.size, dtype=np.uint8)
    assert N.load().fa_host_copy_fresh(C.c_void_p(fresh.ctypes.data), C.c_void_p(src.ctypes.data),
                                       fresh.size, 6) == 0
    assert np.array_equal(fresh, src)


def test_staleness_functions_and_their_errors():
    """Host scalars of the asynchronous aggregators (fedasync_aggregator.py:266-277,
    fedqueue_aggregator.py:177-191): same doubles, same exceptions -- including the reference's
    un-runnable 'exponential' FedQueue function, which raises TypeError when called."""
    from appfl_amd.aggregator.staleness import fedasync_staleness, fedqueue_staleness
    assert [fedasync_staleness("constant")(u) for u in (0, 3)] == [1, 1]
    assert fedasync_staleness("polynomial", a=0.5)(3) == (3 + 1) ** (-0.5)
    hinge = fedasync_staleness("hinge", a=10, b=2)
    assert hinge(2) == 1 and hinge(5) == 1.0 / (10 * (5 - 2) + 1.0)
    assert fedqueue_staleness("harmonic", beta=0.5)(4) == 1.0 / (1.0 + 0.5 * 4)
    assert fedqueue_staleness("constant")(9) == 1.0
    with pytest.raises(NotImplementedError):
        fedasync_staleness("nope")
    with pytest.raises(NotImplementedError):
        fedqueue_staleness("polynomial", a=1)
    with pytest.raises(KeyError):
        fedasync_staleness("hinge", a=1)
    exp = fedqueue_staleness("exponential", beta=0.1)        # building it works, as in the reference
    with pytest.raises(TypeError):
        exp(2)


def test_host_copy_crc32_matches_zlib_per_chunk():
    """The fused threaded copy + CRC-32 of the archive writer: every piece's checksum is zlib's,
    the copy is exact, for lengths around the 8-byte stride and the chunk boundaries, with one and
    several threads; the pieces combine to the checksum of the whole."""
    import zlib

    from appfl_amd import wire
    lib = N.load()
    assert lib.fa_host_copy_crc32(None, None, 8, 4, None, 1) == -1
    rng = np.random.default_rng(5)
    for n, chunk, threads in [(1, 7, 1), (7, 8, 1), (8, 8, 2), (1000, 64, 3), (65537, 4096, 8),
                              (3_000_001, 1 << 20, 8), (5_000_000, 1 << 20, 1)]:
        src = rng.integers(0, 256, n, dtype=np.uint8)
        dst = np.zeros(n + 8, np.uint8)
        pieces = (n + chunk - 1) // chunk
        crcs = (C.c_uint32 * pieces)()
        rc = lib.fa_host_copy_crc32(C.c_void_p(dst.ctypes.data), C.c_void_p(src.ctypes.data), n, chunk, crcs, threads)
        assert rc == 0
        assert np.array_equal(dst[:n], src) and not dst[n:].any()
        whole = 0
        for i in range(pieces):
            part = src[i * chunk:(i + 1) * chunk].tobytes()
            assert crcs[i] == zlib.crc32(part), (n, chunk, i)
            whole = crcs[i] if i == 0 else wire.crc32_combine(whole, crcs[i], len(part))
        assert whole == zlib.crc32(src.tobytes())
