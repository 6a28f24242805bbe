"""Development tool: does a sharded round's kernel lose time beside a running all-gather, and
which kernel loses less?  One GPU: a stand-in for RCCL's gather (tools/gather_standin.hip: a few
persistent blocks writing the gathered bytes into the model buffer) runs on a second stream while
the fused round of a block row runs on the first.

    python tools/coresidency_probe.py [n_row,K[,opt]] ...

For each shape: the round alone (default routing = persistent kernel where the library picks it,
and kept off the persistent kernels), the stand-in alone, and both at once -- the round's time by
events on its stream and the wall time until both are done; then a pipeline of four block rows
(kernel on row p+1 beside the gather of row p) in both routings.
"""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from appfl_amd import ops  # noqa: E402

HP = dict(lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
LIB = os.path.join(ROOT, "tools", "lib", "libgather_standin.so")


def standin():
    if not os.path.exists(LIB):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "gather_standin.hip"), "-o", LIB])
    lib = C.CDLL(LIB)
    lib.standin_launch.restype = C.c_int
    lib.standin_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return lib


def med(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2]


def main():
    lib = standin()
    shapes = sys.argv[1:] or ["81250000,32,none", "40625000,32,none", "81250000,32,adam", "25557032,64,adam"]
    dev = torch.device("cuda")
    side = torch.cuda.Stream()
    for shape in shapes:
        f = shape.split(",")
        n, K, opt = int(f[0]), int(f[1]), (f[2] if len(f) > 2 else "none")
        rows_n = 4
        slab = torch.empty((K, n), device=dev)
        g = torch.empty(rows_n * n, device=dev)
        ops.synth_fill(g, 7, scale=0.05)
        for k in range(K):
            ops.synth_fill(slab[k], 1000 + k, scale=0.01, base=g, n=n)
        m = torch.zeros(rows_n * n, device=dev) if opt in ("adam", "yogi", "adagrad") else None
        v = torch.full((rows_n * n,), 1e-6, device=dev) if opt != "none" else None
        out = torch.empty(rows_n * n * 2, device=dev)            # the gathered model of two ranks
        xs, w = list(slab), [1.0 / K] * K
        alg = 4 * (K + {"none": 2, "avgm": 4}.get(opt, 6)) * n
        name = ops.reduce_kernel_name(K, n, opt)
        print(f"\n== row of {n} elements x {K} clients, {opt}: default routing takes {name}", flush=True)

        def round_(p, persistent):
            a = p * n
            ops.reduce_opt(xs, w, g[a:a + n], None if m is None else m[a:a + n],
                           None if v is None else v[a:a + n], opt, persistent=persistent, **HP)

        def gather(p, blocks, pause, regs, nbytes):
            a = p * n * 2
            rc = lib.standin_launch(out[a:].data_ptr(), None, nbytes, blocks, pause, regs, side.cuda_stream)
            assert rc == 0, rc

        def time_round(persistent, beside=None, reps=9):
            ts, walls = [], []
            for _ in range(reps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                if beside is not None:
                    gather(0, *beside)
                e0.record()
                round_(1, persistent)
                e1.record()
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) * 1e3)
                ts.append(e0.elapsed_time(e1))
            return med(ts), med(walls)

        def time_gather(cfg, reps=5):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(side):
                    e0.record()
                    gather(0, *cfg)
                    e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return med(ts)

        for _ in range(2):
            round_(1, True), round_(1, False)
        alone = {p: time_round(p)[0] for p in (True, False)}
        print(f"round alone: default {alone[True]:.3f} ms (frac {alg / alone[True] / 8e9:.3f}), "
              f"kept off the persistent kernels {alone[False]:.3f} ms (frac {alg / alone[False] / 8e9:.3f})",
              flush=True)
        nbytes = 4 * n                                      # what one peer's block of the row weighs
        for blocks, pause, regs in ((8, 0, 0), (16, 0, 0), (32, 0, 0), (16, 0, 96), (32, 0, 256),
                                    (16, 4, 96), (32, 4, 256), (64, 0, 96)):
            cfg = (blocks, pause, regs, nbytes)
            tg = time_gather(cfg)
            line = f"stand-in {blocks:3d} blocks, {regs:3d} regs, pause {pause}: alone {tg:.3f} ms;"
            for p in (True, False):
                t, wall = time_round(p, cfg)
                line += f" {'default' if p else 'tile   '} round beside it {t:.3f} ms (x{t / alone[p]:.2f}), both done after {wall:.3f} ms;"
            print(line, flush=True)
        # the pipeline of a sharded round: kernel on row p, then its gather on the side stream
        # while the kernel of row p+1 runs
        for blocks, pause, regs in ((16, 0, 96), (32, 4, 256)):
            cfg = (blocks, pause, regs, nbytes)
            for p_ in (True, False):
                walls = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for p in range(rows_n):
                        round_(p, p_)
                        ev = torch.cuda.Event()
                        ev.record()
                        side.wait_event(ev)
                        gather(p, *cfg)
                    torch.cuda.synchronize()
                    walls.append((time.perf_counter() - t0) * 1e3)
                print(f"pipeline of {rows_n} rows, stand-in {blocks} blocks / {regs} regs / pause {pause}: "
                      f"{'default' if p_ else 'tile   '} routing {med(walls):.3f} ms", flush=True)
        del slab, g, m, v, out, xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
