import sys, time, torch
sys.path.insert(0, ".")
import bench
from appfl_amd import ops, staging, workloads as W
if len(sys.argv) > 1:
    staging.UPLOAD_WINDOW = int(sys.argv[1]) << 20      # MiB per pack / transfer window
print("upload window", staging.UPLOAD_WINDOW >> 20, "MiB")
import appfl_amd.aggregator as A
wl = W.WORKLOADS["resnet18_fedavg_k100"]
rows, K = wl["table"](), wl["K"]
nf = W.numel([r for r in rows if r[2] == torch.float32])
base = torch.empty(nf, dtype=torch.float32, device="cuda"); ops.synth_fill(base, 7, scale=0.05)
buf = torch.empty_like(base)
def state(seed):
    ops.synth_fill(buf, seed, scale=0.01, base=base); flat = buf.cpu()
    out, off = {}, 0
    for name, shape, dtype, _ in rows:
        n = 1
        for d in shape: n *= d
        if dtype == torch.float32:
            out[name] = flat[off:off + n].view(shape).clone(); off += n
        else:
            out[name] = torch.full(shape, 3, dtype=dtype)
    return out
clients = {k: state(1000 + k) for k in range(K)}
agg = getattr(A, wl["cls"])(None, dict(W.EXAMPLE_CONFIGS[wl["cls"]]), None)
agg.aggregate(clients); agg.aggregate(clients)
ids = list(clients)
import cProfile, pstats
for rep in range(6):
    for c in ids[:-1]: agg.stage(c, clients[c])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); agg.stage(ids[-1], clients[ids[-1]]); t1 = time.perf_counter()
    with agg._lock:
        agg._stage(clients); t2 = time.perf_counter()
        agg._launch_round(K, agg._client_weights(clients), True); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        out = agg._download(); t5 = time.perf_counter()
    print("stage(last) %.2f  _stage %.2f  launch %.2f  kernels %.2f  download %.2f  total %.2f ms" % tuple(1e3*x for x in (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t5-t0)), flush=True)
lay = agg._layout
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); a = agg._dl.fetch(agg._flats["f32"], lay.n_f32); t1 = time.perf_counter()
    b = agg._dl.fetch(agg._flats["i64"], lay.count["mean_i64"]); t2 = time.perf_counter()
    v = lay.views({"f32": a, "f64": torch.empty(0, dtype=torch.float64), "i64": b, "i32": torch.empty(0, dtype=torch.int32)}); t3 = time.perf_counter()
    print("fetch f32 %.3f  fetch i64 %.3f  views %.3f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2)), flush=True)
ts = []
for rep in range(8):
    for c in ids[:-1]: agg.stage(c, clients[c])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); agg.stage(ids[-1], clients[ids[-1]]); out = agg.aggregate(clients); ts.append(time.perf_counter() - t0)
print("after last arrival (stage + aggregate), ms:", " ".join("%.2f" % (1e3 * t) for t in ts), flush=True)
t0 = time.perf_counter(); agg.aggregate(clients); print("unstaged round %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for c in ids[:-1]: agg.stage(c, clients[c])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
agg.stage(ids[-1], clients[ids[-1]]); agg.aggregate(clients)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
