"""Development tool: A/B of the persistent kernel's meeting options inside the benchmark's own
set-up (same slab, same launches).  usage: python tools/ab_phases.py "<nt,nt,...>" <workload>..."""
import sys

import torch

sys.path.insert(0, ".")
import bench
from appfl_amd import ops, workloads as W

nts = [int(t) for t in sys.argv[1].split(",")]
for name in sys.argv[2:]:
    b = bench.SyncBench(W.WORKLOADS[name], 0)
    res = {}
    for rep in range(3):
        for nt in nts:
            ops.tune(0, 8, nt, 0, 2)
            for _ in range(3):
                b.step()
            ts = []
            for _ in range(15):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                b.step((e0, e1))
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            res.setdefault(nt, []).append(ts[len(ts) // 2])
    print(name, " ".join("nt=%d(sync %s, every %d): %.4f" % (nt, "off" if nt & 16 else ("both" if nt & 32 else "pre"), nt >> 8, min(v))
                         for nt, v in res.items()), flush=True)
    del b
    torch.cuda.empty_cache()
ops.tune()
