"""Development probe: device -> pinned host copy rates for a result-sized vector (one copy, split
copies on several streams) and host -> device for comparison."""
import sys, time, torch
sys.path.insert(0, ".")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_699_132
dev = torch.randn(n, device="cuda")
pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
src = torch.randn(n).pin_memory()
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2]
mb = n * 4 / 1e6
d = t(lambda: pin.copy_(dev, non_blocking=True)); print("D2H one copy   %.3f ms %.1f GB/s" % (d * 1e3, mb / d / 1e3))
d = t(lambda: dev.copy_(src, non_blocking=True)); print("H2D one copy   %.3f ms %.1f GB/s" % (d * 1e3, mb / d / 1e3))
for parts in (2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    per = (n + parts - 1) // parts
    def split():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                pin[i * per:(i + 1) * per].copy_(dev[i * per:(i + 1) * per], non_blocking=True)
    d = t(split); print("D2H %d streams  %.3f ms %.1f GB/s" % (parts, d * 1e3, mb / d / 1e3))
for parts in (2, 4):
    per = (n + parts - 1) // parts
    def same():
        for i in range(parts):
            pin[i * per:(i + 1) * per].copy_(dev[i * per:(i + 1) * per], non_blocking=True)
    d = t(same); print("D2H %d pieces, one stream %.3f ms %.1f GB/s" % (parts, d * 1e3, mb / d / 1e3))
# both directions at once
s2 = torch.cuda.Stream()
def both():
    with torch.cuda.stream(s2):
        dev2.copy_(src, non_blocking=True)
    pin.copy_(dev, non_blocking=True)
dev2 = torch.empty_like(dev)
d = t(both); print("D2H + H2D together %.3f ms (%.1f GB/s each way)" % (d * 1e3, mb / d / 1e3))
# the result tensors of the aggregators: torch.frombuffer over a numpy view of a pinned uint8 buffer
raw = torch.empty(n * 4 + 4096, dtype=torch.uint8, pin_memory=True)
lease = raw.numpy()[:n * 4]
out = torch.frombuffer(lease, dtype=torch.float32, count=n)
print("frombuffer tensor is_pinned:", out.is_pinned())
d = t(lambda: out.copy_(dev, non_blocking=True)); print("D2H into frombuffer view %.3f ms %.1f GB/s" % (d * 1e3, mb / d / 1e3))
view = raw[:n * 4].view(torch.float32)
print("tensor view is_pinned:", view.is_pinned())
d = t(lambda: view.copy_(dev, non_blocking=True)); print("D2H into tensor view %.3f ms %.1f GB/s" % (d * 1e3, mb / d / 1e3))
from appfl_amd.staging import HostDownloader
dl = HostDownloader()
keep = []
def fetch():
    keep.append(dl.fetch(dev, n))
    if len(keep) > 2: keep.pop(0)
d = t(fetch); print("HostDownloader.fetch %.3f ms %.1f GB/s" % (d * 1e3, mb / d / 1e3))
