"""Development tool: what plain streaming mixes reach on this GPU (torch's own kernels), as a
yardstick for the write-heavy kernels: copy (1 read : 1 write), add (2 : 1), in-place add
(2 reads, 1 write to one of the sources), in-place scale (1 : 1 on one buffer)."""
import torch

n = 86_567_656 * 3
x = torch.randn(n, device="cuda")
y = torch.randn(n, device="cuda")
z = torch.empty(n, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for name, fn, streams in (("copy z = x", lambda: z.copy_(x), 2),
                          ("add z = x + y", lambda: torch.add(x, y, out=z), 3),
                          ("in-place x += y", lambda: x.add_(y), 3),
                          ("in-place x *= 1.0001", lambda: x.mul_(1.0001), 2),
                          ("fill z = 0", lambda: z.zero_(), 1),
                          ("sum of x (read only)", lambda: x.sum(), 1)):
    t = timed(fn)
    print(f"{name:24s} {t:.4f} ms  {streams * n * 4 / t / 1e9:.2f} TB/s ({streams * n * 4 / t / 8e9:.3f} of peak)")

# the library's own write-only kernel (write-through 128-bit stores are not used by fa_fill_f32;
# plain stores) and its read-only synthetic check, for the write-bandwidth model in the design notes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from appfl_amd import ops  # noqa: E402

t = timed(lambda: ops.fill(z, 1.5))
print(f"{'ops.fill (write only)':24s} {t:.4f} ms  {n * 4 / t / 1e9:.2f} TB/s ({n * 4 / t / 8e9:.3f} of peak)")
