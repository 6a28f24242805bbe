"""Development check for the 16-bit late-epilogue kernel: runs a few optimizer rounds on fixed
synthetic inputs and writes the bits of g / m / v; run once with APPFL_AMD_HALF_LATE=0 and once
with =1 and compare the files (cmp)."""
import sys

import torch

sys.path.insert(0, ".")
from appfl_amd import half_ops  # noqa: E402


def main():
    out, dt = sys.argv[1], torch.bfloat16
    ops16 = half_ops.HalfOps(dt)
    blobs = []
    for n, K, opt in ((25_557_032, 64, "adam"), (25_557_032, 24, "yogi"), (33_000_001, 17, "adagrad"),
                      (20_000_008, 40, "avgm")):
        gen = torch.Generator(device="cuda").manual_seed(n + K)
        g = (torch.randn(n, device="cuda", generator=gen) * 0.05).to(dt)
        m, v = torch.zeros_like(g), torch.zeros_like(g)
        ops16.fill(v, 1e-6 if opt != "avgm" else 0.0)
        xs = [(g.float() + torch.randn(n, device="cuda", generator=gen) * 0.02).to(dt) for _ in range(K)]
        w = [(k + 1) / (K * (K + 1) / 2) for k in range(K)]
        for r in range(3):
            ops16.reduce_opt(xs, w, g, None if opt == "avgm" else m, v, opt, lr=0.1, beta1=0.9, beta2=0.99, tau=1e-3)
        torch.cuda.synchronize()
        blobs += [g.view(torch.int16).cpu(), m.view(torch.int16).cpu(), v.view(torch.int16).cpu()]
        del xs
    torch.save(blobs, out)
    print("wrote", out, [int(b.long().sum()) for b in blobs])


if __name__ == "__main__":
    main()
