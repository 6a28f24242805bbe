"""Runs `bench.py --only --workload W` once per library variant (APPFL_AMD_LIB) and prints the
kernel time and roofline fraction of each: development tool.

    python tools/ab_variants.py WORKLOAD[,WORKLOAD...] [variant ...]     (no variant: all built ones)
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    workloads = sys.argv[1].split(",")
    vdir = os.path.join(ROOT, "appfl_amd", "lib", "variants")
    names = sys.argv[2:] or sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(vdir, "lib_*.so")))
    libs = [("main", None)] + [(n, os.path.join(vdir, f"lib_{n}.so")) for n in names]
    for rep in range(2):
        for wl in workloads:
            for name, lib in libs:
                env = dict(os.environ)
                if lib:
                    env["APPFL_AMD_LIB"] = lib
                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only", "--no-cpu-baseline",
                                    "--workload", wl, "--steps", "40", "--warmup", "5"], env=env,
                                   capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    print(wl, name, "FAILED", r.stderr[-500:], flush=True)
                    continue
                d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                print(f"{wl:28s} {name:16s} kernel {d['roofline']['kernel_avg_ms']:.4f} ms  "
                      f"min {d['roofline']['kernel_min_ms']:.4f}  frac {d['roofline']['frac']:.4f}", flush=True)


if __name__ == "__main__":
    main()
