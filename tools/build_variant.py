"""Builds a variant of the library in which ONE source is compiled with extra -D flags (the
other objects are the main build's): development tool for A/B runs through APPFL_AMD_LIB.

    python tools/build_variant.py NAME SOURCE.hip -DFLAG=VALUE ...
writes appfl_amd/lib/variants/lib_NAME.so
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    name, source, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    G.build_hip()
    vdir = os.path.join(ROOT, "appfl_amd", "lib", "variants")
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for src in G.HIP_SOURCES:
        base = os.path.splitext(os.path.basename(src))[0]
        if os.path.basename(src) == os.path.basename(source):
            obj = os.path.join(vdir, f"{base}_{name}.o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", *[f for f in G.HIPCC_FLAGS if f != "-shared"], *flags,
                                   "-c", src, "-o", obj], cwd=ROOT)
        else:
            obj = os.path.join(ROOT, "appfl_amd", "lib", "obj", base + ".o")
        objs.append(obj)
    out = os.path.join(vdir, f"lib_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs,
                           "-lpthread"], cwd=ROOT)
    print(out)


if __name__ == "__main__":
    main()
