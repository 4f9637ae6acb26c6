"""Builds libfusmi.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python fenicsx-fus_amd/build.py [--force]

The .so lands in fenicsx-fus_amd/fenicsxfus_amd/ so it travels with the source snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("fusmi.hip", "layout.cpp")]
DEPS = SRC + [os.path.join(HERE, "csrc", f) for f in ("kernels.hpp", "layout.hpp", "tables.hpp", "geom.hpp")] + [
    os.path.join(HERE, "..", "include", "fusmi.h")]
OUT = os.path.join(HERE, "fenicsxfus_amd", "libfusmi.so")


def build(force: bool = False, verbose: bool = False, dev: bool = False) -> str:
    """dev=True: P=4/fp64-only iteration build into abl/libfusmi_dev.so (use with FUSMI_LIB)."""
    if dev:
        out = os.path.join(HERE, "..", "abl", "libfusmi_dev.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-Wno-unused-value", "-munsafe-fp-atomics", "-DFUS_DEV_BUILD", *SRC, "-o", out,
                               "-ldl", "-Wl,-rpath,/opt/rocm/lib"])
        return out
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-munsafe-fp-atomics", *SRC,
           "-o", OUT, "-ldl", "-Wl,-rpath,/opt/rocm/lib"]  # RCCL is bound at run time (fusmi.hip rccl_load)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
