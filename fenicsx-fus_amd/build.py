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


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-function", "-Wno-invalid-offsetof",
         "-munsafe-fp-atomics"]
DEGREES = (2, 3, 4, 5, 6, 7, 8, 9, 10)


def _compile_units(objdir, degrees, extra, verbose):
    """fusmi.hip is compiled once per polynomial degree and scalar type (-DFUS_TU_DEGREE=k -DFUS_TU_DTYPE=64|32:
    the block kernels of that degree) and once as the main unit (C ABI + degree-independent code), concurrently."""
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = [*extra, *os.environ.get("FUSMI_EXTRA_FLAGS", "").split()]   # compiler-flag experiments
    os.makedirs(objdir, exist_ok=True)
    jobs = [([hipcc, *FLAGS, *extra, "-c", SRC[0], "-o", os.path.join(objdir, "fusmi_main.o")]),
            ([hipcc, *FLAGS, *extra, "-c", SRC[1], "-o", os.path.join(objdir, "layout.o")])]
    for k in degrees:
        for bits in (64, 32):
            jobs.append([hipcc, *FLAGS, *extra, f"-DFUS_TU_DEGREE={k}", f"-DFUS_TU_DTYPE={bits}", "-c", SRC[0], "-o",
                         os.path.join(objdir, f"fusmi_p{k}_f{bits}.o")])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}")
        return cmd[-1]

    workers = int(os.environ.get("FUSMI_BUILD_JOBS", min(len(jobs), os.cpu_count() or 1)))
    # heaviest units first so the pool stays busy
    order = jobs[2:][::-1] + jobs[:2]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(run, order))


def _link(objs, out, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # RCCL is bound at run time (fusmi.hip rccl_load)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out, "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = False, dev: bool = False, name: str = "dev", defines=(),
          degree: int = 4) -> str:
    """dev=True: one-degree iteration build into abl/libfusmi_<name>.so (use with FUSMI_LIB); extra -D
    flags select experiment variants (never shipped)."""
    objroot = os.path.join(HERE, "_build")
    if dev:
        out = os.path.join(HERE, "..", "abl", f"libfusmi_{name}.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _link(_compile_units(os.path.join(objroot, name), (degree,),
                             ["-DFUS_DEV_BUILD", f"-DFUS_DEV_DEGREE={degree}", *defines], verbose), out, verbose)
        return out
    if name != "dev":   # full-library variant for A/B runs of compiler flags: abl/libfusmi_<name>.so
        out = os.path.join(HERE, "..", "abl", f"libfusmi_{name}.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _link(_compile_units(os.path.join(objroot, name), DEGREES, list(defines), verbose), out, verbose)
        return out
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    _link(_compile_units(os.path.join(objroot, "full"), DEGREES, [], verbose), OUT, verbose)
    return OUT


if __name__ == "__main__":
    nm = sys.argv[sys.argv.index("--name") + 1] if "--name" in sys.argv else "dev"
    deg = int(sys.argv[sys.argv.index("--degree") + 1]) if "--degree" in sys.argv else 4
    print(build(force="--force" in sys.argv, verbose="-q" not in sys.argv, dev="--dev" in sys.argv, name=nm,
                defines=[a for a in sys.argv if a.startswith("-D")] + (sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []), degree=deg))
