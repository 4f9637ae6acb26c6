"""Phase timeline of the fused block kernel (experiment build: python fenicsx-fus_amd/build.py --dev
--name trace -DFUS_TRACE; run with FUSMI_LIB=abl/libfusmi_trace.so).  Prints per-phase durations of
the last launch (prologue / element trips / epilogue) and how many blocks a CU holds at once."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, "fenicsx-fus_amd")
import fenicsxfus_amd as fa  # noqa: E402
from fenicsxfus_amd import _abi  # noqa: E402

geom = sys.argv[1] if len(sys.argv) > 1 else "stream"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dtype = np.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else np.float64   # f32: build with -DFUS_TRACE_BITS=32
L = 0.12
mesh = fa.BoxMesh([0, 0, 0], [L, L, L], (n, n, n), dtype=dtype)
V = fa.FunctionSpace(mesh, P)
tags = fa.tag_box_boundary(mesh)
nc = mesh.num_cells
dt = 0.5 * (L / n) / (1500.0 * P**2)
ctx = fa.Context(0, geometry=geom)
m = fa.LinearSpectralExplicit(mesh, tags, P, np.full(nc, 1500.0, dtype), np.full(nc, 1000.0, dtype), 0.5e6, 6e4, 1500.0, 4, dt, V=V, ctx=ctx)
m.init()
m.rk4_steps(0.0, dt, 5)
ctx.synchronize()
nb = m.data.info()["nblocks"]
buf = np.zeros((nb, 8), dtype=np.uint64)
rc = _abi.lib().fus_debug_trace(buf.ctypes.data_as(C.c_void_p), C.c_longlong(nb))
assert rc == 0
t = buf[:, :4].astype(np.float64) / 100.0          # 100 MHz -> microseconds
t0 = t[:, 0].min()
pro, trips, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print(f"geometry={geom} blocks={nb} kernel span {t[:, 3].max() - t0:.1f} us")
for name, d in (("prologue", pro), ("trips", trips), ("epilogue", epi), ("block total", t[:, 3] - t[:, 0])):
    print(f"  {name:12s} mean {d.mean():6.2f}  median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
if buf[:, 5].any():   # wave 0's first three element trips (stamps 5, 6, 7) and the rest up to the barrier
    tt = buf[:, 5:8].astype(np.float64) / 100.0
    for name, d in (("trip 0", tt[:, 0] - t[:, 1]), ("trip 1", tt[:, 1] - tt[:, 0]), ("trip 2", tt[:, 2] - tt[:, 1]),
                    ("trip 3 + barrier", t[:, 2] - tt[:, 2])):
        print(f"  {name:16s} mean {d.mean():6.2f}  median {np.median(d):6.2f} us")
# start-time histogram: how staggered are the blocks
order = np.argsort(t[:, 0])
print("  first 8 block start offsets (us):", np.round(t[order[:8], 0] - t0, 2))
cu = buf[:, 4]
print("  distinct CU ids:", len(np.unique(cu)))
# concurrency per CU at mid-kernel
mid = t0 + 0.5 * (t[:, 3].max() - t0)
act = (t[:, 0] <= mid) & (t[:, 3] >= mid)
print("  blocks in flight at mid-kernel:", int(act.sum()), "per CU", act.sum() / len(np.unique(cu)))
m.close()
