"""BASELINE config 1 on the GPU (2-D rectangle, 128 x 128 quads, Q4, Linear RK4): DOF-updates/s.
A 263 169-DOF problem is launch/latency-bound on an MI355X; recorded for completeness."""
import sys
import time

import numpy as np

sys.path.insert(0, "fenicsx-fus_amd")
import fenicsxfus_amd as fa  # noqa: E402

for n, graph in ((128, 0), (128, 1), (512, 0), (512, 1), (2048, 0)):
    L, P = 0.12 * n / 128, 4
    mesh = fa.BoxMesh([0, 0], [L, L], (n, n))
    V = fa.FunctionSpace(mesh, P)
    tags = fa.tag_box_boundary(mesh)
    nc = mesh.num_cells
    dt = 0.5 * (L / n) / (1500.0 * P**2)
    ctx = fa.Context(0)
    ctx.set_option("graph", graph)        # 1: each RK step replayed as one hipGraph
    m = fa.LinearSpectralExplicit(mesh, tags, P, np.full(nc, 1500.0), np.full(nc, 1000.0), 0.5e6, 6e4, 1500.0, 4, dt,
                                  V=V, ctx=ctx)
    m.init()
    m.rk4_steps(0.0, dt, 20)
    ctx.synchronize()
    steps = 200 if n <= 512 else 50
    t0 = time.perf_counter()
    m.rk4_steps(20 * dt, dt, steps)
    ctx.synchronize()
    el = time.perf_counter() - t0
    print(f"{n}x{n} quads Q4 graph={graph}: {V.num_dofs} dofs, {1e3 * el / steps:.4f} ms/step, {V.num_dofs * steps / el:.4e} DOF-updates/s, "
          f"{640.0 * V.num_dofs * steps / el / 8e12:.3f} of 8 TB/s at B = 640 B", flush=True)
    m.close()
    ctx.close()
