"""BASELINE configs[3] on ONE GPU (the strong-scaling curve's N = 1 point): the whole 256^3 p=4 fp64 box, 1 076 890 625
DOFs, heterogeneous skull medium.  Size-independent properties of the operator at that size (K 1 = 0, symmetry, sum of the
lumped mass = volume), then the bench line.  Run through gpurun; the host-side mesh + block layout take several minutes.

    python tools/gpu_config3_full.py [ncells=256] > gpurun_out/r3_c4_n1_props.txt
"""
import sys
import time

import numpy as np

sys.path.insert(0, "fenicsx-fus_amd")
import fenicsxfus_amd as fa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = 0.12 * n / 64
t0 = time.time()
mesh = fa.BoxMesh([0, 0, 0], [L, L, L], (n, n, n))
V = fa.FunctionSpace(mesh, 4)
print(f"mesh + space: {time.time() - t0:.1f} s, {V.num_dofs} dofs, {mesh.num_cells} cells", flush=True)
t0 = time.time()
ctx = fa.Context(0, geometry="trilinear")
d = fa.SpectralOperatorData(V, ctx)
print(f"operator data (layout + upload): {time.time() - t0:.1f} s, {d.info()}", flush=True)
nd, nc = V.num_dofs, mesh.num_cells
rng = np.random.default_rng(0)
x, z = rng.standard_normal(nd), rng.standard_normal(nd)
cx = mesh.cell_centroids()[:, 0]
rho = np.where((cx > 0.4 * L) & (cx < 0.5 * L), 1850.0, 1000.0)
coef = -1.0 / rho                                      # the Linear model's operator coefficient (Linear.hpp:154-155)
y = d.stiffness(x, coef, np.zeros(nd))
scale = np.abs(y).max()
one = np.abs(d.stiffness(np.ones(nd), coef, np.zeros(nd))).max()
yz = d.stiffness(z, coef, np.zeros(nd))
zy, xz = z @ y, x @ yz
mm = d.mass(np.ones(nd), np.ones(nc), np.zeros(nd)).sum()
print(f"|K 1|_max / |K x|_max = {one / scale:.2e}   (z, K x) vs (x, K z) rel diff = {abs(zy - xz) / abs(zy):.2e}   "
      f"sum(K x) / sum|K x| = {abs(y.sum()) / np.abs(y).sum():.2e}   sum(m) / volume - 1 = {mm / L**3 - 1:.2e}", flush=True)
assert one < 1e-11 * scale and abs(zy - xz) < 1e-10 * abs(zy) and abs(mm - L**3) < 1e-12 * L**3
print("properties ok")
d.close()
ctx.close()
