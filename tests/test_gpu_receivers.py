"""Device-side receiver sampling: tensor-Lagrange evaluation of the RESIDENT solution at fixed points,
every k steps, without copying the state to the host.  Reference: Function.eval at the cells found by
compute_eval_params (python/src/fenicsxfus/utils.py:10-47; cpp/mwe/parallel_eval_line/main.cpp:49-84).
Checked against the host interpolant of evaluate.py (itself exact on polynomials of the space's degree), against the
analytic field, and -- sampled during an RK4 run -- against the oracle's state interpolated the same way."""
import os

import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd.evaluate import evaluate
from util import Problem

pytestmark = pytest.mark.gpu
F0, P0, S0 = 0.5e6, 60000.0, 1500.0


def _model(mesh, V, P, dtype=np.float64, ctx=None):
    nc = mesh.num_cells
    dt = 0.5 * mesh.hmin() / (1500.0 * P**2) if hasattr(mesh, "hmin") else 1e-8
    return fa.LinearSpectralExplicit(mesh, fa.tag_box_boundary(mesh), P, np.full(nc, 1500.0, dtype), np.full(nc, 1000.0, dtype),
                                     F0, P0, S0, 4, dt, V=V, ctx=ctx or fa.Context(0)), dt


@pytest.mark.parametrize("P", [2, 4, 7])
def test_sample_matches_host_interpolant_and_polynomial(P):
    mesh = fa.BoxMesh([0, 0, 0], [1.0, 0.8, 0.6], (3, 2, 2))
    V = fa.FunctionSpace(mesh, P)
    mdl, _ = _model(mesh, V, P)
    X = V.tabulate_dof_coordinates()
    f = lambda Y: Y[:, 0] ** 2 - 2 * Y[:, 1] * Y[:, 2] + Y[:, 0] * Y[:, 1] * Y[:, 2] + 1.0  # noqa: E731  (degree <= 2 per variable)
    g = lambda Y: np.sin(3 * Y[:, 0]) * np.cos(2 * Y[:, 1]) + Y[:, 2]                       # noqa: E731
    rng = np.random.default_rng(1)
    pts = np.vstack([rng.uniform([0, 0, 0], [1.0, 0.8, 0.6], size=(257, 3)),
                     [[0, 0, 0], [1.0, 0.8, 0.6], [0.5, 0.4, 0.3], [1.0 / 3, 0.0, 0.6]],   # vertices, a node, faces
                     [[1.5, 0.1, 0.1]]])                                                    # outside: dropped
    on = mdl.set_receivers(pts)
    assert len(on) == len(pts) - 1 and 261 not in on
    mdl.set_state(u=f(X), v=g(X))
    us, vs = mdl.sample("u"), mdl.sample("v")
    assert np.abs(us - f(pts[on])).max() < 1e-13                     # the space holds f exactly
    assert np.abs(us - evaluate(V, f(X), pts[on])).max() < 1e-13
    assert np.abs(vs - evaluate(V, g(X), pts[on])).max() < 1e-13
    assert np.array_equal(us, mdl.sample("u"))                        # fixed summation order
    mdl.close()


def test_sample_fp32_and_quadrilaterals():
    mesh = fa.BoxMesh([0, 0, 0], [1.0, 0.8, 0.6], (3, 2, 2), dtype=np.float32)
    V = fa.FunctionSpace(mesh, 4)
    mdl, _ = _model(mesh, V, 4, np.float32)
    X = V.tabulate_dof_coordinates().astype(np.float64)
    g = lambda Y: np.sin(3 * Y[:, 0]) * np.cos(2 * Y[:, 1]) + Y[:, 2]                       # noqa: E731
    pts = np.random.default_rng(2).uniform([0, 0, 0], [1.0, 0.8, 0.6], size=(100, 3))
    on = mdl.set_receivers(pts)
    mdl.set_state(u=g(X).astype(np.float32))
    assert np.abs(mdl.sample("u") - evaluate(V, g(X), pts[on])).max() < 2e-5
    mdl.close()
    # 2-D (config 1's element type)
    q = fa.BoxMesh([0, 0], [1.0, 0.8], (4, 3))
    Vq = fa.FunctionSpace(q, 4)
    nc = q.num_cells
    m2 = fa.LinearSpectralExplicit(q, fa.tag_box_boundary(q), 4, np.full(nc, 1500.0), np.full(nc, 1000.0), F0, P0, S0, 4, 1e-8,
                                   V=Vq)
    Xq = Vq.tabulate_dof_coordinates()
    f2 = lambda Y: Y[:, 0] ** 4 - 2 * Y[:, 1] ** 3 * Y[:, 0] + Y[:, 0] * Y[:, 1] + 1.0      # noqa: E731
    p2 = np.random.default_rng(3).uniform([0, 0], [1.0, 0.8], size=(150, 2))
    on = m2.set_receivers(p2)
    assert len(on) == 150
    m2.set_state(u=f2(Xq))
    assert np.abs(m2.sample("u") - f2(p2)).max() < 1e-13
    m2.close()


def test_receivers_on_the_reference_mesh_fixture():
    """The reference operator test's own Gmsh mesh (unstructured first-order hexahedra; data-only fixture)."""
    from fenicsxfus_amd.unstructured import VTK_TO_TENSOR, HexFunctionSpace, HexMesh
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_test_operators3d_mesh.npz"))
    mesh = HexMesh(gold["geometry"], gold["topology_vtk"][:, VTK_TO_TENSOR])
    V = HexFunctionSpace(mesh, 4)
    nc = mesh.num_cells
    tags = mesh.facet_tags(gold["facet_topology"], gold["facet_values"])
    mdl = fa.LinearSpectralExplicit(mesh, tags, 4, np.full(nc, 1.5), np.full(nc, 1.0), 10.0, 1.0, 1.5, 4, 1e-4, V=V)
    X = V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(np.pi * X[:, 1])                    # the input of test_operators3d/main.cpp:60-67
    pts = np.random.default_rng(0).uniform(0.02, 0.98, size=(300, 3))
    on = mdl.set_receivers(pts)
    assert len(on) == 300
    mdl.set_state(u=u)
    s = mdl.sample("u")
    assert np.abs(s - evaluate(V, u, pts)).max() < 1e-13            # same interpolant as the host code
    assert np.abs(s - np.sin(pts[:, 0]) * np.cos(np.pi * pts[:, 1])).max() < 1e-6   # spectral accuracy
    mdl.close()


def test_recording_during_rk4_matches_oracle(orc):
    """Receivers sampled every 2 steps INSIDE fus_model_rk4_steps (no host copy of the state) against the oracle's
    state after the same number of steps, interpolated on the host."""
    P, n, hi = 4, (4, 3, 3), [0.016, 0.012, 0.012]
    pr = Problem(orc, n, P, hi=hi, perturb=0.1)
    nc = pr.mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    tags = fa.tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    dt = 0.5 * (hi[0] / n[0]) / (1500.0 * P**2)
    mdl = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, F0, P0, S0, 4, dt, V=pr.V)
    line = np.stack([np.linspace(0.0, hi[0], 50), np.full(50, 0.5 * hi[1]), np.full(50, 0.5 * hi[2])], axis=1)
    on = mdl.set_receivers(line)     # the reference's "evaluate on a line" (parallel_eval_line/main.cpp:49-57)
    assert len(on) == 50
    mdl.init()
    mdl.record(every=2, capacity=8, which="u")
    mdl.rk4_steps(0.0, dt, 6)
    times, rec = mdl.records()
    assert rec.shape == (3, 50) and np.allclose(times, [2 * dt, 4 * dt, 6 * dt], rtol=1e-12)
    # oracle: three runs of 2, 4, 6 steps from rest (+ a ~1e-12 dt remainder step: `while (t < tf)`, Linear.hpp:270)
    for k, ns in enumerate((2, 4, 6)):
        u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
        orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, P0, S0, 0.0, ns * dt * (1 + 1e-12), dt, u, v)
        ref = evaluate(pr.V, u, line)
        assert np.abs(u).max() > 0
        assert np.abs(rec[k] - ref).max() < 1e-10 * np.abs(u).max()
    # recording continues across calls and stops at capacity; a fresh record() restarts it
    mdl.rk4_steps(6 * dt, dt, 20)
    times, rec = mdl.records()
    assert rec.shape[0] == 8
    mdl.record(every=1, capacity=4, which="v")
    mdl.rk4_steps(26 * dt, dt, 2)
    times, rec = mdl.records()
    assert rec.shape == (2, 50) and np.abs(rec[1] - mdl.sample("v")).max() == 0.0
    mdl.close()
