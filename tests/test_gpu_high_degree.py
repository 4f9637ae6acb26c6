"""Degrees 8-10 (the reference's Qdegree map goes to P = 10, cpp/fenicsx-sf/common/spectral_op.hpp:35-44):
a tensor plane has 81-121 columns, two waves work on one element (kernels.hpp, elem_compute_hi).
Operators and the RK4 loop against the oracle on affine and distorted first-order hexahedra, both
accumulation modes, fp32; per-point factors streamed (option "geometry" = stream, and second-order -- 27-node --
hexahedra, which always stream) so that the Qdegree range has no holes (quadrilaterals at these degrees:
tests/test_gpu_quad2d.py); a degree beyond the reference's map is reported as an error."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import tag_box_boundary
from util import Problem

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("P", [8, 9, 10])
@pytest.mark.parametrize("perturb,mode", [(0.2, "trilinear"), (0.0, "affine")])
@pytest.mark.parametrize("det", [0, 1])
def test_operators_vs_oracle(orc, P, perturb, mode, det):
    pr = Problem(orc, (3, 2, 2), P, hi=[1.5, 1.0, 0.8], perturb=perturb)
    c = fa.Context(0, deterministic=det)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == mode
    rng = np.random.default_rng(P)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    y0 = rng.standard_normal(pr.ndofs)
    assert relmax(d.stiffness(x, coef, y0.copy()), y0 + pr.K(x, coef)) < 1e-12
    assert relmax(d.mass(x, coef, y0.copy()), y0 + pr.M(x, coef)) < 1e-13
    w, D = d.tables()
    assert np.allclose(w, pr.wts, rtol=0, atol=1e-15) and np.allclose(D, pr.D, rtol=0, atol=1e-10)
    G, dJ = d.geometry()
    assert relmax(G, pr.G) < 1e-12 and relmax(dJ, pr.detJ) < 1e-13
    d.close()
    c.close()


@pytest.mark.parametrize("P", [8, 10])
def test_fp32_operator(orc, P):
    pr = Problem(orc, (2, 2, 2), P, perturb=0.2, dtype=np.float32)
    c = fa.Context(0)
    d = fa.SpectralOperatorData(pr.V, c)
    x = np.random.default_rng(1).standard_normal(pr.ndofs).astype(np.float32)
    y = d.stiffness(x, np.ones(pr.mesh.num_cells, np.float32), np.zeros(pr.ndofs, np.float32))
    assert relmax(y, pr.K(x)) < 1e-4
    d.close()
    c.close()


@pytest.mark.parametrize("P", [8, 9])
def test_linear_and_westervelt_rk4_vs_oracle(orc, P):
    L, n, nsteps = 0.012, (2, 2, 2), 8
    pr = Problem(orc, n, P, hi=[L, L, L], perturb=0.15)
    nc = pr.mesh.num_cells
    cx = pr.mesh.cell_centroids()[:, 0]
    c, rho = np.where(cx > 0.5 * L, 2800.0, 1500.0), np.where(cx > 0.5 * L, 1850.0, 1000.0)
    tags = tag_box_boundary(pr.mesh)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    tf = nsteps * dt * (1 - 1e-9)
    ctx = fa.Context(0)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, tf, dt, u, v)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < 1e-10 and relmax(vn.x.array, v) < 1e-10
    model.close()
    delta = np.full(nc, fa.compute_diffusivity_of_sound(2 * np.pi * 0.5e6, 1500.0, 0.2))
    beta = np.full(nc, 3.5)
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    n1 = -2.0 * beta / rho**2 / c**4
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, 0.5e6, 6e6, 1500.0,
                       0.0, tf, dt, u, v)
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, 0.5e6, 6e6, 1500.0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < 1e-10 and relmax(vn.x.array, v) < 1e-10
    model.close()
    ctx.close()


@pytest.mark.parametrize("P", [8, 9, 10])
@pytest.mark.parametrize("det", [0, 1])
def test_streamed_geometry_operators_vs_oracle(orc, P, det):
    """Per-point G / detJ from HBM at degrees 8-10 (the reference's data path, precompute.hpp:101-213)."""
    pr = Problem(orc, (3, 2, 2), P, hi=[1.5, 1.0, 0.8], perturb=0.2)
    c = fa.Context(0, deterministic=det, geometry="stream")
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == "stream"
    rng = np.random.default_rng(P)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    y0 = rng.standard_normal(pr.ndofs)
    assert relmax(d.stiffness(x, coef, y0.copy()), y0 + pr.K(x, coef)) < 1e-12
    assert relmax(d.mass(x, coef, y0.copy()), y0 + pr.M(x, coef)) < 1e-13
    d.close()
    c.close()


@pytest.mark.parametrize("P,dtype,tol", [(8, np.float64, 1e-12), (10, np.float64, 1e-12), (9, np.float32, 2e-4)])
def test_second_order_geometry_high_degree(orc, P, dtype, tol):
    """27-node (curved) hexahedra at degrees 8-10: geometry factors of the tri-quadratic map, streamed."""
    warp = lambda X: X + 0.03 * np.stack([np.sin(2 * X[:, 1]) * X[:, 2], np.sin(3 * X[:, 0]), X[:, 0] * X[:, 1]], axis=1)  # noqa: E731
    pr = Problem(orc, (2, 2, 2), P, order=2, warp=warp, dtype=dtype)
    c = fa.Context(0)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == "stream"
    rng = np.random.default_rng(3)
    x = rng.standard_normal(pr.ndofs).astype(dtype)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells).astype(dtype)
    assert relmax(d.stiffness(x, coef, np.zeros(pr.ndofs, dtype)), pr.K(x, coef)) < tol
    assert relmax(d.mass(x, coef, np.zeros(pr.ndofs, dtype)), pr.M(x, coef)) < tol
    d.close()
    c.close()


def test_streamed_geometry_rk4_high_degree(orc):
    """Linear RK4 through the streamed-geometry kernel at p = 8 (fused stage update included)."""
    L, n, nsteps, P = 0.012, (2, 2, 2), 6, 8
    pr = Problem(orc, n, P, hi=[L, L, L], perturb=0.15)
    nc = pr.mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    tags = tag_box_boundary(pr.mesh)
    dt = 0.5 * (L / n[0]) / (1500.0 * P**2)
    tf = nsteps * dt * (1 + 1e-12)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, tf, dt, u, v)
    ctx = fa.Context(0, geometry="stream")
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=ctx)
    assert model.data.geometry_mode() == "stream"
    model.init()
    model.rk4_steps(0.0, dt, nsteps)
    assert np.abs(u).max() > 0 and relmax(model.u_sol().x.array, u) < 1e-10
    model.close()
    ctx.close()


def test_unsupported_combinations_are_errors(orc):
    c = fa.Context(0)
    p11 = Problem(orc, (1, 1, 1), 11)
    with pytest.raises(fa.FusError):
        fa.SpectralOperatorData(p11.V, c)          # the reference's map ends at 10
    c.close()
