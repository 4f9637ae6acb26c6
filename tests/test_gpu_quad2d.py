"""GPU parity of the quadrilateral (2-D) operators and Linear model -- SURVEY 8 row a-5 and
BASELINE config 1 (rectangle, 128 x 128 quads, Q4): StiffnessSpectral2D / MassSpectral2D
(cpp/fenicsx-sf-naive/common/spectral_op.hpp:29-107, 226-359) and LinearSpectral2D
(cpp/fenicsx-sf-naive/common/Linear.hpp:52-350), through the same C ABI with tdim = 2."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import tag_box_boundary
from util import Problem

pytestmark = pytest.mark.gpu

TOL_OP = 1e-12
TOL_RK = 1e-10


@pytest.fixture(scope="module")
def ctx():
    yield fa.Context(0)


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("perturb", [0.0, 0.15])
def test_quad_operators_vs_oracle(orc, ctx, P, perturb):
    pr = Problem(orc, (9, 7), P, hi=[1.5, 1.0], perturb=perturb)
    rng = np.random.default_rng(P)
    x = rng.standard_normal(pr.ndofs)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    d = fa.SpectralOperatorData(pr.V, ctx)
    G, dJ = d.geometry()
    assert G.shape == pr.G.shape and relmax(G, pr.G) < 1e-13 and relmax(dJ, pr.detJ) < 1e-13
    y0 = rng.standard_normal(pr.ndofs)          # y is accumulated, not overwritten
    y = fa.StiffnessSpectral2D(pr.V, d)(x, coef, y0.copy())
    assert relmax(y, y0 + pr.K(x, coef)) < TOL_OP
    ym = fa.MassSpectral2D(pr.V, d)(x, coef, y0.copy())
    assert relmax(ym, y0 + pr.M(x, coef)) < 1e-14
    d.close()


@pytest.mark.parametrize("det", [0, 1])
@pytest.mark.parametrize("be,w", [(16, 1), (50, 2), (300, 4), (1000, 8)])
def test_quad_block_shapes(orc, be, w, det):
    """Block size / waves / deterministic rounds do not change the result beyond rounding; the
    deterministic mode is bitwise reproducible."""
    pr = Problem(orc, (23, 17), 4, hi=[1.5, 1.0], perturb=0.1)
    x = np.random.default_rng(1).standard_normal(pr.ndofs)
    coef = np.ones(pr.mesh.num_cells)
    c = fa.Context(0, block_elems=be, waves=w, deterministic=det)
    d = fa.SpectralOperatorData(pr.V, c)
    y = d.stiffness(x, coef, np.zeros(pr.ndofs))
    assert relmax(y, pr.K(x)) < TOL_OP
    if det:
        assert np.array_equal(y, d.stiffness(x, coef, np.zeros(pr.ndofs)))
    d.close()


def test_reference_2d_operator_recipe(orc, ctx):
    """Input recipe of the naive 2-D operator test (u = sin(x) cos(pi y) style smooth field on the
    unit square, cpp/fenicsx-sf-naive/tests/test_operators2d): energy u^T K u converges to the
    integral of |grad u|^2 and M 1 sums to the area."""
    pr = Problem(orc, (16, 16), 4)
    X = pr.V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(np.pi * X[:, 1])
    d = fa.SpectralOperatorData(pr.V, ctx)
    one = np.ones(pr.mesh.num_cells)
    y = d.stiffness(u, one, np.zeros(pr.ndofs))
    assert relmax(y, pr.K(u)) < TOL_OP
    # int |grad u|^2 = int cos^2 x cos^2 (pi y) + pi^2 sin^2 x sin^2 (pi y)
    ex = (0.5 + np.sin(2.0) / 4) * 0.5 + np.pi**2 * (0.5 - np.sin(2.0) / 4) * 0.5
    assert abs(u @ y - ex) < 1e-8
    assert abs(d.mass(np.ones(pr.ndofs), one, np.zeros(pr.ndofs)).sum() - 1.0) < 1e-13
    d.close()


@pytest.mark.parametrize("hetero,perturb", [(False, 0.0), (True, 0.1)])
def test_quad_linear_rk4_vs_oracle(orc, ctx, hetero, perturb):
    L, P, n = 0.012, 4, (12, 10)
    pr = Problem(orc, n, P, hi=[L, L], perturb=perturb)
    nc = pr.mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    if hetero:
        cx = pr.mesh.cell_centroids()[:, 0]
        sel = (cx > 0.4 * L) & (cx < 0.6 * L)
        c[sel], rho[sel] = 2800.0, 1850.0
    tags = tag_box_boundary(pr.mesh)
    f0, p0, s0 = 0.5e6, 60000.0, 1500.0
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 20
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    ns = orc.linear_rk4(2, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, s0, 0.0,
                        nsteps * dt * (1 - 1e-9), dt, u, v)
    assert ns == nsteps and np.abs(u).max() > 0
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    assert relmax(model.mass_vector(), m) < 1e-14
    model.init()
    un, vn, _ = model.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert model.nsteps == nsteps
    assert relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()


def test_config1_full_size(orc, ctx):
    """BASELINE config 1 at full size (128 x 128 quads, Q4, 263 169 DOFs): operator parity against
    the oracle, symmetry, constants in the kernel, and 5 RK4 steps against the oracle."""
    L, P, n = 0.12, 4, (128, 128)
    pr = Problem(orc, n, P, hi=[L, L])
    assert pr.ndofs == 263169 and pr.mesh.num_cells == 16384
    rng = np.random.default_rng(0)
    x, z = rng.standard_normal(pr.ndofs), rng.standard_normal(pr.ndofs)
    one = np.ones(pr.mesh.num_cells)
    d = fa.SpectralOperatorData(pr.V, ctx)
    y = d.stiffness(x, one, np.zeros(pr.ndofs))
    assert relmax(y, pr.K(x)) < TOL_OP
    assert abs(z @ y - x @ d.stiffness(z, one, np.zeros(pr.ndofs))) < 1e-11 * abs(z @ y)
    assert np.abs(d.stiffness(np.ones(pr.ndofs), one, np.zeros(pr.ndofs))).max() < 1e-11 * np.abs(y).max()
    d.close()
    c, rho = np.full(16384, 1500.0), np.full(16384, 1000.0)
    tags = tag_box_boundary(pr.mesh)
    f0, p0, s0 = 0.5e6, 60000.0, 1500.0
    dt = 0.5 * (L / 128) / (1500.0 * P**2)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(2, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, s0, 0.0, 5 * dt * (1 - 1e-9), dt, u, v)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, 5 * dt * (1 - 1e-9))
    assert model.nsteps == 5 and np.abs(u).max() > 0
    assert relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()
