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


@pytest.mark.parametrize("P", [8, 9, 10])
@pytest.mark.parametrize("perturb,det,dtype,tol", [(0.0, 0, np.float64, TOL_OP), (0.15, 1, np.float64, TOL_OP),
                                                   (0.15, 0, np.float32, 2e-4)])
def test_quad_operators_high_degree(orc, P, perturb, det, dtype, tol):
    """Degrees 8-10 on quadrilaterals (the naive reference's Qdegree map goes to 10): an element's 81-121 nodes span two
    waves, tile exchanges fenced by workgroup barriers (elem_compute2d, HI)."""
    pr = Problem(orc, (7, 5), P, hi=[1.5, 1.0], perturb=perturb, dtype=dtype)
    rng = np.random.default_rng(P)
    x = rng.standard_normal(pr.ndofs).astype(dtype)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells).astype(dtype)
    c = fa.Context(0, deterministic=det)
    d = fa.SpectralOperatorData(pr.V, c)
    y0 = rng.standard_normal(pr.ndofs).astype(dtype)
    assert relmax(fa.StiffnessSpectral2D(pr.V, d)(x, coef, y0.copy()), y0 + pr.K(x, coef)) < tol
    assert relmax(fa.MassSpectral2D(pr.V, d)(x, coef, y0.copy()), y0 + pr.M(x, coef)) < tol
    d.close()
    c.close()


def test_quad_linear_rk4_high_degree(orc):
    """LinearSpectral2D at p = 8 through the two-wave kernel with the fused stage update."""
    P, n, hi, nsteps = 8, (5, 4), [0.02, 0.016], 8
    pr = Problem(orc, n, P, hi=hi, perturb=0.1)
    nc = pr.mesh.num_cells
    c0, rho0 = np.full(nc, 1500.0), np.full(nc, 1000.0)
    tags = tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(c0, rho0, tags)
    dt = 0.4 * (hi[0] / n[0]) / (1500.0 * P**2)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(2, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    mdl = fa.LinearSpectralExplicit(pr.mesh, tags, P, c0, rho0, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V)
    mdl.init()
    mdl.rk4_steps(0.0, dt, nsteps)
    assert np.abs(u).max() > 0 and relmax(mdl.u_sol().x.array, u) < TOL_RK and relmax(mdl.v_n.x.array, v) < TOL_RK
    mdl.close()


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


def test_quad_lossy_and_westervelt_vs_oracle(orc, ctx):
    """LossySpectral2D / WesterveltSpectral2D (cpp/fenicsx-sf-naive/common/Lossy.hpp, Westervelt.hpp):
    the two-input fused pass and the nonlinear mass terms on quadrilaterals."""
    L, P, n = 0.012, 4, (10, 8)
    pr = Problem(orc, n, P, hi=[L, L], perturb=0.1)
    nc = pr.mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    cx = pr.mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * L) & (cx < 0.6 * L)
    c[sel], rho[sel] = 2800.0, 1850.0
    tags = tag_box_boundary(pr.mesh)
    f0, s0 = 0.5e6, 1500.0
    w0 = 2 * np.pi * f0
    delta = np.full(nc, fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    delta[sel] = fa.compute_diffusivity_of_sound(w0, 2800.0, 400.0 / 20.0 * np.log(10.0))
    beta = np.where(sel, 6.0, 3.5)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 20
    tf = nsteps * dt * (1 - 1e-9)
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    assert orc.lossy_rk4(2, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, f0, 6e4, s0, 0.0, tf, dt, u, v) == nsteps
    model = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, f0, 6e4, s0, 4, dt, V=pr.V, ctx=ctx)
    assert relmax(model.mass_vector(), m) < 1e-14
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()
    n1 = -2.0 * beta / rho**2 / c**4
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.westervelt_rk4(2, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, f0, 6e6, s0,
                       0.0, tf, dt, u, v)
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, f0, 6e6, s0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()


@pytest.mark.parametrize("size", [2, 3])
def test_quad_slabs_in_process(orc, size):
    """x-slab partition of a quadrilateral mesh over `size` contexts on one GPU (in-process
    transport): same state as the single-rank oracle, interface lines bit-identical."""
    L, P, n = [0.024, 0.012], 4, (9, 5)
    f0, p0, s0, nsteps = 0.5e6, 60000.0, 1500.0, 6
    pr = Problem(orc, n, P, hi=L, perturb=0.1)
    nc = pr.mesh.num_cells
    tags = tag_box_boundary(pr.mesh)
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    dt = 0.5 * (L[0] / n[0]) / (1500.0 * P**2)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(2, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, s0, 0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    ctxs = [fa.Context(0) for _ in range(size)]
    fa.Context.init_local_group(ctxs)
    models, offs = [], []
    for r in range(size):
        mesh = fa.BoxMesh([0, 0], L, n, rank=r, size=size, perturb=0.1)
        V = fa.FunctionSpace(mesh, P)
        k = mesh.num_cells
        models.append(fa.LinearSpectralExplicit(mesh, tag_box_boundary(mesh), P, np.full(k, 1500.0), np.full(k, 1000.0),
                                                f0, p0, s0, 4, dt, V=V, ctx=ctxs[r]))
        offs.append(V.global_offset)
    fa.group_finish_setup(models)
    for mdl in models:
        mdl.init()
    fa.group_rk4_steps(models, 0.0, dt, nsteps)
    us = []
    for r, mdl in enumerate(models):
        k = mdl.data.ndofs
        ur = mdl.u_sol().x.array
        us.append(ur)
        assert np.abs(ur - u[offs[r]:offs[r] + k]).max() < 1e-10 * np.abs(u).max()
    for r in range(size - 1):
        line = len(us[r]) - (offs[r + 1] - offs[r])
        assert np.array_equal(us[r][-line:], us[r + 1][:line])
    for mdl in models:
        mdl.close()
    for cx in ctxs:
        cx.close()
