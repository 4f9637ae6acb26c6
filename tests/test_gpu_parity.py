"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  fp64 tolerances: 1e-12 relative per operator action,
1e-10 after 20 RK4 steps; fp32: 1e-5."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import FacetTags, tag_box_boundary
from util import Problem

pytestmark = pytest.mark.gpu

TOL_OP = 1e-12
TOL_RK = 1e-10


# every test taking `ctx` runs through the streamed per-point factors (the reference's data path) and
# through the default selection (affine cells: 7 numbers per cell; distorted first-order cells: the
# cell's trilinear map, G recomputed per point)
@pytest.fixture(scope="module", params=["stream", "auto"])
def ctx(request):
    c = fa.Context(0, geometry=request.param)
    yield c


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7])
def test_tables_and_geometry(orc, ctx, P):
    pr = Problem(orc, (3, 4, 2), P, hi=[1.5, 1.0, 0.8], perturb=0.15)
    d = fa.SpectralOperatorData(pr.V, ctx)
    w, D = d.tables()
    assert np.allclose(w, pr.wts, rtol=0, atol=1e-15) and np.allclose(D, pr.D, rtol=0, atol=1e-11)
    G, dJ = d.geometry()
    assert relmax(G, pr.G) < 1e-13 and relmax(dJ, pr.detJ) < 1e-13
    d.close()


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("perturb", [0.0, 0.15])
def test_stiffness_and_mass_vs_oracle(orc, ctx, P, perturb):
    n = (6, 5, 4) if P <= 4 else (3, 3, 2)
    pr = Problem(orc, n, P, hi=[1.5, 1.0, 0.8], perturb=perturb)
    rng = np.random.default_rng(P)
    x = rng.standard_normal(pr.ndofs)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    d = fa.SpectralOperatorData(pr.V, ctx)
    y0 = rng.standard_normal(pr.ndofs)          # y is accumulated, not overwritten
    y = fa.StiffnessSpectral3D(pr.V, d)(x, coef, y0.copy())
    ref = y0 + pr.K(x, coef)
    assert relmax(y, ref) < TOL_OP
    ym = fa.MassSpectral3D(pr.V, d)(x, coef, y0.copy())
    assert relmax(ym, y0 + pr.M(x, coef)) < 1e-14
    d.close()


@pytest.mark.parametrize("geometry", ["stream", "trilinear"])
@pytest.mark.parametrize("det", [0, 1])
@pytest.mark.parametrize("be,w", [(16, 1), (16, 2), (7, 4), (64, 4), (200, 4)])
def test_block_shapes_and_waves(orc, be, w, det, geometry):
    # ragged blocks / single-wave workgroups / one block for the whole mesh, both accumulation modes
    pr = Problem(orc, (5, 4, 3), 4, perturb=0.1)
    c = fa.Context(0, block_elems=be, waves=w, deterministic=bool(det), geometry=geometry)
    d = fa.SpectralOperatorData(pr.V, c)
    x = np.random.default_rng(0).standard_normal(pr.ndofs)
    coef = np.full(pr.mesh.num_cells, -1.0 / 3)
    y = d.stiffness(x, coef, np.zeros(pr.ndofs))
    assert relmax(y, pr.K(x, coef)) < TOL_OP
    d.close()
    c.close()


def test_reference_operator_test_recipe(orc, ctx):
    # cpp/fenicsx-sf/tests/test_operators3d/main.cpp:27-38,60-67,78-79,85,129: unit cube 20^3, P=4,
    # u = sin(x) cos(pi y), c0 = 1.5e-3, rho0 = 1e-3, mass coeff 1/(rho c^2), stiffness coeff -1/rho
    pr = Problem(orc, (20, 20, 20), 4)
    X = pr.V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(np.pi * X[:, 1])
    nc = pr.mesh.num_cells
    c0, rho0 = 1.5e-3, 1e-3
    d = fa.SpectralOperatorData(pr.V, ctx)
    m1 = d.mass(u, np.full(nc, 1 / rho0 / c0 / c0), np.zeros(pr.ndofs))
    s1 = d.stiffness(u, np.full(nc, -1 / rho0), np.zeros(pr.ndofs))
    assert relmax(m1, pr.M(u, np.full(nc, 1 / rho0 / c0 / c0))) < 1e-14
    assert relmax(s1, pr.K(u, np.full(nc, -1 / rho0), fast=True)) < TOL_OP
    d.close()


@pytest.mark.parametrize("P", [2, 4, 7])
def test_affine_geometry_path(orc, P):
    """Affine meshes (every cell a parallelepiped) take the per-cell geometry path (7 numbers per cell,
    G(q) = Gc w_q); it must agree with the streamed per-point path and with the oracle, and a
    perturbed mesh must leave it (for the cells' trilinear maps)."""
    n = (5, 4, 3) if P <= 4 else (3, 2, 2)
    pr = Problem(orc, n, P, hi=[1.5, 1.0, 0.8])
    rng = np.random.default_rng(P)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    ca, cs = fa.Context(0, geometry="auto"), fa.Context(0, geometry="stream")
    da, ds = fa.SpectralOperatorData(pr.V, ca), fa.SpectralOperatorData(pr.V, cs)
    assert da.is_affine() and not ds.is_affine()
    ref = pr.K(x, coef)
    ya = da.stiffness(x, coef, np.zeros(pr.ndofs))
    ys = ds.stiffness(x, coef, np.zeros(pr.ndofs))
    assert relmax(ya, ref) < TOL_OP and relmax(ys, ref) < TOL_OP
    assert relmax(da.mass(x, coef, np.zeros(pr.ndofs)), pr.M(x, coef)) < 1e-14
    G, dJ = da.geometry()           # per-point factors built on demand
    assert relmax(G, pr.G) < 1e-13 and relmax(dJ, pr.detJ) < 1e-13
    pp = Problem(orc, n, P, hi=[1.5, 1.0, 0.8], perturb=0.1)
    dp = fa.SpectralOperatorData(pp.V, ca)
    assert not dp.is_affine() and dp.geometry_mode() == "trilinear"
    assert relmax(dp.stiffness(x, coef, np.zeros(pr.ndofs)), pp.K(x, coef)) < TOL_OP
    for d in (da, ds, dp):
        d.close()
    ca.close(), cs.close()


def test_node_order_invariance_gpu(orc, ctx):
    P = 4
    order = np.r_[0, P, 1:P]          # endpoints first (Basix-like 1-D order)
    a = Problem(orc, (3, 3, 3), P, perturb=0.1)
    b = Problem(orc, (3, 3, 3), P, perturb=0.1, node_order=order)
    x = np.random.default_rng(1).standard_normal(a.ndofs)
    coef = np.ones(a.mesh.num_cells)
    da, db = fa.SpectralOperatorData(a.V, ctx), fa.SpectralOperatorData(b.V, ctx)
    ya = da.stiffness(x, coef, np.zeros(a.ndofs))
    yb = db.stiffness(x, coef, np.zeros(a.ndofs))
    assert relmax(yb, ya) < TOL_OP and relmax(ya, a.K(x)) < TOL_OP
    da.close(), db.close()


def test_bitwise_reproducible(orc):
    # deterministic mode: conflict-free rounds, fixed summation order
    pr = Problem(orc, (8, 8, 8), 4, perturb=0.1)
    ctx = fa.Context(0, deterministic=True)
    d = fa.SpectralOperatorData(pr.V, ctx)
    x = np.random.default_rng(2).standard_normal(pr.ndofs)
    coef = np.ones(pr.mesh.num_cells)
    y1 = d.stiffness(x, coef, np.zeros(pr.ndofs))
    y2 = d.stiffness(x, coef, np.zeros(pr.ndofs))
    assert np.array_equal(y1, y2)
    assert relmax(y1, pr.K(x, coef)) < TOL_OP
    d.close()
    ctx.close()


def test_fp32_operator(orc, ctx):
    pr = Problem(orc, (4, 4, 4), 4, perturb=0.1, dtype=np.float32)
    pr64 = Problem(orc, (4, 4, 4), 4, perturb=0.1)
    x = np.random.default_rng(3).standard_normal(pr.ndofs).astype(np.float32)
    coef = np.ones(pr.mesh.num_cells, np.float32)
    d = fa.SpectralOperatorData(pr.V, ctx)
    y = d.stiffness(x, coef, np.zeros(pr.ndofs, np.float32))
    ref = pr64.K(x.astype(np.float64))
    assert relmax(y, ref) < 1e-5
    assert relmax(y, pr.K(x, coef)) < 1e-5
    d.close()


def _linear_setup(orc, ctx, n, P, hi, perturb=0.0, c0=1500.0, rho0=1000.0, hetero=False):
    pr = Problem(orc, n, P, hi=hi, perturb=perturb)
    nc = pr.mesh.num_cells
    c = np.full(nc, c0)
    rho = np.full(nc, rho0)
    if hetero:   # cortical-bone slab, BM7-SC1/main.cpp:37-40
        cx = pr.mesh.cell_centroids()[:, 0]
        sel = (cx > 0.4 * hi[0]) & (cx < 0.6 * hi[0])
        c[sel], rho[sel] = 2800.0, 1850.0
    tags = tag_box_boundary(pr.mesh)
    return pr, c, rho, tags


@pytest.mark.parametrize("model_kind", ["linear", "westervelt"])
def test_rk4_accumulator_streams_kept_or_rebuilt(orc, model_kind):
    """Option "lean_rk4" (default 1): the classical RK4 keeps no accumulators u_, v_ in HBM -- the last stage
    builds the new state from the three stage velocities, and u0 is rebuilt from the stage input (kernels.hpp,
    stage kinds 4-7); 0 streams the accumulators at every stage like Linear.hpp:282-294.  Both against the oracle, and against each
    other to rounding, on interior, shared and boundary dofs (16 blocks)."""
    L, P, n, nsteps = 0.012, 4, (6, 6, 6), 20
    pr, c, rho, tags = _linear_setup(orc, None, n, P, [L, L, L], perturb=0.1, hetero=True)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    tf = nsteps * dt * (1 - 1e-9)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    if model_kind == "linear":
        m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
        orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, tf, dt, u, v)
    else:
        delta = np.full(pr.mesh.num_cells, fa.compute_diffusivity_of_sound(2 * np.pi * 0.5e6, 1500.0, 0.2))
        beta = np.full(pr.mesh.num_cells, 3.5)
        m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
        n1 = -2.0 * beta / rho**2 / c**4
        orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, 0.5e6, 6e6, 1500.0,
                           0.0, tf, dt, u, v)
    sols = []
    for lean in (1, 0):
        cx = fa.Context(0, block_elems=16)
        cx.set_option("lean_rk4", lean)
        if model_kind == "linear":
            model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=cx)
        else:
            model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, 0.5e6, 6e6, 1500.0, 4, dt,
                                                  V=pr.V, ctx=cx)
        model.init()
        un, vn, _ = model.rk(0.0, tf)
        assert np.abs(u).max() > 0
        assert relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
        sols.append((un.x.array.copy(), vn.x.array.copy()))
        model.close()
        cx.close()
    assert relmax(sols[0][0], sols[1][0]) < 1e-13 and relmax(sols[0][1], sols[1][1]) < 1e-13


@pytest.mark.parametrize("hetero,perturb", [(False, 0.0), (True, 0.1)])
def test_linear_rk4_vs_oracle(orc, ctx, hetero, perturb):
    L = 0.012
    P, n = 4, (6, 6, 6)
    pr, c, rho, tags = _linear_setup(orc, ctx, n, P, [L, L, L], perturb=perturb, hetero=hetero)
    f0, p0, s0 = 0.5e6, 60000.0, 1500.0
    h = L / n[0]
    dt = 0.5 * h / (c.max() * P**2)
    nsteps = 20
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    ns = orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, s0, 0.0, nsteps * dt * (1 - 1e-9),
                        dt, u, v)
    assert ns == nsteps
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    assert relmax(model.mass_vector(), m) < 1e-14
    model.init()
    un, vn, _ = model.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert model.nsteps == nsteps
    assert np.abs(u).max() > 0
    assert relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    # exact-step entry gives the same state
    model.init()
    model.rk4_steps(0.0, dt, nsteps - 1)
    model.rk4_steps((nsteps - 1) * dt, dt * (1 - 1e-9 * nsteps), 1)
    model.close()


@pytest.mark.parametrize("hetero", [False, True])
def test_lossy_rk4_vs_oracle(orc, ctx, hetero):
    # LossySpectral3D (Lossy.hpp:56-342): two operator actions per stage fused into one pass
    L = 0.012
    P, n = 4, (6, 6, 6)
    pr, c, rho, tags = _linear_setup(orc, ctx, n, P, [L, L, L], perturb=0.1, hetero=hetero)
    f0, p0, s0 = 0.5e6, 60000.0, 1500.0
    w0 = 2 * np.pi * f0
    delta = np.full(pr.mesh.num_cells, fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    if hetero:   # attenuating bone, BM7-SC1/main.cpp:43-46 (alpha = 400/20 ln 10 Np/m ... scaled)
        delta[c > 2000.0] = fa.compute_diffusivity_of_sound(w0, 2800.0, 400.0 / 20.0 * np.log(10.0))
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 20
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    ns = orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, f0, p0, s0, 0.0,
                       nsteps * dt * (1 - 1e-9), dt, u, v)
    assert ns == nsteps
    model = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    assert relmax(model.mass_vector(), m) < 1e-14
    model.init()
    un, vn, _ = model.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert np.abs(u).max() > 0
    assert relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    # attenuation does something: differs from the linear model beyond tolerance
    lm = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    lm.init()
    ul, _, _ = lm.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert relmax(un.x.array, 2 * ul.x.array) > 1e-6
    model.close(), lm.close()


def test_westervelt_rk4_vs_oracle(orc, ctx):
    # WesterveltSpectral3D (Westervelt.hpp:58-373): lossy + per-stage nonlinear mass terms
    L = 0.012
    P, n = 4, (6, 6, 6)
    pr, c, rho, tags = _linear_setup(orc, ctx, n, P, [L, L, L], perturb=0.1, hetero=True)
    f0, p0, s0 = 0.5e6, 6.0e6, 1500.0     # high drive so the nonlinear terms matter in 20 steps
    w0 = 2 * np.pi * f0
    nc = pr.mesh.num_cells
    delta = np.full(nc, fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    beta = np.where(c > 2000.0, 6.0, 3.5)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 20
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    n1 = -2.0 * beta / rho**2 / c**4
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    ns = orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, f0, p0, s0,
                            0.0, nsteps * dt * (1 - 1e-9), dt, u, v)
    assert ns == nsteps
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    # the nonlinearity is visible: differs from the lossy solution beyond tolerance
    lm = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    lm.init()
    ul, _, _ = lm.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert relmax(un.x.array, ul.x.array) > 1e-7
    model.close(), lm.close()


@pytest.mark.parametrize("P", [5, 6, 7])
def test_westervelt_higher_degrees_vs_oracle(orc, ctx, P):
    """Westervelt (two operator inputs + nonlinear mass terms) through the single-register-set block
    kernel of the higher degrees (fp64, streamed geometry)."""
    L, n = 0.012, (3, 3, 2)
    pr, c, rho, tags = _linear_setup(orc, ctx, n, P, [L, L, L], perturb=0.1, hetero=True)
    f0, p0, s0 = 0.5e6, 6.0e6, 1500.0
    w0 = 2 * np.pi * f0
    nc = pr.mesh.num_cells
    delta = np.full(nc, fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    beta = np.where(c > 2000.0, 6.0, 3.5)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 10
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    n1 = -2.0 * beta / rho**2 / c**4
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, f0, p0, s0,
                       0.0, nsteps * dt * (1 - 1e-9), dt, u, v)
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    assert not model.data.is_affine()
    model.init()
    un, vn, _ = model.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()


@pytest.mark.parametrize("order", [1, 2, 3])
def test_lower_rk_orders_vs_oracle(orc, ctx, order):
    # rk_order 1-3 of LinearSpectralExplicit (_linear.py:286-311): forward Euler, Ralston 2 / 3
    L = 0.012
    P, n = 3, (5, 4, 4)
    pr, c, rho, tags = _linear_setup(orc, ctx, n, P, [L, L, L], perturb=0.1, hetero=True)
    f0, p0, s0 = 0.5e6, 60000.0, 1500.0
    dt = 0.1 * (L / n[0]) / (c.max() * P**2)
    nsteps = 12
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, s0, 0.0, nsteps * dt * (1 - 1e-9), dt, u, v,
                   order=order)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, f0, p0, s0, order, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert model.nsteps == nsteps and np.abs(u).max() > 0
    assert relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()


def test_plane_wave_vs_analytical_gpu(orc, ctx):
    # python/tests/test_linearspectral_1d.py:12-107 (degree 4, epw 4): L2 error < 1e-3
    f0, c0, rho0, L, degree, epw = 10.0, 1.0, 4.0, 1.0, 4, 4
    p0 = rho0 * c0
    nx = int(epw * L / (c0 / f0) + 1)
    h = L / nx
    pr = Problem(orc, (nx, 1, 1), degree, hi=[L, h, h])
    cells, lf, ax, sd = pr.mesh.exterior_facets()
    keep = ax == 0
    tags = FacetTags(cells[keep], lf[keep], np.where(sd[keep] == 0, 1, 2))
    tend = L / c0 + 16 / f0
    dt = 0.9 * h / (c0 * degree**2)
    nc = pr.mesh.num_cells
    model = fa.LinearSpectralExplicit(pr.mesh, tags, degree, np.full(nc, c0), np.full(nc, rho0), f0, p0, c0, 4, dt,
                                      V=pr.V, ctx=ctx)
    model.init()
    un, _, _ = model.rk(0.0, tend)
    X = pr.V.tabulate_dof_coordinates()[:, 0]
    ue = p0 * np.sin(2 * np.pi * f0 * (tend - X / c0)) * (tend - X / c0 > 0)
    w = pr.M(np.ones(pr.ndofs))
    err = np.sqrt(w @ (un.x.array - ue) ** 2) / np.sqrt(w @ ue**2)
    assert err < 1e-3, err
    model.close()


def test_full_size_properties(orc, ctx):
    """BASELINE config 2 (64^3 hex, p=4, fp64): size-independent properties of the operator --
    K 1 = 0, sum(K x) = 0, symmetry, sum(m) = volume/(rho c^2) -- at the full benchmark size."""
    L = 0.12
    m = fa.BoxMesh([0, 0, 0], [L, L, L], (64, 64, 64))
    V = fa.FunctionSpace(m, 4)
    d = fa.SpectralOperatorData(V, ctx)
    n, nc = V.num_dofs, m.num_cells
    assert n == 16974593
    rng = np.random.default_rng(0)
    x, z = rng.standard_normal(n), rng.standard_normal(n)
    coef = np.full(nc, -1e-3)
    y = d.stiffness(x, coef, np.zeros(n))
    scale = np.abs(y).max()
    assert np.abs(d.stiffness(np.ones(n), coef, np.zeros(n))).max() < 1e-11 * scale
    assert abs(y.sum()) < 1e-9 * np.abs(y).sum()
    yz = d.stiffness(z, coef, np.zeros(n))
    assert abs(z @ y - x @ yz) < 1e-10 * abs(z @ y)
    mm = d.mass(np.ones(n), np.full(nc, 1.0), np.zeros(n))
    assert abs(mm.sum() - L**3) < 1e-12 * L**3
    info = d.info()
    if d.geometry_mode() == "affine":      # affine default: 16-element blocks
        assert info["nblocks"] == 262144 // 32 and info["shapes"] == 27
    else:                                  # streamed factors: 32-element blocks
        assert d.geometry_mode() == "stream" and info["nblocks"] == 262144 // 32
    d.close()


def test_full_size_rk4_three_kernel_variants_agree():
    """BASELINE config 2 after 20 RK4 steps: the streamed-geometry kernel (the reference's data path),
    the affine-geometry kernel and the deterministic (conflict-free rounds) kernel are three different
    code paths over different block layouts; their states agree to rounding, the deterministic one is
    bitwise reproducible, and the wave has left the source face."""
    L, P, nsteps = 0.12, 4, 20
    m = fa.BoxMesh([0, 0, 0], [L, L, L], (64, 64, 64))
    V = fa.FunctionSpace(m, P)
    tags = tag_box_boundary(m)
    nc = m.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    dt = 0.5 * (L / 64) / (1500.0 * P**2)
    sols = {}
    for name, kw in (("stream", dict(geometry="stream")), ("affine", dict(geometry="auto")),
                     ("rounds", dict(geometry="stream", deterministic=1)), ("rounds2", dict(geometry="stream", deterministic=1))):
        cx = fa.Context(0, **kw)
        mdl = fa.LinearSpectralExplicit(m, tags, P, c, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=V, ctx=cx)
        assert mdl.data.is_affine() == (name == "affine")
        mdl.init()
        mdl.rk4_steps(0.0, dt, nsteps)
        sols[name] = (mdl.u_sol().x.array.copy(), mdl.v_n.x.array.copy())
        mdl.close()
        cx.close()
    u, v = sols["stream"]
    assert np.isfinite(u).all() and np.abs(u).max() > 0
    for other in ("affine", "rounds"):
        assert relmax(sols[other][0], u) < 1e-11 and relmax(sols[other][1], v) < 1e-11
    assert np.array_equal(sols["rounds"][0], sols["rounds2"][0]) and np.array_equal(sols["rounds"][1], sols["rounds2"][1])
    X0 = np.repeat(np.linspace(0, L, 64 * P + 1), (64 * P + 1) ** 2)       # x-slowest dof order
    assert np.abs(u[X0 > 0.5 * L]).max() < 1e-6 * np.abs(u).max()          # nothing has reached mid-box yet


def test_edge_cases_and_errors(orc, ctx):
    """Single-cell mesh, no boundary facets, bad arguments -> error codes (no exceptions across the ABI)."""
    import ctypes as C

    from fenicsxfus_amd import _abi

    pr = Problem(orc, (1, 1, 1), 3)
    d = fa.SpectralOperatorData(pr.V, ctx)
    x = np.random.default_rng(0).standard_normal(pr.ndofs)
    assert relmax(d.stiffness(x, np.ones(1), np.zeros(pr.ndofs)), pr.K(x)) < TOL_OP
    # model without any boundary facet (pure Neumann box): runs, stays zero from zero data
    empty = fa.FacetTags(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    mdl = fa.LinearSpectralExplicit(pr.mesh, empty, 3, np.ones(1), np.ones(1), 1.0, 1.0, 1.0, 4, 1e-3, V=pr.V, ctx=ctx)
    with pytest.raises(fa.FusError, match="init"):
        mdl.rk(0.0, 1e-2)                      # rk before init -> FUS_ERR_STATE
    mdl.init()
    u, v, _ = mdl.rk(0.0, 5e-3)
    assert mdl.nsteps == 5 and not u.x.array.any()
    mdl.close()
    L = _abi.lib()
    op = C.c_void_p()
    dm = np.ascontiguousarray(pr.V.tensor_dofmap)
    xg = np.ascontiguousarray(pr.mesh.geometry.x)
    gd = np.ascontiguousarray(pr.mesh.geometry.dofmap)
    nodes = np.ascontiguousarray(pr.V.nodes1d)
    args = lambda P, dt, order, nd: (ctx.h, 3, P, dt, C.c_int64(1), C.c_int64(pr.ndofs), _abi.ptr(dm), _abi.ptr(nd),  # noqa
                                     _abi.ptr(xg), C.c_int64(len(xg)), _abi.ptr(gd), order, C.byref(op))
    assert L.fus_op_create(*args(11, 1, 1, nodes)) == -1 and b"degree" in L.fus_last_error()   # the map ends at 10
    assert L.fus_op_create(*args(3, 7, 1, nodes)) == -1
    assert L.fus_op_create(*args(3, 1, 2, nodes)) == -1 and b"geometry" in L.fus_last_error()
    bad = nodes.copy()
    bad[1] += 0.01
    assert L.fus_op_create(*args(3, 1, 1, bad)) == -1 and b"GLL" in L.fus_last_error()
    with pytest.raises(fa.FusError, match="fields"):      # lossy model on single-field operator data
        check = _abi.check
        m = C.c_void_p()
        one = np.ones(1)
        check(L.fus_model_create(ctx.h, 1, d.h, _abi.ptr(one), _abi.ptr(one), _abi.ptr(one), None, C.c_int64(0), None,
                                 None, None, C.c_double(1), C.c_double(1), C.c_double(1), C.byref(m)))
    d.close()


def test_second_order_geometry_gpu(orc, ctx):
    """Curved 27-node hexahedra: geometry factors, operators and a short RK4 run against the oracle."""
    from test_oracle_operators import _bend

    pr = Problem(orc, (4, 3, 3), 4, hi=[0.016, 0.012, 0.012], order=2,
                 warp=lambda x: x + np.c_[20.0 * x[:, 1] ** 2 - 12.0 * x[:, 2] ** 2, 15.0 * x[:, 2] ** 2, 0 * x[:, 0]])
    d = fa.SpectralOperatorData(pr.V, ctx)
    assert not d.is_affine()
    G, dJ = d.geometry()
    assert relmax(G, pr.G) < 1e-13 and relmax(dJ, pr.detJ) < 1e-13
    rng = np.random.default_rng(5)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    assert relmax(d.stiffness(x, coef, np.zeros(pr.ndofs)), pr.K(x, coef)) < TOL_OP
    assert relmax(d.mass(x, coef, np.zeros(pr.ndofs)), pr.M(x, coef)) < 1e-14
    d.close()
    nc = pr.mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    tags = tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    dt = 0.3 * 0.004 / (1500.0 * 16)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, 10 * dt * (1 - 1e-9), dt, u, v)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, 4, c, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=ctx)
    assert relmax(model.mass_vector(), m) < 1e-14
    model.init()
    un, vn, _ = model.rk(0.0, 10 * dt * (1 - 1e-9))
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()
    assert _bend is not None


class _Space:
    """Minimal duck-typed function space built from raw arrays (what a DOLFINx-side caller hands over)."""

    def __init__(self, mesh, P, tensor_dofmap, nodes1d, ndofs):
        from fenicsxfus_amd.mesh import _DofMap, _IndexMap

        self.mesh, self.P, self.tensor_dofmap, self.nodes1d = mesh, P, tensor_dofmap, nodes1d
        self.dofmap = _DofMap(tensor_dofmap, _IndexMap(ndofs))
        self.num_dofs = ndofs


def test_unstructured_numbering_and_holes(orc, ctx):
    """No hidden structured-mesh assumption: a perturbed box with 30 % of its cells removed (holes,
    ragged boundary), the remaining cells shuffled, the vertices and the DOFs renumbered at random.
    Operators and a Linear RK4 run must match the oracle fed with the same scrambled arrays."""
    from fenicsxfus_amd.mesh import _Geometry, _Topology

    P = 3
    base = Problem(orc, (6, 5, 4), P, hi=[0.018, 0.015, 0.012], perturb=0.15)
    rng = np.random.default_rng(42)
    keep = np.sort(rng.permutation(base.mesh.num_cells)[: int(0.7 * base.mesh.num_cells)])
    keep = rng.permutation(keep)                                   # shuffled cell order
    dm = base.dm[keep]
    used = np.unique(dm)
    newid = np.full(base.ndofs, -1, dtype=np.int64)
    newid[used] = rng.permutation(len(used))                       # random DOF numbering
    dm = newid[dm].astype(np.int32)
    ndofs = len(used)
    gdm = base.mesh.geometry.dofmap[keep]
    vused = np.unique(gdm)
    vnew = np.full(len(base.mesh.geometry.x), -1, dtype=np.int64)
    vnew[vused] = rng.permutation(len(vused))                      # random vertex numbering
    xg = np.zeros((len(vused), 3))
    xg[vnew[vused]] = base.mesh.geometry.x[vused]
    gdm = vnew[gdm].astype(np.int32)

    class M:
        pass

    mesh = M()
    mesh.geometry = _Geometry(xg, gdm, 3)
    mesh.topology = _Topology(3, len(keep), len(keep))
    V = _Space(mesh, P, dm, base.nodes, ndofs)
    G, detJ = orc.geometry(3, xg, gdm, base.nodes, base.wts)
    nc = len(keep)
    x, coef = rng.standard_normal(ndofs), rng.uniform(0.5, 2.0, nc)
    for be, w in ((32, 4), (10, 2)):
        c = fa.Context(0, block_elems=be, waves=w)
        d = fa.SpectralOperatorData(V, c)
        ref = orc.stiffness(3, P + 1, dm, G, base.D, coef, x, np.zeros(ndofs))
        assert relmax(d.stiffness(x, coef, np.zeros(ndofs)), ref) < TOL_OP
        refm = orc.mass(3, P + 1, dm, detJ, coef, x, np.zeros(ndofs))
        assert relmax(d.mass(x, coef, np.zeros(ndofs)), refm) < 1e-14
        d.close(), c.close()
    # model: source on the cells' x-low facets that lie on the original x = 0 plane, absorbing nowhere
    cx = xg[gdm].mean(axis=1)[:, 0]
    src_cells = np.nonzero(cx < 0.018 / 6)[0].astype(np.int32)
    tags = fa.FacetTags(src_cells, np.full(len(src_cells), 2, np.int32), np.ones(len(src_cells), np.int32))
    cc, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    m = orc.mass(3, P + 1, dm, detJ, 1.0 / (rho * cc * cc), np.ones(ndofs), np.zeros(ndofs))
    src = orc.facet_diag(3, tags.cells, tags.local_facets, 1.0 / rho, xg, gdm, base.nodes, base.wts, dm, ndofs)
    dt = 0.3 * 0.003 / (1500.0 * P**2)
    u, v = np.zeros(ndofs), np.zeros(ndofs)
    orc.linear_rk4(3, P + 1, dm, G, base.D, -1.0 / rho, m, src, np.zeros(ndofs), 0.5e6, 6e4, 1500.0, 0.0,
                   12 * dt * (1 - 1e-9), dt, u, v)
    model = fa.LinearSpectralExplicit(mesh, tags, P, cc, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, 12 * dt * (1 - 1e-9))
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()


def test_mesh_size_norm_and_allreduce(orc):
    """What the examples' mains compute around the model (linear_planewave2d_1/main.cpp:60-68, 102,
    151-157): smallest cell size (cell size = largest vertex distance), its global minimum, the time
    step from it, and the L2 norm of the solution."""
    L, n, P = [1.5, 1.0, 0.8], (6, 5, 4), 4
    pr = Problem(orc, n, P, hi=L)
    c = fa.Context(0)
    d = fa.SpectralOperatorData(pr.V, c)
    hx = [L[i] / n[i] for i in range(3)]
    assert abs(d.hmin() - np.sqrt(sum(h * h for h in hx))) < 1e-14
    assert c.allreduce([d.hmin()], "min")[0] == d.hmin()          # one rank: identity
    x = np.random.default_rng(0).standard_normal(pr.ndofs)
    ref = x @ pr.M(x)
    assert abs(d.norm2(x) - ref) < 1e-13 * ref
    d.close()
    # distorted and second-order cells: the vertices of each cell decide
    pp = Problem(orc, n, P, hi=L, perturb=0.2)
    dp = fa.SpectralOperatorData(pp.V, c)
    X, dm = pp.mesh.geometry.x, pp.mesh.geometry.dofmap
    h = min(max(np.linalg.norm(X[a] - X[b]) for a in cell for b in cell) for cell in dm)
    assert abs(dp.hmin() - h) < 1e-14
    assert abs(dp.norm2(x) - x @ pp.M(x)) < 1e-13 * ref
    dp.close()
    p2 = Problem(orc, (3, 3, 2), 4, hi=[0.012, 0.012, 0.008], order=2,
                 warp=lambda y: y + np.c_[20.0 * y[:, 1] ** 2, 15.0 * y[:, 2] ** 2, 0 * y[:, 0]])
    d2 = fa.SpectralOperatorData(p2.V, c)
    X, dm = p2.mesh.geometry.x, p2.mesh.geometry.dofmap
    corners = [0, 2, 6, 8, 18, 20, 24, 26]
    h = min(max(np.linalg.norm(X[cell[a]] - X[cell[b]]) for a in corners for b in corners) for cell in dm)
    assert abs(d2.hmin() - h) < 1e-14
    d2.close()
    c.close()


@pytest.mark.parametrize("tdim", [2, 3])
def test_graph_replay_of_the_rk_step(orc, tdim):
    """Option "graph": the RK step captured and replayed as one hipGraph (updated in place with each
    step's stage scalars) gives the bits of the directly launched step (deterministic accumulation),
    also across a re-initialisation and a change of dt."""
    L = 0.012
    n = (6, 5, 4)[:tdim] if tdim == 3 else (9, 7)
    pr = Problem(orc, n, 4, hi=[L] * tdim, perturb=0.1)
    nc = pr.mesh.num_cells
    tags = tag_box_boundary(pr.mesh)
    c0, rho0 = np.full(nc, 1500.0), np.full(nc, 1000.0)
    dt = 0.5 * (L / n[0]) / (1500.0 * 16)
    res = []
    for graph in (0, 1):
        c = fa.Context(0, deterministic=1)
        c.set_option("graph", graph)
        m = fa.LinearSpectralExplicit(pr.mesh, tags, 4, c0, rho0, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=c)
        m.init()
        m.rk4_steps(0.0, dt, 25)
        a = m.u_sol().x.array.copy()
        un, vn, _ = m.rk(25 * dt, 25 * dt + 7.5 * dt)     # last step shorter: one more launch in that step
        b = un.x.array.copy()
        m.init()                                          # back to zero state: first step is launched directly again
        m.rk4_steps(0.0, dt, 3)
        res.append((a, b, m.u_sol().x.array.copy()))
        m.close()
        c.close()
    for x, y in zip(*res):
        assert np.abs(x).max() > 0 and np.array_equal(x, y)
    # and against the oracle
    mv, src, absb, coeff = pr.linear_model_vectors(c0, rho0, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    # 25 full steps (+ a ~1e-12 dt remainder step, far below the tolerance)
    orc.linear_rk4(tdim, pr.N, pr.dm, pr.G, pr.D, coeff, mv, src, absb, 0.5e6, 6e4, 1500.0, 0.0, 25 * dt * (1 + 1e-12), dt, u, v)
    assert relmax(res[1][0], u) < TOL_RK


@pytest.mark.parametrize("geometry,perturb", [("trilinear", 0.15), ("auto", 0.0), ("stream", 0.15)])
@pytest.mark.parametrize("walk", [1, 2])
def test_walking_workgroups(orc, geometry, perturb, walk):
    """Option "walk": `walk` workgroups per CU, each walking every (walk * CUs)-th block with the next block's
    prologue loads in flight under the current block's epilogue.  A mesh with more blocks than workgroups
    (4-element blocks) so that workgroups really walk 2-4 blocks, of different shapes: operator actions and
    the RK4 loop against the oracle, and bit-identical to one workgroup per block in deterministic mode."""
    P, n, L = 4, (16, 12, 12), 0.012
    pr, c, rho, tags = _linear_setup(orc, None, n, P, [L * 4 / 3, L, L], perturb=perturb, hetero=True)
    rng = np.random.default_rng(5)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    refK, refM = pr.K(x, coef), pr.M(x, coef)
    outs = {}
    for w in (walk, 0):
        cx = fa.Context(0, geometry=geometry, block_elems=4, deterministic=1)
        cx.set_option("walk", w)
        d = fa.SpectralOperatorData(pr.V, cx)
        assert d.info()["nblocks"] == 576
        y = d.stiffness(x, coef, np.zeros(pr.ndofs))
        ym = d.mass(x, coef, np.zeros(pr.ndofs))
        assert relmax(y, refK) < TOL_OP and relmax(ym, refM) < 1e-13
        outs[w] = (y, ym)
        d.close()
        cx.close()
    if geometry != "stream":      # (the streamed-geometry kernel keeps one workgroup per block)
        assert np.array_equal(outs[walk][0], outs[0][0]) and np.array_equal(outs[walk][1], outs[0][1])
    dt = 0.5 * (L / 12) / (c.max() * P**2)
    nsteps = 6
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, nsteps * dt * (1 - 1e-9), dt, u, v)
    cx = fa.Context(0, geometry=geometry, block_elems=4)
    cx.set_option("walk", walk)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=cx)
    model.init()
    un, vn, _ = model.rk(0.0, nsteps * dt * (1 - 1e-9))
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()
    cx.close()


@pytest.mark.parametrize("block_elems,P", [(1, 2), (4, 3), (None, 4)])
@pytest.mark.parametrize("kind", ["linear", "lossy"])
def test_shared_stage_planes_and_csr_agree_bitwise(orc, block_elems, P, kind):
    """The shared-dof stage kernel in its two forms (option "planes"): partial sums read as planes at the dof's
    own index, or through the shared-dof CSR.  Same addends in the same order (ascending block, boundary term
    last): the deterministic kernels must give identical bits, and both must match the oracle.  One-element
    blocks: every dof off a cell interior is shared, vertices by 8 blocks (8 planes)."""
    n, L = (6, 5, 4), 0.012
    pr, c, rho, tags = _linear_setup(orc, None, n, P, [L * 1.5, L * 1.25, L], perturb=0.15, hetero=True)
    dt = 0.5 * (L / 4) / (c.max() * P**2)
    nsteps = 5
    tf = nsteps * dt * (1 - 1e-9)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    if kind == "linear":
        m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
        orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, tf, dt, u, v)
    else:
        delta = np.where(c > 2000.0, fa.compute_diffusivity_of_sound(2 * np.pi * 0.5e6, 2800.0, 46.0),
                         fa.compute_diffusivity_of_sound(2 * np.pi * 0.5e6, 1500.0, 0.2))
        m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
        orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, 0.5e6, 6e4, 1500.0, 0.0, tf, dt, u, v)
    assert np.abs(u).max() > 0
    outs = {}
    # (4: a plane limit below the 8 sharers of a vertex of one-element blocks -> the CSR form is taken by itself)
    for planes in (1, 0, 4):
        kw = {} if block_elems is None else {"block_elems": block_elems}
        cx = fa.Context(0, deterministic=1, **kw)
        cx.set_option("planes", planes)
        if kind == "linear":
            model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=cx)
        else:
            model = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=cx)
        model.init()
        un, vn, _ = model.rk(0.0, tf)
        outs[planes] = (un.x.array.copy(), vn.x.array.copy())
        assert relmax(outs[planes][0], u) < TOL_RK and relmax(outs[planes][1], v) < TOL_RK
        model.close()
        cx.close()
    assert np.array_equal(outs[1][0], outs[0][0]) and np.array_equal(outs[1][1], outs[0][1])
    assert np.array_equal(outs[4][0], outs[0][0]) and np.array_equal(outs[4][1], outs[0][1])
