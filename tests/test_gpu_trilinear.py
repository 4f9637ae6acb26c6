"""Option "geometry" = "trilinear": the block operator keeps the 21 coefficients of every cell's
trilinear map and recomputes J(q), G(q) = K K^T |det J| w_q and |det J| w_q per point in registers
(the formulas of precompute.hpp:101-213 / 33-94) instead of streaming them.  Same parity bar as the
streamed path: operators against the oracle on distorted meshes, every degree, both accumulation
modes, fp32, the three models, and agreement with the streamed path at full size."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import tag_box_boundary
from util import Problem

pytestmark = pytest.mark.gpu

TOL_OP = 1e-12     # north_star: fp64 operator action within 1e-12 of the reference
TOL_RK = 1e-10


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def ctx():
    yield fa.Context(0, geometry="trilinear")


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("det", [0, 1])
def test_operators_vs_oracle(orc, P, det):
    n = (6, 5, 4) if P <= 4 else (3, 3, 2)
    pr = Problem(orc, n, P, hi=[1.5, 1.0, 0.8], perturb=0.2)
    c = fa.Context(0, geometry="trilinear", deterministic=det)
    rng = np.random.default_rng(P)
    x = rng.standard_normal(pr.ndofs)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == "trilinear"
    y0 = rng.standard_normal(pr.ndofs)
    y = fa.StiffnessSpectral3D(pr.V, d)(x, coef, y0.copy())
    assert relmax(y, y0 + pr.K(x, coef)) < TOL_OP
    ym = fa.MassSpectral3D(pr.V, d)(x, coef, y0.copy())
    assert relmax(ym, y0 + pr.M(x, coef)) < 1e-13
    # the inspection path still returns the per-point factors of the reference layout
    G, dJ = d.geometry()
    assert relmax(G, pr.G) < 1e-12 and relmax(dJ, pr.detJ) < 1e-13
    d.close()


@pytest.mark.parametrize("P", [5, 7])
@pytest.mark.parametrize("geometry,perturb", [("trilinear", 0.2), (None, 0.0)])
def test_one_wave_workgroups_high_degree(orc, P, geometry, perturb):
    """Option waves = 1 at the higher degrees: a 64-thread workgroup has fewer threads than the prologue
    has table entries (N^2 + 2 N = 80 at degree 7), which the per-cell geometry kernels read their
    quadrature weights and points from."""
    pr = Problem(orc, (3, 2, 2), P, hi=[1.5, 1.0, 0.8], perturb=perturb)
    c = fa.Context(0, geometry=geometry, waves=1)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == ("trilinear" if perturb else "affine")
    rng = np.random.default_rng(P)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    assert relmax(d.stiffness(x, coef, np.zeros(pr.ndofs)), pr.K(x, coef)) < TOL_OP
    assert relmax(d.mass(x, coef, np.zeros(pr.ndofs)), pr.M(x, coef)) < 1e-13
    d.close()
    c.close()


def test_far_from_origin_and_small_cells(orc, ctx):
    # millimetre cells a metre away from the origin: the map coefficients are differences of vertex
    # coordinates, so the accuracy must follow the cell size, not the coordinate magnitude
    pr = Problem(orc, (5, 4, 3), 4, lo=[1.0, 2.0, -3.0], hi=[1.005, 2.004, -2.997], perturb=0.2)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(pr.ndofs)
    d = fa.SpectralOperatorData(pr.V, ctx)
    y = fa.StiffnessSpectral3D(pr.V, d)(x, np.ones(pr.mesh.num_cells), np.zeros(pr.ndofs))
    assert relmax(y, pr.K(x)) < 1e-10      # the oracle's own G carries eps * |x| / h here
    ds = fa.SpectralOperatorData(pr.V, fa.Context(0, geometry="stream"))
    ys = fa.StiffnessSpectral3D(pr.V, ds)(x, np.ones(pr.mesh.num_cells), np.zeros(pr.ndofs))
    assert relmax(y, ys) < 1e-10
    d.close(), ds.close()


def test_affine_mesh_is_not_promoted(orc, ctx):
    # "trilinear" is taken literally (no affine shortcut), e.g. to time it on the benchmark box
    pr = Problem(orc, (4, 4, 4), 4)
    d = fa.SpectralOperatorData(pr.V, ctx)
    assert d.geometry_mode() == "trilinear" and not d.is_affine()
    x = np.random.default_rng(0).standard_normal(pr.ndofs)
    assert relmax(fa.StiffnessSpectral3D(pr.V, d)(x, np.ones(pr.mesh.num_cells), np.zeros(pr.ndofs)), pr.K(x)) < TOL_OP
    d.close()


def test_second_order_geometry_falls_back_to_stream(orc, ctx):
    pr = Problem(orc, (3, 3, 2), 4, hi=[0.012, 0.012, 0.008], order=2,
                 warp=lambda x: x + np.c_[20.0 * x[:, 1] ** 2, 15.0 * x[:, 2] ** 2, 0 * x[:, 0]])
    d = fa.SpectralOperatorData(pr.V, ctx)
    assert d.geometry_mode() == "stream"
    x = np.random.default_rng(0).standard_normal(pr.ndofs)
    assert relmax(fa.StiffnessSpectral3D(pr.V, d)(x, np.ones(pr.mesh.num_cells), np.zeros(pr.ndofs)), pr.K(x)) < TOL_OP
    d.close()


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7])
def test_fp32(orc, P):
    n = (6, 5, 4) if P <= 4 else (3, 3, 2)
    pr = Problem(orc, n, P, perturb=0.2, dtype=np.float32)
    c = fa.Context(0)                      # default selection: distorted first-order cells -> trilinear
    rng = np.random.default_rng(2)
    x = rng.standard_normal(pr.ndofs).astype(np.float32)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == "trilinear"
    y = fa.StiffnessSpectral3D(pr.V, d)(x, np.ones(pr.mesh.num_cells, np.float32), np.zeros(pr.ndofs, np.float32))
    assert y.dtype == np.float32 and relmax(y, pr.K(x)) < 5e-5
    ym = fa.MassSpectral3D(pr.V, d)(x, np.ones(pr.mesh.num_cells, np.float32), np.zeros(pr.ndofs, np.float32))
    assert relmax(ym, pr.M(x)) < 1e-5
    d.close()
    c.close()


def _setup(orc, n, P, L, hetero=True):
    pr = Problem(orc, n, P, hi=[L, L, L], perturb=0.15)
    nc = pr.mesh.num_cells
    cx = pr.mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * L) & (cx < 0.6 * L) if hetero else np.zeros(nc, bool)
    return pr, np.where(sel, 2800.0, 1500.0), np.where(sel, 1850.0, 1000.0), tag_box_boundary(pr.mesh)


@pytest.mark.parametrize("P", [4, 6])
def test_three_models_vs_oracle(orc, ctx, P):
    L = 0.012
    n = (6, 6, 6) if P == 4 else (3, 3, 3)
    pr, c, rho, tags = _setup(orc, n, P, L)
    nc = pr.mesh.num_cells
    f0, s0 = 0.5e6, 1500.0
    w0 = 2 * np.pi * f0
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 20
    tf = nsteps * dt * (1 - 1e-9)
    # Linear (Linear.hpp:228-314)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, 6e4, s0, 0.0, tf, dt, u, v)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, f0, 6e4, s0, 4, dt, V=pr.V, ctx=ctx)
    assert model.data.geometry_mode() == "trilinear"
    assert relmax(model.mass_vector(), m) < 1e-13
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()
    # Lossy (two operator inputs per pass) and Westervelt (+ nonlinear mass terms)
    delta = np.where(c > 2000.0, fa.compute_diffusivity_of_sound(w0, 2800.0, 400.0 / 20.0 * np.log(10.0)),
                     fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    beta = np.where(c > 2000.0, 6.0, 3.5)
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, f0, 6e4, s0, 0.0, tf, dt, u, v)
    model = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, f0, 6e4, s0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()
    n1 = -2.0 * beta / rho**2 / c**4
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, f0, 6e6, s0,
                       0.0, tf, dt, u, v)
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, f0, 6e6, s0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < TOL_RK and relmax(vn.x.array, v) < TOL_RK
    model.close()


def test_full_size_agrees_with_streamed_path():
    """BASELINE configs[1] (64^3 p=4, 16.97 M dofs), distorted mesh: 5 RK4 steps through the trilinear
    and the streamed geometry agree to rounding."""
    from fenicsxfus_amd import BoxMesh, FunctionSpace
    mesh = BoxMesh([0, 0, 0], [0.12, 0.12, 0.12], (64, 64, 64), perturb=0.15)
    V = FunctionSpace(mesh, 4)
    tags = tag_box_boundary(mesh)
    nc = mesh.num_cells
    c0, rho0 = np.full(nc, 1500.0), np.full(nc, 1000.0)
    dt = 0.5 * (0.12 / 64) / (1500.0 * 16)
    res = {}
    for g in ("stream", "trilinear"):
        cx = fa.Context(0, geometry=g)
        model = fa.LinearSpectralExplicit(mesh, tags, 4, c0, rho0, 0.5e6, 6e4, 1500.0, 4, dt, V=V, ctx=cx)
        assert model.data.geometry_mode() == g
        model.init()
        un, vn, _ = model.rk(0.0, 5 * dt * (1 - 1e-9))
        res[g] = (un.x.array.copy(), vn.x.array.copy())
        model.close()
    assert np.abs(res["stream"][0]).max() > 0
    assert relmax(res["trilinear"][0], res["stream"][0]) < 1e-11
    assert relmax(res["trilinear"][1], res["stream"][1]) < 1e-11


@pytest.mark.parametrize("geometry", ["stream", "trilinear"])
def test_mirrored_cells_negative_jacobian(orc, geometry):
    """A reflected mesh (det J < 0 in every cell): the factors use |det J| (precompute.hpp:84, 201),
    on both geometry paths."""
    pr = Problem(orc, (5, 4, 3), 4, hi=[1.5, 1.0, 0.8], perturb=0.2)
    pr.mesh.geometry.x[:, 0] *= -1.0
    pr.G, pr.detJ = orc.geometry(3, pr.mesh.geometry.x, pr.mesh.geometry.dofmap, pr.nodes, pr.wts)
    assert (pr.detJ > 0).all()
    c = fa.Context(0, geometry=geometry)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == geometry
    rng = np.random.default_rng(3)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    assert relmax(d.stiffness(x, coef, np.zeros(pr.ndofs)), pr.K(x, coef)) < TOL_OP
    assert relmax(d.mass(x, coef, np.zeros(pr.ndofs)), pr.M(x, coef)) < 1e-13
    d.close()
    c.close()
