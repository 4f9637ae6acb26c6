"""The reference's own Python integration tests (its only CI tests, .github/workflows/python-app.yml:
python/tests/test_linearspectral_1d.py, test_lossyspectral_1d.py, test_westerveltspectral_1d.py)
run on the HIP path: same parameters, same analytical solutions, same L2 thresholds.  The reference
runs them on 1-D interval meshes; the offloaded path is hexahedral, so the interval becomes an
nx x 1 x 1 box with the source on x = 0 (tag 1), the absorbing end on x = L (tag 2) and natural
side walls, which carries the same plane wave exactly.  Lossy / Westervelt use the Python package's
boundary forms (``forms="python"``)."""
import numpy as np
import pytest
from scipy.special import jv

import fenicsxfus_amd as fa
from fenicsxfus_amd import FacetTags
from fenicsxfus_amd.utils import compute_diffusivity_of_sound
from util import Problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    yield fa.Context(0)


def interval_as_box(orc, degree, epw, f0, c0, L):
    nx = int(epw * L / (c0 / f0) + 1)                     # test_linearspectral_1d.py:29-32
    h = L / nx
    pr = Problem(orc, (nx, 1, 1), degree, hi=[L, h, h])
    cells, lf, ax, sd = pr.mesh.exterior_facets()
    keep = ax == 0
    tags = FacetTags(cells[keep], lf[keep], np.where(sd[keep] == 0, 1, 2))
    return pr, tags, h


def rel_l2(pr, u, ue):
    w = pr.M(np.ones(pr.ndofs))                           # GLL-collocated L2 norm
    return np.sqrt(w @ (u - ue) ** 2) / np.sqrt(w @ ue**2)


@pytest.mark.parametrize("degree,epw", [(3, 8), (4, 4), (5, 2), (6, 2)])
def test_linearspectral_explicit(orc, ctx, degree, epw):
    # test_linearspectral_1d.py:12-107: f0 = 10, c0 = 1, rho0 = 4, CFL 0.9, L2 error < 1e-3
    f0, c0, rho0, L = 10.0, 1.0, 4.0, 1.0
    p0 = rho0 * c0 * 1.0
    pr, tags, h = interval_as_box(orc, degree, epw, f0, c0, L)
    nc = pr.mesh.num_cells
    tend = L / c0 + 16 / f0
    dt = 0.9 * h / (c0 * degree**2)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, degree, np.full(nc, c0), np.full(nc, rho0), f0, p0, c0, 4, dt,
                                      V=pr.V, ctx=ctx)
    model.init()
    un, _, tf = model.rk(0.0, tend)
    X = pr.V.tabulate_dof_coordinates()[:, 0]
    ue = (p0 * np.exp(1j * (2 * np.pi * f0 * tf - 2 * np.pi * f0 / c0 * X))).imag   # :147-164
    assert rel_l2(pr, un.x.array, ue) < 1e-3
    model.close()


@pytest.mark.parametrize("degree,epw", [(3, 8), (4, 4), (5, 2), (6, 2)])
def test_lossyspectral_explicit(orc, ctx, degree, epw):
    # test_lossyspectral_1d.py:12-119: alpha = 5 dB/m, CFL 0.5, L2 error < 1e-2
    f0, c0, rho0, L, alphadB = 10.0, 1.0, 4.0, 1.0, 5.0
    w0 = 2 * np.pi * f0
    alphaNp = alphadB / 20 * np.log(10)
    delta0 = compute_diffusivity_of_sound(w0, c0, alphadB)
    p0 = rho0 * c0 * 1.0
    pr, tags, h = interval_as_box(orc, degree, epw, f0, c0, L)
    nc = pr.mesh.num_cells
    tend = L / c0 + 16 / f0
    dt = 0.5 * h / (c0 * degree**2)
    model = fa.LossySpectralExplicit(pr.mesh, tags, degree, np.full(nc, c0), np.full(nc, rho0), np.full(nc, delta0),
                                     f0, p0, c0, 4, dt, V=pr.V, ctx=ctx, forms="python")
    model.init()
    un, _, tf = model.rk(0.0, tend)
    X = pr.V.tabulate_dof_coordinates()[:, 0]
    ue = (p0 * np.exp(1j * (w0 * tf - w0 / c0 * X)) * np.exp(-alphaNp * X)).imag      # :83-101
    assert rel_l2(pr, un.x.array, ue) < 1e-2
    model.close()


@pytest.mark.parametrize("degree,epw", [(3, 16), (4, 8), (5, 4), (6, 2)])
def test_westerveltspectral_L2(orc, ctx, degree, epw):
    # test_westerveltspectral_1d.py:12-127: beta = 0.01, delta = 0, CFL 0.9, Fubini series, L2 < 1e-1
    f0, c0, rho0, beta0, L = 10.0, 1.0, 1.0, 0.01, 1.0
    w0 = 2 * np.pi * f0
    u0 = 1.0
    p0 = rho0 * c0 * u0
    pr, tags, h = interval_as_box(orc, degree, epw, f0, c0, L)
    nc = pr.mesh.num_cells
    tend = L / c0 + 8 / f0
    dt = 0.9 * h / (c0 * degree**2)
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, degree, np.full(nc, c0), np.full(nc, rho0), np.zeros(nc),
                                          np.full(nc, beta0), f0, p0, c0, 4, dt, V=pr.V, ctx=ctx, forms="python")
    model.init()
    un, _, tf = model.rk(0.0, tend)
    X = pr.V.tabulate_dof_coordinates()[:, 0]
    xsh = c0**2 / w0 / beta0 / u0                                                     # :85-111
    sigma = (X + 0.0000001) / xsh
    ue = np.zeros_like(X)
    for term in range(1, 50):
        ue += 2 / term / sigma * jv(term, term * sigma) * np.sin(term * w0 * (tf - X / c0))
    ue *= p0
    assert rel_l2(pr, un.x.array, ue) < 1e-1
    model.close()


def test_python_forms_vs_oracle(orc, ctx):
    """forms="python" against the oracle with the same convention (absorbing / delta-mass weights on
    tag 2 only, unscaled source), 3-D heterogeneous box, Lossy and Westervelt."""
    from fenicsxfus_amd import tag_box_boundary
    L, P, n = 0.012, 4, (6, 5, 4)
    pr = Problem(orc, n, P, hi=[L, L, L], perturb=0.1)
    nc = pr.mesh.num_cells
    cx = pr.mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * L) & (cx < 0.6 * L)
    c, rho = np.where(sel, 2800.0, 1500.0), np.where(sel, 1850.0, 1000.0)
    f0, s0 = 0.5e6, 1500.0
    w0 = 2 * np.pi * f0
    delta = np.where(sel, fa.compute_diffusivity_of_sound(w0, 2800.0, 400.0 / 20.0 * np.log(10.0)),
                     fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    beta = np.where(sel, 6.0, 3.5)
    tags = tag_box_boundary(pr.mesh)
    one = np.ones(pr.ndofs)
    m = pr.M(one, 1.0 / (rho * c * c)) + pr.facet_diag(tags, 2, delta / (rho * c**3))   # _lossy.py:107-114
    src = pr.facet_diag(tags, 1, 1.0 / rho)
    absb = pr.facet_diag(tags, 2, 1.0 / (rho * c))
    src2 = pr.facet_diag(tags, 1, delta / (rho * c * c))
    lin, att = -1.0 / rho, -delta / (rho * c * c)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 20
    tf = nsteps * dt * (1 - 1e-9)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, f0, 6e4, s0, 0.0, tf, dt, u, v,
                  source_scale=1.0)
    model = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, f0, 6e4, s0, 4, dt, V=pr.V, ctx=ctx,
                                     forms="python")
    assert rel(model.mass_vector(), m) < 1e-14
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and rel(un.x.array, u) < 1e-10 and rel(vn.x.array, v) < 1e-10
    model.close()
    # the two conventions really differ
    mc = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, f0, 6e4, s0, 4, dt, V=pr.V, ctx=ctx)
    mc.init()
    uc, _, _ = mc.rk(0.0, tf)
    assert rel(uc.x.array, u) > 1e-3
    mc.close()
    n1 = -2.0 * beta / rho**2 / c**4
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, f0, 6e6, s0,
                       0.0, tf, dt, u, v, source_scale=1.0)
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, f0, 6e6, s0, 4, dt, V=pr.V, ctx=ctx,
                                          forms="python")
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and rel(un.x.array, u) < 1e-10 and rel(vn.x.array, v) < 1e-10
    model.close()
