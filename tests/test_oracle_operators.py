"""Analytic known-answer tests that pin the operator-level oracle (CPU only).

The reference stores no golden vectors for this level (its C++ tests print a norm and never
assert -- cpp/fenicsx-sf/tests/test_operators3d/main.cpp:117-122), so these KATs + the independent
dense-table evaluation are what stands behind the oracle."""
import numpy as np
import pytest

from util import Problem
from fenicsxfus_amd import tag_box_boundary, FacetTags

TOL = 1e-12


@pytest.mark.parametrize("P", [2, 3, 4, 5, 7])
def test_gll_tables(orc, P):
    N = P + 1
    pts, wts = orc.gll(N)
    assert abs(wts.sum() - 1) < 1e-14 and pts[0] == 0 and pts[-1] == 1
    # exact for polynomials up to degree 2N-3 on [0,1]
    for k in range(2 * N - 2):
        assert abs(np.dot(wts, pts**k) - 1.0 / (k + 1)) < 1e-13
    D = orc.dphi(pts)
    assert np.allclose(D @ np.ones(N), 0, atol=1e-12)
    assert np.allclose(D @ pts**2, 2 * pts, atol=1e-11)
    # any node order gives the same weights / consistent D
    perm = np.r_[0, N - 1, 1:N - 1]
    assert np.allclose(orc.gll_weights_at(pts[perm]), wts[perm], atol=1e-15)
    assert np.allclose(orc.dphi(pts[perm]), D[np.ix_(perm, perm)], atol=1e-11)


@pytest.mark.parametrize("P,perturb", [(2, 0.0), (4, 0.0), (4, 0.15), (5, 0.1)])
def test_stiffness_invariants_3d(orc, P, perturb):
    pr = Problem(orc, (3, 2, 2), P, hi=[1.5, 1.0, 0.8], perturb=perturb)
    rng = np.random.default_rng(0)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    x, z = rng.standard_normal(pr.ndofs), rng.standard_normal(pr.ndofs)
    y = pr.K(x, coef)
    scale = np.abs(y).max()
    assert np.abs(pr.K(np.ones(pr.ndofs), coef)).max() < TOL * scale          # K 1 = 0
    assert abs(y.sum()) < 1e-11 * scale                                       # sum(K x) = 0
    assert abs(z @ y - x @ pr.K(z, coef)) < 1e-11 * abs(z @ y)                # symmetry
    yd = pr.K(x, coef, dense=True)                                            # independent O(N^6)
    assert np.abs(y - yd).max() < 1e-12 * scale


@pytest.mark.parametrize("P", [2, 4, 6])
def test_linear_field_affine(orc, P):
    # nodal interpolant of x0 on affine cells: grad is constant, GLL exact: x'Kx = c*Volume and
    # (Kx)_i = 0 away from the faces x0 = lo, hi
    L = [1.5, 1.0, 0.8]
    pr = Problem(orc, (3, 2, 2), P, hi=L)
    X = pr.V.tabulate_dof_coordinates()
    x = X[:, 0].copy()
    c = 2.5
    y = pr.K(x, np.full(pr.mesh.num_cells, c))
    assert abs(x @ y - c * np.prod(L)) < 1e-12 * c * np.prod(L)
    inner = (X[:, 0] > 1e-9) & (X[:, 0] < L[0] - 1e-9)
    assert np.abs(y[inner]).max() < 1e-12


def test_mass_3d(orc):
    L = [1.5, 1.0, 0.8]
    pr = Problem(orc, (3, 2, 2), 4, hi=L, perturb=0.1)
    rho, c = 1000.0, 1500.0
    m = pr.M(np.ones(pr.ndofs), np.full(pr.mesh.num_cells, 1 / (rho * c * c)))
    assert abs(m.sum() - np.prod(L) / (rho * c * c)) < 1e-12 * m.sum()
    # affine: diagonal entries are tensor products h^3 w_a w_b w_c summed over sharing cells
    pa = Problem(orc, (2, 2, 2), 3)
    ma = pa.M(np.ones(pa.ndofs))
    w = orc.gll(4)[1] * 0.5
    w1 = np.zeros(7)
    w1[0:4] += w
    w1[3:7] += w
    assert np.allclose(ma.reshape(7, 7, 7), np.einsum("i,j,k", w1, w1, w1), atol=1e-15)


def test_node_order_invariance(orc):
    # endpoints-first 1-D order (Basix-like) gives the same global vectors
    P = 4
    order = np.r_[0, P, 1:P]
    a = Problem(orc, (2, 2, 3), P, perturb=0.1)
    b = Problem(orc, (2, 2, 3), P, perturb=0.1, node_order=order)
    x = np.random.default_rng(1).standard_normal(a.ndofs)
    ya, yb = a.K(x), b.K(x)
    assert np.abs(ya - yb).max() < 1e-12 * np.abs(ya).max()
    assert np.abs(a.M(x) - b.M(x)).max() < 1e-15


@pytest.mark.parametrize("perturb", [0.0, 0.15])
def test_stiffness_2d(orc, perturb):
    L = [1.5, 1.0]
    pr = Problem(orc, (4, 3), 4, hi=L, perturb=perturb)
    rng = np.random.default_rng(0)
    x, z = rng.standard_normal(pr.ndofs), rng.standard_normal(pr.ndofs)
    y = pr.K(x)
    assert np.abs(pr.K(np.ones(pr.ndofs))).max() < 1e-12 * np.abs(y).max()
    assert abs(z @ y - x @ pr.K(z)) < 1e-11 * abs(z @ y)
    X = pr.V.tabulate_dof_coordinates()
    for d in (0, 1):   # linear fields: energy = area on any (also non-affine: bilinear map keeps
        if perturb:    # x_d in the space only for affine cells)
            continue
        e = X[:, d] @ pr.K(X[:, d].copy())
        assert abs(e - np.prod(L)) < 1e-12
    m = pr.M(np.ones(pr.ndofs))
    assert abs(m.sum() - np.prod(L)) < 1e-13


def test_facet_diag(orc):
    L = [1.5, 1.0, 0.8]
    pr = Problem(orc, (3, 2, 2), 4, hi=L, perturb=0.1)   # boundary vertices stay put
    tags = tag_box_boundary(pr.mesh)
    rho = 2.0
    src = pr.facet_diag(tags, 1, np.full(pr.mesh.num_cells, 1 / rho))
    ab = pr.facet_diag(tags, 2, np.ones(pr.mesh.num_cells))
    assert abs(src.sum() - L[1] * L[2] / rho) < 1e-13
    total = 2 * (L[0] * L[1] + L[1] * L[2] + L[0] * L[2])
    assert abs(ab.sum() - (total - L[1] * L[2])) < 1e-12
    X = pr.V.tabulate_dof_coordinates()
    assert np.all(src[X[:, 0] > 1e-12] == 0)


@pytest.mark.parametrize("degree,epw", [(4, 4), (5, 2)])
def test_plane_wave_vs_analytical(orc, degree, epw):
    """The reference's 1-D analytical check (python/tests/test_linearspectral_1d.py:12-107:
    f0=10, c0=1, rho0=4, L=1, CFL=0.9, L2 error < 1e-3) run as an nx x 1 x 1 box with the source on
    x=0, absorbing x=L and natural side walls (exact plane wave)."""
    f0, c0, rho0, L = 10.0, 1.0, 4.0, 1.0
    p0 = rho0 * c0 * 1.0
    nx = int(epw * L / (c0 / f0) + 1)
    h = L / nx
    pr = Problem(orc, (nx, 1, 1), degree, hi=[L, h, h])
    cells, lf, ax, sd = pr.mesh.exterior_facets()
    keep = ax == 0
    tags = FacetTags(cells[keep], lf[keep], np.where(sd[keep] == 0, 1, 2))
    m, src, absb, coeff = pr.linear_model_vectors(c0, rho0, tags)
    tend = L / c0 + 16 / f0
    dt = 0.9 * h / (c0 * degree**2)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    nsteps = orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, c0, 0.0, tend, dt, u, v)
    assert nsteps == int(np.ceil(tend / dt - 1e-9))
    X = pr.V.tabulate_dof_coordinates()[:, 0]
    ue = p0 * np.sin(2 * np.pi * f0 * (tend - X / c0)) * (tend - X / c0 > 0)
    w = pr.M(np.ones(pr.ndofs))
    err = np.sqrt(w @ (u - ue) ** 2) / np.sqrt(w @ ue**2)
    assert err < 1e-3, err


def test_lossy_oracle_reduces_to_linear(orc):
    """delta = 0: Lossy.hpp's f1 is Linear.hpp's with the source doubled (heterogeneous scaling,
    Lossy.hpp:216-220 vs Linear.hpp:192) and the absorbing term on every facet; with only
    source/absorbing facets listed both oracles must agree up to that factor 2 (linearity)."""
    pr = Problem(orc, (4, 3, 3), 3, hi=[0.02, 0.015, 0.015], perturb=0.1)
    tags = tag_box_boundary(pr.mesh)
    keep = tags.values == 2      # absorbing faces only -> 'every facet' == tag 2 ...
    src_keep = tags.values == 1  # ... except the source face, which the lossy form also absorbs on
    c0, rho0 = 1500.0, 1000.0
    nc = pr.mesh.num_cells
    m, src, absb, coeff = pr.linear_model_vectors(c0, rho0, tags)
    ml, srcl, absl, src2, lin, att = pr.lossy_model_vectors(c0, rho0, 0.0, tags)
    assert np.array_equal(ml, m) and np.array_equal(srcl, src) and not src2.any()
    f0, p0 = 0.5e6, 6e4
    dt = 0.5 * (0.02 / 4) / (c0 * 9)
    u1, v1 = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    u2, v2 = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    # same absorbing vector in both (the lossy 'all facets' one)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absl, f0, p0, c0, 0.0, 10 * dt * (1 - 1e-9), dt, u1, v1)
    orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, ml, srcl, absl, src2, f0, p0, c0, 0.0,
                  10 * dt * (1 - 1e-9), dt, u2, v2)
    assert np.abs(u1).max() > 0
    assert np.abs(u2 - 2 * u1).max() < 1e-12 * np.abs(u2).max()
    assert keep.any() and src_keep.any()


def test_threaded_cpu_baseline_matches_serial(orc):
    """orc_linear_rk4_mt (bench.py's cpu_baseline leg: one thread per x-slab of cells, even/odd
    passes) reproduces the serial restatement up to the order of interface additions."""
    pr = Problem(orc, (8, 3, 3), 3, hi=[0.02, 0.0075, 0.0075], perturb=0.1)
    tags = tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(1500.0, 1000.0, tags)
    dt = 0.5 * (0.02 / 8) / (1500.0 * 9)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, 8 * dt * (1 - 1e-9), dt, u, v)
    for nslabs in (1, 3, 8):
        off = np.linspace(0, 8, nslabs + 1).astype(np.int64) * 9
        u2, v2 = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
        ns = orc.linear_rk4_mt(pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0,
                               8 * dt * (1 - 1e-9), dt, u2, v2, off, fast=False)
        assert ns == 8 and np.abs(u).max() > 0
        assert np.abs(u2 - u).max() < 1e-13 * np.abs(u).max()
        assert np.abs(v2 - v).max() < 1e-13 * np.abs(v).max()


def _bend(x):
    """Volume-preserving quadratic map (det = 1): triquadratic cells represent it exactly."""
    y = x.copy()
    y[:, 0] += 0.3 * x[:, 1] ** 2 - 0.2 * x[:, 2] ** 2
    y[:, 1] += 0.25 * x[:, 2] ** 2
    return y


def test_second_order_geometry(orc):
    """27-node hexahedra (tensor node order): straight cells reproduce the trilinear factors; on
    curved cells the operator keeps K 1 = 0, symmetry, agreement with the dense-table evaluation,
    exact volume (det of the bend = 1) and exact energy of a physically linear field."""
    L = [1.2, 1.0, 0.8]
    a = Problem(orc, (3, 2, 2), 4, hi=L)
    b = Problem(orc, (3, 2, 2), 4, hi=L, order=2)
    assert b.mesh.geometry.dofmap.shape[1] == 27
    assert np.abs(a.G - b.G).max() < 1e-13 * np.abs(a.G).max()
    assert np.abs(a.detJ - b.detJ).max() < 1e-13 * np.abs(a.detJ).max()
    c = Problem(orc, (3, 2, 2), 4, hi=L, order=2, warp=_bend)
    assert np.abs(c.G - a.G).max() > 1e-3 * np.abs(a.G).max()          # really curved
    rng = np.random.default_rng(0)
    x, z = rng.standard_normal(c.ndofs), rng.standard_normal(c.ndofs)
    y = c.K(x)
    assert np.abs(c.K(np.ones(c.ndofs))).max() < 1e-12 * np.abs(y).max()
    assert abs(z @ y - x @ c.K(z)) < 1e-11 * abs(z @ y)
    assert np.abs(y - c.K(x, dense=True)).max() < 1e-12 * np.abs(y).max()
    assert abs(c.M(np.ones(c.ndofs)).sum() - np.prod(L)) < 1e-13 * np.prod(L)
    Xp = c.V.tabulate_dof_coordinates()           # physical dof coordinates (through the triquadratic map)
    u = Xp[:, 0].copy()                                                  # grad u = e_x everywhere
    assert abs(u @ c.K(u) - np.prod(L)) < 1e-12 * np.prod(L)
    # boundary weights: area of the bent x = 0 face, tangents (0.6y, 1, 0) and (-0.4z, 0.5z, 1)
    tags = tag_box_boundary(c.mesh)
    area_src = c.facet_diag(tags, 1, np.ones(c.mesh.num_cells)).sum()
    yy, zz = np.meshgrid(np.linspace(0, L[1], 2001), np.linspace(0, L[2], 2001), indexing="ij")
    dA = np.sqrt(1 + (0.6 * yy) ** 2 + (0.3 * yy * zz + 0.4 * zz) ** 2)  # |d(x')/dy x d(x')/dz| on x = 0
    ref = np.trapezoid(np.trapezoid(dA, zz[0], axis=1), yy[:, 0])
    assert abs(area_src - ref) < 1e-5 * ref      # GLL quadrature of a smooth non-polynomial integrand
