"""Diagonal-metric form of the affine stiffness kernel (option "diag_metric", fus_op_uses_diag_metric): on cells
with mutually orthogonal edges G of stiffness::transform (spectral_op.hpp:113-130) is diag(g) w_q, and the action
of spectral_op.hpp:173-243 is three 1-D stiffness contractions.  Same parity bar as every other kernel: against
the oracle (which forms the full G), on axis-aligned and rotated boxes, and the general affine form stays
covered on the same boxes (option off) and on a sheared mesh (where the diagonal form must not be chosen)."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import tag_box_boundary
from util import Problem

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _mapped(orc, n, P, hi, A, dtype=np.float64):
    """Box mesh with every vertex mapped by the matrix A (rotation: orthogonal edges stay orthogonal; shear: not)."""
    pr = Problem(orc, n, P, hi=hi, dtype=dtype)
    x = pr.mesh.geometry.x
    x[:] = (x.astype(np.float64) @ np.asarray(A).T).astype(x.dtype)
    pr.G, pr.detJ = orc.geometry(3, pr.mesh.geometry.x, pr.mesh.geometry.dofmap, pr.nodes, pr.wts, dtype=dtype)
    return pr


def _rotation():
    a, b = 0.7, -0.4
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    return Rz @ Rx


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 5e-5)])
@pytest.mark.parametrize("deterministic", [0, 1])
def test_operator_vs_oracle_on_boxes(orc, P, dtype, tol, deterministic):
    pr = Problem(orc, (4, 3, 3), P, hi=[1.5, 1.0, 0.8], dtype=dtype)
    rng = np.random.default_rng(P)
    x = rng.standard_normal(pr.ndofs).astype(dtype)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells).astype(dtype)
    ref = pr.K(x, coef)
    ys = {}
    for on in (1, 0):
        c = fa.Context(0, deterministic=deterministic)
        c.set_option("diag_metric", on)
        d = fa.SpectralOperatorData(pr.V, c)
        assert d.geometry_mode() == "affine" and d.uses_diag_metric() == bool(on)
        y0 = rng.standard_normal(pr.ndofs).astype(dtype)
        ys[on] = d.stiffness(x, coef, y0.copy()) - y0        # y is accumulated, not overwritten
        assert relmax(ys[on], ref) < tol + (1e-6 if dtype == np.float32 else 1e-15)
        assert relmax(d.mass(x, coef, np.zeros(pr.ndofs, dtype)), pr.M(x, coef)) < (1e-13 if dtype == np.float64 else 1e-5)
        d.close()
        c.close()
    assert relmax(ys[1], ys[0]) < (1e-13 if dtype == np.float64 else 2e-5)


@pytest.mark.parametrize("P", [3, 4, 7])
def test_rotated_box_is_diagonal_and_sheared_box_is_not(orc, P):
    rng = np.random.default_rng(11)
    pr = _mapped(orc, (3, 3, 2), P, [1.2, 1.0, 0.7], _rotation())
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    c = fa.Context(0)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.is_affine() and d.uses_diag_metric()
    assert relmax(d.stiffness(x, coef, np.zeros(pr.ndofs)), pr.K(x, coef)) < 1e-12
    d.close()
    shear = np.array([[1.0, 0.3, 0.0], [0.0, 1.0, 0.2], [0.1, 0.0, 1.0]])
    ps = _mapped(orc, (3, 3, 2), P, [1.2, 1.0, 0.7], shear)
    d = fa.SpectralOperatorData(ps.V, c)
    assert d.is_affine() and not d.uses_diag_metric()         # parallelepipeds, edges not orthogonal: general affine form
    assert relmax(d.stiffness(x, coef, np.zeros(ps.ndofs)), ps.K(x, coef)) < 1e-12
    d.close()
    c.close()


@pytest.mark.parametrize("P,dtype,tol", [(4, np.float64, 1e-10), (6, np.float32, 2e-4)])
def test_linear_and_lossy_rk4_on_boxes(orc, P, dtype, tol):
    """Both model kernels (one and two operator inputs) through the diagonal-metric form: 10 RK4 steps vs the oracle."""
    L, n, nsteps = 0.012, (4, 3, 3), 10
    pr = Problem(orc, n, P, hi=[L * 4 / 3, L, L], dtype=dtype)
    cx = pr.mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * L) & (cx < 0.8 * L)
    c0 = np.where(sel, 2800.0, 1500.0).astype(dtype)
    rho = np.where(sel, 1850.0, 1000.0).astype(dtype)
    tags = tag_box_boundary(pr.mesh)
    f0, p0, s0 = 0.5e6, 6e4, 1500.0
    dt = 0.5 * (L / 3) / (c0.max() * P**2)
    tf = nsteps * dt * (1 - 1e-6)
    ctx = fa.Context(0)
    m, src, absb, coeff = pr.linear_model_vectors(c0, rho, tags)
    u, v = np.zeros(pr.ndofs, dtype), np.zeros(pr.ndofs, dtype)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, s0, 0.0, tf, dt, u, v, dtype=dtype)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c0, rho, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    assert model.data.uses_diag_metric()
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < tol and relmax(vn.x.array, v) < tol
    model.close()
    delta = np.where(sel, fa.compute_diffusivity_of_sound(2 * np.pi * f0, 2800.0, 46.0),
                     fa.compute_diffusivity_of_sound(2 * np.pi * f0, 1500.0, 0.2)).astype(dtype)
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c0, rho, delta, tags)
    u, v = np.zeros(pr.ndofs, dtype), np.zeros(pr.ndofs, dtype)
    orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, f0, p0, s0, 0.0, tf, dt, u, v, dtype=dtype)
    model = fa.LossySpectralExplicit(pr.mesh, tags, P, c0, rho, delta, f0, p0, s0, 4, dt, V=pr.V, ctx=ctx)
    assert model.data.uses_diag_metric()
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < tol and relmax(vn.x.array, v) < tol
    model.close()
    ctx.close()
