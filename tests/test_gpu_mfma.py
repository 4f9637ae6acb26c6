"""Matrix-core form of the index-1 / index-2 contractions (option "mfma"; degrees 6 and 7 on the per-cell
geometry kernels, kernels.hpp elem_compute_mfma): the (N x N) . (N x N^2) products of the reference's
contract<T, N, N, N, N, bool> (cpp/fenicsx-sf/common/sum_factorisation.hpp:70-86, called at
spectral_op.hpp:199-210, 222-238) as v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 tiles.  Same parity
bar as the vector form: operator action against the oracle (fp64 1e-12, fp32 5e-5) and 10 RK4 steps of the
three models (1e-10), on distorted (trilinear) and affine meshes, against the vector form to rounding."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import tag_box_boundary
from util import Problem

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _ctx(mfma, geometry=None):
    c = fa.Context(0, geometry=geometry)
    c.set_option("mfma", mfma)
    return c


@pytest.mark.parametrize("P", [6, 7])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 5e-5)])
@pytest.mark.parametrize("perturb,mode", [(0.2, "trilinear"), (0.0, "affine")])
def test_operator_vs_oracle(orc, P, dtype, tol, perturb, mode):
    pr = Problem(orc, (3, 3, 2), P, hi=[1.5, 1.0, 0.8], perturb=perturb, dtype=dtype)
    rng = np.random.default_rng(P)
    x = rng.standard_normal(pr.ndofs).astype(dtype)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells).astype(dtype)
    ref = pr.K(x, coef)
    ys = {}
    for mf in (1, 0):
        c = _ctx(mf)
        d = fa.SpectralOperatorData(pr.V, c)
        assert d.geometry_mode() == mode and d.uses_mfma() == bool(mf)
        y0 = rng.standard_normal(pr.ndofs).astype(dtype)
        y = d.stiffness(x, coef, y0.copy()) - y0          # y is accumulated, not overwritten
        assert relmax(y, ref) < (tol if dtype == np.float64 else tol) + (1e-6 if dtype == np.float32 else 1e-15)
        ys[mf] = y
        d.close()
        c.close()
    assert relmax(ys[1], ys[0]) < (1e-13 if dtype == np.float64 else 2e-5)


def test_mfma_is_not_used_where_no_variant_exists(orc):
    pr = Problem(orc, (3, 3, 2), 4, perturb=0.2)
    c = _ctx(1)
    d = fa.SpectralOperatorData(pr.V, c)                  # degree 4: an N = 5 contraction would pad 16x16x4 tiles to 19 %
    assert not d.uses_mfma()
    d.close()
    c.close()
    pr7 = Problem(orc, (2, 2, 2), 7, perturb=0.2)
    c = _ctx(1, geometry="stream")                        # streamed geometry keeps the vector form
    d = fa.SpectralOperatorData(pr7.V, c)
    assert not d.uses_mfma()
    d.close()
    c.close()
    c = fa.Context(0, deterministic=1)                    # conflict-free rounds keep the vector form
    c.set_option("mfma", 1)
    d = fa.SpectralOperatorData(pr7.V, c)
    assert not d.uses_mfma()
    x = np.random.default_rng(0).standard_normal(pr7.ndofs)
    assert relmax(d.stiffness(x, np.ones(pr7.mesh.num_cells), np.zeros(pr7.ndofs)), pr7.K(x)) < 1e-12
    d.close()
    c.close()


@pytest.mark.parametrize("P", [6, 7])
def test_three_models_rk4_vs_oracle(orc, P):
    L = 0.012
    n = (3, 3, 3)
    pr = Problem(orc, n, P, hi=[L, L, L], perturb=0.15)
    nc = pr.mesh.num_cells
    cx = pr.mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * L) & (cx < 0.6 * L)
    c, rho = np.where(sel, 2800.0, 1500.0), np.where(sel, 1850.0, 1000.0)
    tags = tag_box_boundary(pr.mesh)
    f0, s0 = 0.5e6, 1500.0
    w0 = 2 * np.pi * f0
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 10
    tf = nsteps * dt * (1 - 1e-9)
    ctx = _ctx(1)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, 6e4, s0, 0.0, tf, dt, u, v)
    model = fa.LinearSpectralExplicit(pr.mesh, tags, P, c, rho, f0, 6e4, s0, 4, dt, V=pr.V, ctx=ctx)
    assert model.data.uses_mfma() and model.data.geometry_mode() == "trilinear"
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < 1e-10 and relmax(vn.x.array, v) < 1e-10
    model.close()
    delta = np.where(c > 2000.0, fa.compute_diffusivity_of_sound(w0, 2800.0, 46.0), fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    beta = np.where(c > 2000.0, 6.0, 3.5)
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, f0, 6e4, s0, 0.0, tf, dt, u, v)
    model = fa.LossySpectralExplicit(pr.mesh, tags, P, c, rho, delta, f0, 6e4, s0, 4, dt, V=pr.V, ctx=ctx)
    assert model.data.uses_mfma()
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < 1e-10 and relmax(vn.x.array, v) < 1e-10
    model.close()
    n1 = -2.0 * beta / rho**2 / c**4
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, f0, 6e6, s0,
                       0.0, tf, dt, u, v)
    model = fa.WesterveltSpectralExplicit(pr.mesh, tags, P, c, rho, delta, beta, f0, 6e6, s0, 4, dt, V=pr.V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, tf)
    assert np.abs(u).max() > 0 and relmax(un.x.array, u) < 1e-10 and relmax(vn.x.array, v) < 1e-10
    model.close()
    ctx.close()


def test_fp32_p6_two_slabs_mfma(orc):
    """configs[4]'s arithmetic through the matrix-core variant: two x-slabs, fp32, p = 6."""
    P, n, L, nsteps = 6, (4, 2, 2), [0.024, 0.012, 0.012], 10
    pr32 = Problem(orc, n, P, hi=L, perturb=0.1, dtype=np.float32)
    c0, rho0 = np.full(pr32.mesh.num_cells, 1500.0, np.float32), np.full(pr32.mesh.num_cells, 1000.0, np.float32)
    tags = tag_box_boundary(pr32.mesh)
    dt = 0.5 * (L[0] / n[0]) / (1500.0 * P**2)
    m, src, absb, coeff = pr32.linear_model_vectors(c0, rho0, tags)
    u, v = np.zeros(pr32.ndofs, np.float32), np.zeros(pr32.ndofs, np.float32)
    orc.linear_rk4(3, pr32.N, pr32.dm, pr32.G, pr32.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, nsteps * dt * (1 - 1e-6),
                   dt, u, v, dtype=np.float32)
    ctxs = [_ctx(1) for _ in range(2)]
    fa.Context.init_local_group(ctxs)
    models, offs = [], []
    for r in range(2):
        mesh = fa.BoxMesh([0, 0, 0], L, n, rank=r, size=2, perturb=0.1, dtype=np.float32)
        V = fa.FunctionSpace(mesh, P)
        k = mesh.num_cells
        models.append(fa.LinearSpectralExplicit(mesh, tag_box_boundary(mesh), P, np.full(k, 1500.0, np.float32),
                                                np.full(k, 1000.0, np.float32), 0.5e6, 6e4, 1500.0, 4, dt, V=V, ctx=ctxs[r]))
        assert models[-1].data.uses_mfma()
        offs.append(V.global_offset)
    fa.group_finish_setup(models)
    for mdl in models:
        mdl.init()
    fa.group_rk4_steps(models, 0.0, dt, nsteps)
    assert np.abs(u).max() > 0
    for r, mdl in enumerate(models):
        k = mdl.data.ndofs
        assert np.abs(mdl.u_sol().x.array - u[offs[r]:offs[r] + k]).max() < 1e-4 * np.abs(u).max()
        mdl.close()
    for cx in ctxs:
        cx.close()


def test_mfma_4x4x4_index1_is_the_default_at_p7_fp64_trilinear(orc):
    """v_mfma_f64_4x4x4_4b_f64 (kernels.hpp mf4_contract_b): the index-1 contraction of the p=7 fp64 trilinear kernel runs
    on the matrix cores straight from the registers BY DEFAULT (BASELINE configs[2]'s "MFMA per-element GEMM path");
    operator action and the fused RK4 loop against the oracle on distorted cells, mirrored cells included, and the flag
    is off where the path is not taken (affine cells, fp32, other degrees, the opt-in 16x16x4 form)."""
    pr = Problem(orc, (3, 3, 2), 7, hi=[0.012, 0.012, 0.008], perturb=0.2)
    c = fa.Context(0)
    d = fa.SpectralOperatorData(pr.V, c)
    assert d.geometry_mode() == "trilinear" and d.uses_mfma4() and not d.uses_mfma()
    rng = np.random.default_rng(7)
    x, coef = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, pr.mesh.num_cells)
    y0 = rng.standard_normal(pr.ndofs)
    y = d.stiffness(x, coef, y0.copy())
    assert np.abs(y - (y0 + pr.K(x, coef))).max() < 1e-12 * np.abs(y).max()
    d.close()
    nc = pr.mesh.num_cells
    cc, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    tags = fa.tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(cc, rho, tags)
    dt = 0.4 * (0.012 / 3) / (1500.0 * 49)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, 0.5e6, 6e4, 1500.0, 0.0, 6 * dt * (1 + 1e-12), dt, u, v)
    mdl = fa.LinearSpectralExplicit(pr.mesh, tags, 7, cc, rho, 0.5e6, 6e4, 1500.0, 4, dt, V=pr.V, ctx=c)
    assert mdl.data.uses_mfma4()
    mdl.init()
    mdl.rk4_steps(0.0, dt, 6)
    assert np.abs(u).max() > 0 and np.abs(mdl.u_sol().x.array - u).max() < 1e-10 * np.abs(u).max()
    mdl.close()
    for P, perturb, dtype in ((7, 0.0, np.float64), (7, 0.2, np.float32), (6, 0.2, np.float64)):
        q = Problem(orc, (2, 2, 2), P, perturb=perturb, dtype=dtype)
        dq = fa.SpectralOperatorData(q.V, c)
        assert not dq.uses_mfma4()
        dq.close()
    c.close()
    c1 = fa.Context(0)
    c1.set_option("mfma", 1)
    d1 = fa.SpectralOperatorData(pr.V, c1)
    assert d1.uses_mfma() and not d1.uses_mfma4()
    d1.close()
    c1.close()
