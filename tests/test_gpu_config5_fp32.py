"""BASELINE configs[4] arithmetic (256^3, p=6, fp32, x-slabs) and the full-size property tests of the
headline kernels.

fp32 parity of the whole RK4 loop (Linear.hpp:228-314 with T = float, as cpp/fenicsx-sf/tests/
test_operators3d/main.cpp:13 instantiates the operators): the HIP path in fp32 against the oracle
instantiated for float on the same inputs (tolerance 1e-4: both accumulate rounding over 20 steps in a
different order) and against the fp64 oracle (the fp32 discretisation error of the state, 1e-3)."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import tag_box_boundary
from util import Problem

pytestmark = pytest.mark.gpu

TOL_F32_VS_F32 = 1e-4
TOL_F32_VS_F64 = 1e-3
F0, P0, S0 = 0.5e6, 6.0e4, 1500.0


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _materials(mesh, L):
    cx = mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * L) & (cx < 0.6 * L)         # cortical-bone slab, BM7-SC1/main.cpp:37-40
    return np.where(sel, 2800.0, 1500.0), np.where(sel, 1850.0, 1000.0)


def _oracle_rk4(orc, pr, c, rho, tags, dt, nsteps, dtype):
    m, src, absb, coeff = pr.linear_model_vectors(c.astype(dtype), rho.astype(dtype), tags)
    u, v = np.zeros(pr.ndofs, dtype), np.zeros(pr.ndofs, dtype)
    ns = orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, P0, S0, 0.0, nsteps * dt * (1 - 1e-6), dt,
                        u, v, dtype=dtype)
    assert ns == nsteps
    return u, v


@pytest.mark.parametrize("P,n", [(4, (6, 6, 6)), (6, (4, 3, 3))])
@pytest.mark.parametrize("perturb,mode", [(0.15, "trilinear"), (0.0, "affine"), (0.15, "stream")])
def test_fp32_linear_rk4_vs_float_and_double_oracle(orc, P, n, perturb, mode):
    L = 0.012
    nsteps = 20
    pr32 = Problem(orc, n, P, hi=[L, L, L], perturb=perturb, dtype=np.float32)
    pr64 = Problem(orc, n, P, hi=[L, L, L], perturb=perturb)
    c, rho = _materials(pr64.mesh, L)
    tags = tag_box_boundary(pr64.mesh)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    u32, v32 = _oracle_rk4(orc, pr32, c, rho, tags, dt, nsteps, np.float32)
    u64, v64 = _oracle_rk4(orc, pr64, c, rho, tags, dt, nsteps, np.float64)
    assert np.abs(u64).max() > 0
    ctx = fa.Context(0, geometry="stream" if mode == "stream" else None)
    model = fa.LinearSpectralExplicit(pr32.mesh, tag_box_boundary(pr32.mesh), P, c.astype(np.float32),
                                      rho.astype(np.float32), F0, P0, S0, 4, dt, V=pr32.V, ctx=ctx)
    assert model.data.geometry_mode() == mode and model.data.dtype == np.float32
    model.init()
    model.rk4_steps(0.0, dt, nsteps)
    u, v = model.u_sol().x.array.copy(), model.v_n.x.array.copy()
    assert u.dtype == np.float32
    assert relmax(u, u32) < TOL_F32_VS_F32 and relmax(v, v32) < TOL_F32_VS_F32
    assert relmax(u, u64) < TOL_F32_VS_F64 and relmax(v, v64) < TOL_F32_VS_F64
    model.close()
    ctx.close()


@pytest.mark.parametrize("P,n", [(6, (6, 3, 3)), (4, (8, 4, 4))])
def test_fp32_two_slabs_in_process(orc, P, n):
    """configs[4]'s partition in small: two x-slabs in fp32 through the library's pack / ordered-sum /
    stage kernels (in-process transport) against the single-rank float and double oracles; the
    interface plane is bit-identical on both sharers."""
    L = [0.024, 0.012, 0.012]
    nsteps = 15
    pr32 = Problem(orc, n, P, hi=L, perturb=0.1, dtype=np.float32)
    pr64 = Problem(orc, n, P, hi=L, perturb=0.1)
    c, rho = _materials(pr64.mesh, L[0])
    tags = tag_box_boundary(pr64.mesh)
    dt = 0.5 * (L[0] / n[0]) / (c.max() * P**2)
    u32, v32 = _oracle_rk4(orc, pr32, c, rho, tags, dt, nsteps, np.float32)
    u64, v64 = _oracle_rk4(orc, pr64, c, rho, tags, dt, nsteps, np.float64)
    size = 2
    ctxs = [fa.Context(0) for _ in range(size)]
    fa.Context.init_local_group(ctxs)
    models, offs = [], []
    for r in range(size):
        mesh = fa.BoxMesh([0, 0, 0], L, n, rank=r, size=size, perturb=0.1, dtype=np.float32)
        V = fa.FunctionSpace(mesh, P)
        cr, rr = _materials(mesh, L[0])
        models.append(fa.LinearSpectralExplicit(mesh, tag_box_boundary(mesh), P, cr.astype(np.float32),
                                                rr.astype(np.float32), F0, P0, S0, 4, dt, V=V, ctx=ctxs[r]))
        assert models[-1].data.geometry_mode() == "trilinear"
        offs.append(V.global_offset)
    fa.group_finish_setup(models)
    for m in models:
        m.init()
    fa.group_rk4_steps(models, 0.0, dt, nsteps)
    us = []
    for r, mdl in enumerate(models):
        k = mdl.data.ndofs
        u = mdl.u_sol().x.array.copy()
        us.append(u)
        assert np.abs(u - u32[offs[r]:offs[r] + k]).max() < TOL_F32_VS_F32 * np.abs(u32).max()
        assert np.abs(u - u64[offs[r]:offs[r] + k]).max() < TOL_F32_VS_F64 * np.abs(u64).max()
    plane = len(us[0]) - (offs[1] - offs[0])
    assert plane > 0 and np.array_equal(us[0][-plane:], us[1][:plane])
    for mdl in models:
        mdl.close()
    for cx in ctxs:
        cx.close()


def _full_size_properties(geometry, P, ncell, L, dtype=np.float64, tol_sym=1e-10, tol_one=1e-11):
    """Size-independent properties of the operator at a benchmark size: K 1 = 0, sum(K x) = 0, symmetry,
    sum(m) = volume."""
    ncell = (ncell,) * 3 if np.isscalar(ncell) else tuple(ncell)     # cells per axis; L = edge of the first axis' box
    h = L / ncell[0]
    vol = h**3 * ncell[0] * ncell[1] * ncell[2]
    m = fa.BoxMesh([0, 0, 0], [h * k for k in ncell], ncell, dtype=dtype)
    V = fa.FunctionSpace(m, P)
    ctx = fa.Context(0, geometry=geometry)
    d = fa.SpectralOperatorData(V, ctx)
    assert d.geometry_mode() == geometry
    n, nc = V.num_dofs, m.num_cells
    rng = np.random.default_rng(0)
    x, z = rng.standard_normal(n).astype(dtype), rng.standard_normal(n).astype(dtype)
    coef = np.full(nc, -1e-3, dtype)
    y = d.stiffness(x, coef, np.zeros(n, dtype))
    scale = np.abs(y).max()
    assert np.isfinite(scale) and scale > 0
    assert np.abs(d.stiffness(np.ones(n, dtype), coef, np.zeros(n, dtype))).max() < tol_one * scale
    assert abs(y.astype(np.float64).sum()) < 1e2 * tol_sym * np.abs(y).astype(np.float64).sum()
    yz = d.stiffness(z, coef, np.zeros(n, dtype))
    zy = z.astype(np.float64) @ y.astype(np.float64)
    assert abs(zy - x.astype(np.float64) @ yz.astype(np.float64)) < tol_sym * abs(zy)
    mm = d.mass(np.ones(n, dtype), np.ones(nc, dtype), np.zeros(n, dtype))
    assert abs(mm.astype(np.float64).sum() - vol) < (1e-12 if dtype == np.float64 else 1e-5) * vol
    info = d.info()
    d.close()
    ctx.close()
    return n, info


def test_full_size_properties_headline_trilinear_kernel():
    """BASELINE configs[1] through the headline kernel itself (32-element / 8-wave trilinear blocks)."""
    n, info = _full_size_properties("trilinear", 4, 64, 0.12)
    assert n == 16974593 and info["nblocks"] == 262144 // 32


@pytest.mark.parametrize("geometry", ["trilinear", "stream"])
def test_full_size_properties_p7(geometry):
    """64^3 p=7 (90.5 M dofs, 8-element blocks): the degree of BASELINE configs[2] at a size that needs the
    same block layout and kernels as 128^3."""
    n, info = _full_size_properties(geometry, 7, 64, 0.12)
    assert n == (64 * 7 + 1) ** 3 and info["nblocks"] == 262144 // 8


def test_full_size_properties_config3_slab():
    """One rank's 32 x 256 x 256 x-slab of BASELINE configs[3] (256^3 p=4 fp64 over 8 GPUs): 135 M dofs, 65 536 blocks.
    (The whole 256^3 box on one GPU -- 1.08e9 dofs -- is checked by tools/gpu_config3_full.py; its layout alone
    takes minutes of host time, too long for this suite.)"""
    n, info = _full_size_properties("trilinear", 4, (32, 256, 256), 0.12 * 32 / 64)
    assert n == 129 * 1025 * 1025 and info["nblocks"] == 32 * 256 * 256 // 32


def test_full_size_properties_p6_fp32():
    """64^3 p=6 fp32 (57 M dofs): configs[4]'s degree and arithmetic, one GPU's share of it in x."""
    n, _ = _full_size_properties("trilinear", 6, 64, 0.12, dtype=np.float32, tol_sym=2e-4, tol_one=2e-4)
    assert n == (64 * 6 + 1) ** 3


@pytest.mark.parametrize("P", [5, 6, 7])
@pytest.mark.parametrize("perturb,mode", [(0.2, "trilinear"), (0.0, "affine")])
def test_packed_fp32_kernels(orc, P, perturb, mode):
    """Option "pack32" (auto: where it measured faster): fp32 at degrees 5-7 on the per-cell geometry paths works on two elements per
    wave in packed float2 (kernels.hpp, elem_compute_pk).  Operator action (odd number of cells: the last pair
    is a lone element) and the Lossy RK4 loop (two operator inputs) against the float oracle and against the
    scalar kernel."""
    pr = Problem(orc, (3, 3, 3), P, hi=[0.012, 0.012, 0.012], perturb=perturb, dtype=np.float32)
    rng = np.random.default_rng(P)
    x = rng.standard_normal(pr.ndofs).astype(np.float32)
    coef = rng.uniform(0.5, 2.0, pr.mesh.num_cells).astype(np.float32)
    ref = pr.K(x, coef)
    ys = {}
    for pk in (1, 0):
        c = fa.Context(0)
        c.set_option("pack32", pk)
        d = fa.SpectralOperatorData(pr.V, c)
        assert d.geometry_mode() == mode and d.uses_pack32() == bool(pk)
        ys[pk] = d.stiffness(x, coef, np.zeros(pr.ndofs, np.float32))
        assert relmax(ys[pk], ref) < 5e-5
        assert relmax(d.mass(x, coef, np.zeros(pr.ndofs, np.float32)), pr.M(x, coef)) < 1e-5
        d.close()
        c.close()
    assert relmax(ys[1], ys[0]) < 2e-5
    # Lossy model (NF = 2) through the packed kernel
    nc = pr.mesh.num_cells
    c0, rho0 = np.full(nc, 1500.0, np.float32), np.full(nc, 1000.0, np.float32)
    delta = np.full(nc, fa.compute_diffusivity_of_sound(2 * np.pi * F0, 1500.0, 0.2), np.float32)
    tags = tag_box_boundary(pr.mesh)
    dt = 0.5 * (0.012 / 3) / (1500.0 * P**2)
    nsteps = 10
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c0, rho0, delta, tags)
    u, v = np.zeros(pr.ndofs, np.float32), np.zeros(pr.ndofs, np.float32)
    orc.lossy_rk4(3, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, F0, P0, S0, 0.0, nsteps * dt * (1 - 1e-6), dt, u, v,
                  dtype=np.float32)
    ctx = fa.Context(0)
    ctx.set_option("pack32", 1)
    model = fa.LossySpectralExplicit(pr.mesh, tags, P, c0, rho0, delta, F0, P0, S0, 4, dt, V=pr.V, ctx=ctx)
    assert model.data.uses_pack32()
    model.init()
    model.rk4_steps(0.0, dt, nsteps)
    un = model.u_sol().x.array
    assert np.abs(u).max() > 0 and relmax(un, u) < 2e-4
    model.close()
    ctx.close()
