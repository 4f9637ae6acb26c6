"""Unstructured meshes: the reference's own Gmsh fixture (cpp/fenicsx-sf/tests/test_operators3d/
mesh.xdmf, committed as data in tests/golden/ref_test_operators3d_mesh.npz) read by this
repository's XDMF/HDF5 reader, the geometric tensor-dofmap builder, and the operators on it with the
reference test's input recipe (main.cpp:60-67, 78-79, 85, 129)."""
import os

import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd.unstructured import VTK_TO_TENSOR, HexFunctionSpace, HexMesh, read_xdmf_hex_mesh
from util import Problem

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_test_operators3d_mesh.npz")
REF_XDMF = "/root/reference/cpp/fenicsx-sf/tests/test_operators3d/mesh.xdmf"


@pytest.fixture(scope="module")
def ref_mesh():
    g = np.load(GOLD)
    mesh = HexMesh(g["geometry"], g["topology_vtk"][:, VTK_TO_TENSOR])
    tags = mesh.facet_tags(g["facet_topology"], g["facet_values"])
    return mesh, tags


def test_xdmf_hdf5_reader_matches_fixture():
    if not os.path.exists(REF_XDMF):
        pytest.skip("reference tree absent")
    mesh, cell_vals, tags = read_xdmf_hex_mesh(REF_XDMF)
    g = np.load(GOLD)
    assert np.array_equal(mesh.geometry.x, g["geometry"])
    assert np.array_equal(mesh.geometry.dofmap, g["topology_vtk"][:, VTK_TO_TENSOR])
    assert np.array_equal(cell_vals, g["cell_values"]) and len(tags.cells) == len(g["facet_values"])


def test_reference_mesh_space_and_oracle_kats(orc, ref_mesh):
    mesh, tags = ref_mesh
    assert mesh.num_cells == 6312 and mesh.entity_counts() == (7939, 21624, 19998, 6312)
    assert len(tags.cells) == 2124 and set(tags.values.tolist()) == {1}
    V = HexFunctionSpace(mesh, 4)            # raises unless #dofs matches the topological count
    assert V.num_dofs == 7939 + 3 * 21624 + 9 * 19998 + 27 * 6312
    wts, D = orc.gll_weights_at(V.nodes1d), orc.dphi(V.nodes1d)
    G, dJ = orc.geometry(3, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    assert abs(dJ.sum() - 1.0) < 1e-13                       # unit cube
    n, nc = V.num_dofs, mesh.num_cells
    K = lambda x: orc.stiffness(3, 5, V.tensor_dofmap, G, D, np.ones(nc), x, np.zeros(n))  # noqa: E731
    assert np.abs(K(np.ones(n))).max() < 1e-14
    X = V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(np.pi * X[:, 1])            # the reference test's input function
    exact = (0.5 + np.sin(2) / 4) * 0.5 + np.pi**2 * (0.5 - np.sin(2) / 4) * 0.5   # int |grad u|^2
    assert abs(u @ K(u) - exact) < 1e-9 * exact              # spectral accuracy on the Gmsh mesh
    area = orc.facet_diag(3, tags.cells, tags.local_facets, np.ones(nc), mesh.geometry.x, mesh.geometry.dofmap,
                          V.nodes1d, wts, V.tensor_dofmap, n).sum()
    assert abs(area - 6.0) < 1e-12


def test_layout_on_unstructured_mesh(ref_mesh):
    mesh, _ = ref_mesh
    V = HexFunctionSpace(mesh, 3)
    info = fa.layout_check(3, V.tensor_dofmap, mesh.cell_centroids(), block_elems=32, waves=4)
    assert info[1] + info[2] == V.num_dofs and info[0] == -(-mesh.num_cells // 32)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [4, 6])
def test_reference_operator_test_on_its_own_mesh(orc, ref_mesh, P):
    """cpp/fenicsx-sf/tests/test_operators3d/main.cpp with its Gmsh mesh (the commented read_mesh
    block :40-48): P = 4 (and 6: the single-register-set kernel on an unstructured mesh),
    u = sin(x) cos(pi y), c0 = 1.5e-3, rho0 = 1e-3, mass coefficient 1/(rho c^2),
    stiffness coefficient -1/rho; HIP operators vs the oracle, then 10 Linear RK4 steps."""
    mesh, tags = ref_mesh
    V = HexFunctionSpace(mesh, P)
    n, nc = V.num_dofs, mesh.num_cells
    wts, D = orc.gll_weights_at(V.nodes1d), orc.dphi(V.nodes1d)
    G, dJ = orc.geometry(3, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    X = V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(np.pi * X[:, 1])
    c0, rho0 = 1.5e-3, 1e-3
    ctx = fa.Context(0)
    d = fa.SpectralOperatorData(V, ctx)
    assert not d.is_affine()
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    mc, sc = np.full(nc, 1 / rho0 / c0 / c0), np.full(nc, -1 / rho0)
    assert rel(d.mass(u, mc, np.zeros(n)), orc.mass(3, P + 1, V.tensor_dofmap, dJ, mc, u, np.zeros(n))) < 1e-14
    assert rel(d.stiffness(u, sc, np.zeros(n)),
               orc.stiffness(3, P + 1, V.tensor_dofmap, G, D, sc, u, np.zeros(n), fast=True)) < 1e-12
    Gd, dJd = d.geometry()
    assert rel(Gd, G) < 1e-12 and rel(dJd, dJ) < 1e-13
    d.close()
    # Linear model, water, every boundary facet is tag 1 (source) in this file
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    m = orc.mass(3, P + 1, V.tensor_dofmap, dJ, 1 / (rho * c * c), np.ones(n), np.zeros(n))
    src = orc.facet_diag(3, tags.cells, tags.local_facets, 1 / rho, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d,
                         wts, V.tensor_dofmap, n)
    hmin = np.cbrt(dJ.reshape(nc, -1).sum(axis=1)).min()
    dt = 0.2 * hmin / (1500.0 * P**2)
    uo, vo = np.zeros(n), np.zeros(n)
    orc.linear_rk4(3, P + 1, V.tensor_dofmap, G, D, -1 / rho, m, src, np.zeros(n), 5e3, 6e4, 1500.0, 0.0,
                   10 * dt * (1 - 1e-9), dt, uo, vo, fast=True)
    model = fa.LinearSpectralExplicit(mesh, tags, P, c, rho, 5e3, 6e4, 1500.0, 4, dt, V=V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, 10 * dt * (1 - 1e-9))
    assert np.abs(uo).max() > 0 and rel(un.x.array, uo) < 1e-10 and rel(vn.x.array, vo) < 1e-10
    model.close()
    ctx.close()


def test_point_evaluation(ref_mesh):
    """evaluate(): a polynomial of the space's degree in PHYSICAL coordinates is not in the space on
    non-affine cells, but a trilinear-mapped field is reproduced; use an affine box for exactness and
    the Gmsh mesh for the locate/Newton machinery (smooth field, spectral accuracy)."""
    from fenicsxfus_amd.evaluate import evaluate, locate

    box = fa.BoxMesh([0, 0, 0], [1.0, 0.8, 0.6], (3, 2, 2))
    Vb = fa.FunctionSpace(box, 4)
    Xd = Vb.tabulate_dof_coordinates()
    f = lambda X: X[:, 0] ** 4 - 2 * X[:, 1] ** 3 * X[:, 2] + X[:, 0] * X[:, 1] * X[:, 2] ** 2 + 1.0  # noqa: E731
    rng = np.random.default_rng(0)
    pts = rng.uniform([0, 0, 0], [1.0, 0.8, 0.6], size=(200, 3))
    assert np.abs(evaluate(Vb, f(Xd), pts) - f(pts)).max() < 1e-13
    assert np.isnan(evaluate(Vb, f(Xd), np.array([[1.5, 0.1, 0.1]]))).all()
    mesh, _ = ref_mesh
    V = HexFunctionSpace(mesh, 4)
    X = V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(np.pi * X[:, 1])
    pts = rng.uniform(0.02, 0.98, size=(300, 3))
    cell, _ = locate(mesh, pts)
    assert (cell >= 0).all()
    assert np.abs(evaluate(V, u, pts) - np.sin(pts[:, 0]) * np.cos(np.pi * pts[:, 1])).max() < 1e-6


# ---- the naive 2-D operator test's own quadrilateral mesh -------------------------------------------
GOLD2 = os.path.join(os.path.dirname(__file__), "golden", "ref_test_operators2d_mesh.npz")
REF_XDMF2 = "/root/reference/cpp/fenicsx-sf-naive/tests/test_operators2d/mesh_1/mesh.xdmf"


@pytest.fixture(scope="module")
def ref_mesh2d():
    from fenicsxfus_amd.unstructured import VTK_QUAD_TO_TENSOR, QuadMesh
    g = np.load(GOLD2)
    mesh = QuadMesh(g["geometry"], g["topology_vtk"][:, VTK_QUAD_TO_TENSOR])
    return mesh, mesh.facet_tags(g["facet_topology"], g["facet_values"])


def _recipe2d(V, nc):
    """Inputs of cpp/fenicsx-sf-naive/tests/test_operators2d/main.cpp:63-104, 107, 154, 202, 249, 296:
    u = 1, u_n = 2, v_n = cos(x) sin(pi y), w_n = u_n^2, c0 = 1.5, rho0 = 1, delta0 = beta0 = 10;
    (operator, input, per-cell coefficient) of the five spectral-vs-FFCx comparisons m1, m2, m3, b1, b2."""
    X = V.tabulate_dof_coordinates()
    n = V.num_dofs
    u, un = np.ones(n), np.full(n, 2.0)
    vn = np.cos(X[:, 0]) * np.sin(np.pi * X[:, 1])
    c0, rho0, delta0, beta0 = 1.5, 1.0, 10.0, 10.0
    full = lambda v: np.full(nc, v)  # noqa: E731
    return [("mass", u, full(1.0 / rho0 / c0 / c0)),
            ("mass", un, full(-2.0 * beta0 / rho0**2 / c0**4)),
            ("mass", un * un, full(2.0 * beta0 / rho0**2 / c0**4)),
            ("stiffness", vn, full(-1.0 / rho0)),
            ("stiffness", vn, full(-delta0 / rho0 / c0 / c0))]


def test_reference_2d_mesh_reader_space_and_oracle_kats(orc, ref_mesh2d):
    mesh, tags = ref_mesh2d
    if os.path.exists(REF_XDMF2):
        m2, cv, t2 = fa.read_xdmf_mesh(REF_XDMF2)
        assert np.array_equal(m2.geometry.dofmap, mesh.geometry.dofmap) and np.array_equal(m2.geometry.x, mesh.geometry.x)
        assert len(cv) == 265 and np.array_equal(t2.cells, tags.cells)
    assert mesh.tdim == 2 and mesh.num_cells == 265 and mesh.entity_counts() == (296, 560, 0, 265)
    assert len(tags.cells) == 60                                 # every boundary edge is tagged
    V = HexFunctionSpace(mesh, 4)
    assert V.num_dofs == 296 + 3 * 560 + 9 * 265
    wts, D = orc.gll_weights_at(V.nodes1d), orc.dphi(V.nodes1d)
    G, dJ = orc.geometry(2, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    assert abs(dJ.sum() - 1.0) < 1e-13                           # unit square
    n, nc = V.num_dofs, mesh.num_cells
    ops = _recipe2d(V, nc)
    # known answers of the recipe: m1 sums to area / (rho c^2); K(const) = 0; energy of v_n
    m1 = orc.mass(2, 5, V.tensor_dofmap, dJ, ops[0][2], ops[0][1], np.zeros(n))
    assert abs(m1.sum() - 1.0 / 2.25) < 1e-13
    b1 = orc.stiffness(2, 5, V.tensor_dofmap, G, D, ops[3][2], ops[3][1], np.zeros(n))
    exact = (0.5 - np.sin(2) / 4) * 0.5 + np.pi**2 * (0.5 + np.sin(2) / 4) * 0.5   # int |grad v_n|^2
    assert abs(-(ops[3][1] @ b1) - exact) < 1e-8 * exact
    perim = orc.facet_diag(2, tags.cells, tags.local_facets, np.ones(nc), mesh.geometry.x, mesh.geometry.dofmap,
                           V.nodes1d, wts, V.tensor_dofmap, n).sum()
    assert abs(perim - 4.0) < 1e-12


@pytest.mark.gpu
def test_reference_2d_operator_test_on_its_own_mesh(orc, ref_mesh2d):
    """The five operator comparisons of cpp/fenicsx-sf-naive/tests/test_operators2d/main.cpp on its
    Gmsh mesh, HIP vs oracle, then 10 Linear RK4 steps (LinearSpectral2D, naive Linear.hpp:52-350)."""
    mesh, tags = ref_mesh2d
    P = 4
    V = HexFunctionSpace(mesh, P)
    n, nc = V.num_dofs, mesh.num_cells
    wts, D = orc.gll_weights_at(V.nodes1d), orc.dphi(V.nodes1d)
    G, dJ = orc.geometry(2, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    ctx = fa.Context(0)
    d = fa.SpectralOperatorData(V, ctx)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    for kind, x, coef in _recipe2d(V, nc):
        if kind == "mass":
            assert rel(fa.MassSpectral2D(V, d)(x, coef, np.zeros(n)),
                       orc.mass(2, P + 1, V.tensor_dofmap, dJ, coef, x, np.zeros(n))) < 1e-14
        else:
            assert rel(fa.StiffnessSpectral2D(V, d)(x, coef, np.zeros(n)),
                       orc.stiffness(2, P + 1, V.tensor_dofmap, G, D, coef, x, np.zeros(n))) < 1e-12
    d.close()
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    m = orc.mass(2, P + 1, V.tensor_dofmap, dJ, 1 / (rho * c * c), np.ones(n), np.zeros(n))
    src = orc.facet_diag(2, tags.cells, tags.local_facets, 1 / rho, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d,
                         wts, V.tensor_dofmap, n)
    hmin = np.sqrt(dJ.reshape(nc, -1).sum(axis=1)).min()
    dt = 0.2 * hmin / (1500.0 * P**2)
    uo, vo = np.zeros(n), np.zeros(n)
    orc.linear_rk4(2, P + 1, V.tensor_dofmap, G, D, -1 / rho, m, src, np.zeros(n), 5e3, 6e4, 1500.0, 0.0,
                   10 * dt * (1 - 1e-9), dt, uo, vo)
    model = fa.LinearSpectralExplicit(mesh, tags, P, c, rho, 5e3, 6e4, 1500.0, 4, dt, V=V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, 10 * dt * (1 - 1e-9))
    assert np.abs(uo).max() > 0 and rel(un.x.array, uo) < 1e-10 and rel(vn.x.array, vo) < 1e-10
    model.close()
    ctx.close()


@pytest.mark.parametrize("binary", [True, False])
def test_vtu_output_round_trip(ref_mesh2d, tmp_path, binary):
    """output.write_vtu: sub-cells on the GLL lattice tile every element exactly (areas / volumes add
    up to the domain's), vertices in VTK order (positive orientation), fields survive the round trip;
    hexahedra (box, permuted node order), quadrilaterals (the reference's Gmsh mesh)."""
    from fenicsxfus_amd.output import read_vtu, subcell_connectivity, write_vtu
    mesh = fa.BoxMesh([0, 0, 0], [1.5, 1.0, 0.8], (3, 2, 2), perturb=0.1)
    V = fa.FunctionSpace(mesh, 3, node_order=[3, 0, 2, 1])
    X = V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(X[:, 1]) + X[:, 2]
    p = str(tmp_path / "box.vtu")
    write_vtu(p, V, {"u": u, "v": 2 * u}, binary=binary)
    d = read_vtu(p)
    assert np.array_equal(d["Points"], X) and np.array_equal(d["u"], u) and np.array_equal(d["v"], 2 * u)
    conn = d["connectivity"].reshape(-1, 8)
    assert conn.shape[0] == mesh.num_cells * 27 and np.all(d["types"] == 12)
    assert np.array_equal(conn, subcell_connectivity(V))
    c = X[conn]                                   # sub-cell volume by the triple product at vertex 0 (positive)
    vol0 = np.einsum("ci,ci->c", np.cross(c[:, 1] - c[:, 0], c[:, 3] - c[:, 0]), c[:, 4] - c[:, 0])
    assert np.all(vol0 > 0)
    # quadrilaterals: shoelace areas of the sub-cells add up to the unit square
    qm, _ = ref_mesh2d
    Vq = HexFunctionSpace(qm, 4)
    p2 = str(tmp_path / "quad.vtu")
    write_vtu(p2, Vq, {"w": np.arange(Vq.num_dofs, dtype=np.float64)}, binary=binary)
    d2 = read_vtu(p2)
    q = d2["Points"][d2["connectivity"].reshape(-1, 4)][:, :, :2]
    area = 0.5 * np.abs(sum(q[:, k, 0] * q[:, (k + 1) % 4, 1] - q[:, (k + 1) % 4, 0] * q[:, k, 1] for k in range(4)))
    assert abs(area.sum() - 1.0) < 1e-12 and np.all(d2["types"] == 9) and len(area) == 265 * 16


# ---- second-order (9-node) quadrilaterals: the naive 2-D operator test's mesh_2 -----------------------
GOLD2B = os.path.join(os.path.dirname(__file__), "golden", "ref_test_operators2d_mesh2.npz")
REF_XDMF2B = "/root/reference/cpp/fenicsx-sf-naive/tests/test_operators2d/mesh_2/mesh.xdmf"


def _q2_mesh(curved=0.0, seed=0):
    """mesh_2 of the reference's 2-D operator test; curved > 0 moves the interior mid-edge and centre
    nodes by that fraction of the local cell size (the same offset for both cells sharing an edge),
    which makes the cells genuinely biquadratic."""
    from fenicsxfus_amd.unstructured import VTK_QUAD9_TO_TENSOR, QuadMesh
    g = np.load(GOLD2B)
    x = g["geometry"].copy()
    cells = g["topology_vtk"][:, VTK_QUAD9_TO_TENSOR]
    if curved:
        rng = np.random.default_rng(seed)
        corner = np.zeros(len(x), bool)
        corner[np.unique(cells[:, [0, 2, 6, 8]])] = True
        onb = (np.abs(x - 0.5).max(axis=1) > 0.5 - 1e-12)
        h = np.sqrt(1.0 / len(cells))
        move = ~corner & ~onb
        x[move] += curved * h * rng.uniform(-1, 1, (move.sum(), 2))
    mesh = QuadMesh(x, cells)
    return mesh, mesh.facet_tags(g["facet_topology"], g["facet_values"])


def test_second_order_quads_reader_and_oracle(orc, ref_mesh2d):
    mesh, tags = _q2_mesh()
    if os.path.exists(REF_XDMF2B):
        m2, cv, t2 = fa.read_xdmf_mesh(REF_XDMF2B)
        assert m2.order == 2 and np.array_equal(m2.geometry.dofmap, mesh.geometry.dofmap)
        assert len(cv) == 265 and np.array_equal(t2.cells, tags.cells)
    assert mesh.order == 2 and mesh.geometry.dofmap.shape == (265, 9) and len(tags.cells) == 60
    assert mesh.entity_counts() == (296, 560, 0, 265)
    V = HexFunctionSpace(mesh, 4)
    wts = orc.gll_weights_at(V.nodes1d)
    G, dJ = orc.geometry(2, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    # the file's second-order cells are straight-sided: same factors as the first-order twin mesh_1
    m1, _ = ref_mesh2d
    G1, dJ1 = orc.geometry(2, m1.geometry.x, m1.geometry.dofmap, V.nodes1d, wts)
    assert np.abs(G - G1).max() < 1e-11 and np.abs(dJ - dJ1).max() < 1e-14
    # curved cells: area and perimeter are still those of the unit square (interior nodes moved only)
    mc, tc = _q2_mesh(curved=0.05)
    Vc = HexFunctionSpace(mc, 5)
    wc, Dc = orc.gll_weights_at(Vc.nodes1d), orc.dphi(Vc.nodes1d)
    Gc, dJc = orc.geometry(2, mc.geometry.x, mc.geometry.dofmap, Vc.nodes1d, wc)
    assert np.abs(Gc - orc.geometry(2, m1.geometry.x, m1.geometry.dofmap, Vc.nodes1d, wc)[0]).max() > 1e-3
    assert abs(dJc.sum() - 1.0) < 1e-12 and dJc.min() > 0
    n, nc = Vc.num_dofs, mc.num_cells
    per = orc.facet_diag(2, tc.cells, tc.local_facets, np.ones(nc), mc.geometry.x, mc.geometry.dofmap, Vc.nodes1d, wc,
                         Vc.tensor_dofmap, n).sum()
    assert abs(per - 4.0) < 1e-12
    K = lambda v: orc.stiffness(2, 6, Vc.tensor_dofmap, Gc, Dc, np.ones(nc), v, np.zeros(n))  # noqa: E731
    assert np.abs(K(np.ones(n))).max() < 1e-12
    X = Vc.tabulate_dof_coordinates()
    for d in (0, 1):    # the coordinate functions are in the isoparametric space: energy = area
        assert abs(X[:, d] @ K(X[:, d].copy()) - 1.0) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("curved", [0.0, 0.05])
def test_second_order_quads_gpu(orc, curved):
    """The naive 2-D operator test with G = 2 (main.cpp:31) on its mesh_2, and on a curved variant:
    HIP operators, geometry and 10 Linear RK4 steps against the oracle."""
    mesh, tags = _q2_mesh(curved=curved)
    P = 4
    V = HexFunctionSpace(mesh, P)
    n, nc = V.num_dofs, mesh.num_cells
    wts, D = orc.gll_weights_at(V.nodes1d), orc.dphi(V.nodes1d)
    G, dJ = orc.geometry(2, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    ctx = fa.Context(0)
    d = fa.SpectralOperatorData(V, ctx)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    Gd, dJd = d.geometry()
    assert rel(Gd, G) < 1e-12 and rel(dJd, dJ) < 1e-13
    for kind, x, coef in _recipe2d(V, nc):
        if kind == "mass":
            assert rel(d.mass(x, coef, np.zeros(n)), orc.mass(2, P + 1, V.tensor_dofmap, dJ, coef, x, np.zeros(n))) < 1e-14
        else:
            assert rel(d.stiffness(x, coef, np.zeros(n)),
                       orc.stiffness(2, P + 1, V.tensor_dofmap, G, D, coef, x, np.zeros(n))) < 1e-12
    d.close()
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    m = orc.mass(2, P + 1, V.tensor_dofmap, dJ, 1 / (rho * c * c), np.ones(n), np.zeros(n))
    src = orc.facet_diag(2, tags.cells, tags.local_facets, 1 / rho, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d,
                         wts, V.tensor_dofmap, n)
    dt = 0.1 * np.sqrt(dJ.reshape(nc, -1).sum(axis=1)).min() / (1500.0 * P**2)
    uo, vo = np.zeros(n), np.zeros(n)
    orc.linear_rk4(2, P + 1, V.tensor_dofmap, G, D, -1 / rho, m, src, np.zeros(n), 5e3, 6e4, 1500.0, 0.0,
                   10 * dt * (1 - 1e-9), dt, uo, vo)
    model = fa.LinearSpectralExplicit(mesh, tags, P, c, rho, 5e3, 6e4, 1500.0, 4, dt, V=V, ctx=ctx)
    model.init()
    un, vn, _ = model.rk(0.0, 10 * dt * (1 - 1e-9))
    assert np.abs(uo).max() > 0 and rel(un.x.array, uo) < 1e-10 and rel(vn.x.array, vo) < 1e-10
    model.close()
    ctx.close()


def test_point_evaluation_quadrilaterals(ref_mesh2d):
    """evaluate() / locate() in 2-D: exact on an affine rectangle for a field of the space's degree,
    spectrally accurate on the reference's Gmsh quadrilateral mesh."""
    from fenicsxfus_amd.evaluate import evaluate, locate
    box = fa.BoxMesh([0, 0], [1.0, 0.8], (4, 3))
    Vb = fa.FunctionSpace(box, 4)
    Xd = Vb.tabulate_dof_coordinates()
    f = lambda X: X[:, 0] ** 4 - 2 * X[:, 1] ** 3 * X[:, 0] + X[:, 0] * X[:, 1] + 1.0  # noqa: E731
    rng = np.random.default_rng(0)
    pts = rng.uniform([0, 0], [1.0, 0.8], size=(200, 2))
    assert np.abs(evaluate(Vb, f(Xd), pts) - f(pts)).max() < 1e-13
    assert np.isnan(evaluate(Vb, f(Xd), np.array([[1.5, 0.1]]))).all()
    qm, _ = ref_mesh2d
    V = HexFunctionSpace(qm, 5)
    X = V.tabulate_dof_coordinates()
    u = np.sin(X[:, 0]) * np.cos(np.pi * X[:, 1])
    pts = rng.uniform(0.02, 0.98, size=(300, 2))
    cell, _ = locate(qm, pts)
    assert (cell >= 0).all()
    assert np.abs(evaluate(V, u, pts) - np.sin(pts[:, 0]) * np.cos(np.pi * pts[:, 1])).max() < 1e-6


def _write_hex27_xdmf(path, x, cells_tensor, facet_verts=None, facet_vals=None):
    """A Hexahedron_27 XDMF grid with inline (Format="XML") data, nodes per cell in VTK order -- what
    DOLFINx / meshio write for second-order hexahedra (here without the HDF5 side file)."""
    from fenicsxfus_amd.unstructured import VTK_HEX27_TO_TENSOR
    vtk = np.empty_like(cells_tensor)
    vtk[:, VTK_HEX27_TO_TENSOR] = cells_tensor               # tensor[k] = vtk[perm[k]]
    def block(a, kind):
        a = np.asarray(a)
        return (f'<DataItem Dimensions="{" ".join(str(k) for k in a.shape)}" NumberType="{kind}" Format="XML">'
                + " ".join(repr(v) for v in a.ravel().tolist()) + "</DataItem>")
    txt = ('<?xml version="1.0"?><Xdmf Version="3.0"><Domain><Grid Name="mesh" GridType="Uniform">'
           f'<Topology TopologyType="Hexahedron_27" NumberOfElements="{len(vtk)}" NodesPerElement="27">'
           + block(vtk, "Int") + '</Topology><Geometry GeometryType="XYZ">' + block(x, "Float") + "</Geometry></Grid>")
    if facet_verts is not None:
        txt += ('<Grid Name="mesh_facets" GridType="Uniform">'
                f'<Topology TopologyType="Quadrilateral_9" NumberOfElements="{len(facet_verts)}" NodesPerElement="9">'
                + block(facet_verts, "Int") + '</Topology><Attribute Name="f" AttributeType="Scalar" Center="Cell">'
                + block(np.asarray(facet_vals).reshape(-1, 1), "Int") + "</Attribute></Grid>")
    txt += "</Domain></Xdmf>"
    open(path, "w").write(txt)


def _curved_box(orc, n, P):
    return Problem(orc, n, P, hi=[0.012, 0.012, 0.008], order=2,
                   warp=lambda y: y + np.c_[20.0 * y[:, 1] ** 2, 15.0 * y[:, 2] ** 2, 0 * y[:, 0]])


def test_hexahedron_27_xdmf_reader(orc, tmp_path):
    """Second-order (27-node) hexahedra from an XDMF ``Hexahedron_27`` grid: VTK node order -> the tensor
    order libfusmi takes (the reference tabulates the coordinate element of any order,
    cpp/fenicsx-sf/common/precompute.hpp:52-55; its 27-node fixture has no data file), and the geometric
    GLL space on the curved cells."""
    from fenicsxfus_amd.unstructured import VTK_HEX27_TO_TENSOR, HexFunctionSpace, read_xdmf_mesh
    assert sorted(VTK_HEX27_TO_TENSOR.tolist()) == list(range(27))
    # corners, one edge, one face, centre: VTK position -> tensor position
    assert [int(np.nonzero(VTK_HEX27_TO_TENSOR == i)[0][0]) for i in (0, 2, 6, 9, 20, 25, 26)] == [0, 8, 26, 5, 12, 22, 13]
    pr = _curved_box(orc, (3, 2, 2), 3)
    x, cells = np.asarray(pr.mesh.geometry.x, np.float64), pr.mesh.geometry.dofmap
    # exterior facets of the x = 0 face as Quadrilateral_9 (corner vertices first), tagged 1
    c0 = np.arange(2 * 2)                                         # cells are x-major: the first layer
    fverts = np.array([[cells[c][n] for n in (0, 6, 24, 18, 3, 15, 21, 9, 12)] for c in c0])
    path = str(tmp_path / "hex27.xdmf")
    _write_hex27_xdmf(path, x, cells, fverts, np.ones(len(c0), int))
    mesh, cv, tags = read_xdmf_mesh(path)
    assert mesh.order == 2 and mesh.tdim == 3 and np.array_equal(mesh.geometry.dofmap, cells)
    assert np.array_equal(mesh.geometry.x, x) and cv is None
    assert len(tags.cells) == len(c0) and set(tags.local_facets.tolist()) == {2} and set(tags.values.tolist()) == {1}
    V = HexFunctionSpace(mesh, 3)
    assert V.num_dofs == pr.ndofs
    # the geometric dof matching on the curved cells reproduces the structured numbering up to a permutation
    G, dJ = orc.geometry(3, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, orc.gll_weights_at(V.nodes1d))
    assert np.abs(G - pr.G).max() < 1e-13 * np.abs(pr.G).max() and np.abs(dJ.sum() - pr.detJ.sum()) < 1e-15
    xg = np.random.default_rng(0).standard_normal(pr.ndofs)
    perm = np.empty(pr.ndofs, np.int64)
    perm[V.tensor_dofmap.ravel()] = pr.dm.ravel()                # unstructured dof -> structured dof
    y = orc.stiffness(3, 4, V.tensor_dofmap, G, pr.D, np.ones(len(cells)), xg[perm], np.zeros(pr.ndofs))
    assert np.abs(y - pr.K(xg)[perm]).max() < 1e-12 * np.abs(y).max()


@pytest.mark.gpu
def test_hexahedron_27_mesh_on_gpu(orc, tmp_path):
    from fenicsxfus_amd.unstructured import HexFunctionSpace, read_xdmf_mesh
    pr = _curved_box(orc, (3, 3, 2), 4)
    path = str(tmp_path / "hex27.xdmf")
    _write_hex27_xdmf(path, np.asarray(pr.mesh.geometry.x, np.float64), pr.mesh.geometry.dofmap)
    mesh, _, _ = read_xdmf_mesh(path)
    V = HexFunctionSpace(mesh, 4)
    wts = orc.gll_weights_at(V.nodes1d)
    G, dJ = orc.geometry(3, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    ctx = fa.Context(0)
    d = fa.SpectralOperatorData(V, ctx)
    assert d.geometry_mode() == "stream"                         # second-order geometry streams its factors
    rng = np.random.default_rng(1)
    x, coef = rng.standard_normal(V.num_dofs), rng.uniform(0.5, 2.0, mesh.num_cells)
    ref = orc.stiffness(3, 5, V.tensor_dofmap, G, orc.dphi(V.nodes1d), coef, x, np.zeros(V.num_dofs))
    y = d.stiffness(x, coef, np.zeros(V.num_dofs))
    assert np.abs(y - ref).max() < 1e-12 * np.abs(ref).max()
    refm = orc.mass(3, 5, V.tensor_dofmap, dJ, coef, x, np.zeros(V.num_dofs))
    assert np.abs(d.mass(x, coef, np.zeros(V.num_dofs)) - refm).max() < 1e-13 * np.abs(refm).max()
    d.close()
    ctx.close()


def test_hexahedron_27_hand_written_vtk_cell(tmp_path):
    """One unit-cube Hexahedron_27 cell whose 27 nodes are written out BY HAND in VTK order (vtkTriQuadraticHexahedron:
    corners 0-3 counter-clockwise on z = 0, 4-7 above them; mid-edge nodes 8-11 bottom ring, 12-15 top ring, 16-19
    the vertical edges 0-4, 1-5, 2-6, 3-7; face centres 20 x-, 21 x+, 22 y-, 23 y+, 24 z-, 25 z+; 26 body centre) --
    independent of VTK_HEX27_TO_TENSOR and of the writer used by the other tests.  After reading, node n = nx + 3 ny +
    9 nz of the cell must sit at (nx, ny, nz) / 2."""
    from fenicsxfus_amd.unstructured import read_xdmf_mesh
    vtk_xyz = [
        (0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1),          # 0-7
        (.5, 0, 0), (1, .5, 0), (.5, 1, 0), (0, .5, 0),                                                  # 8-11
        (.5, 0, 1), (1, .5, 1), (.5, 1, 1), (0, .5, 1),                                                  # 12-15
        (0, 0, .5), (1, 0, .5), (1, 1, .5), (0, 1, .5),                                                  # 16-19
        (0, .5, .5), (1, .5, .5), (.5, 0, .5), (.5, 1, .5), (.5, .5, 0), (.5, .5, 1),                    # 20-25
        (.5, .5, .5)]                                                                                    # 26
    # scramble the point numbering so that the cell's connectivity is not the identity
    ids = [13, 4, 22, 9, 0, 17, 26, 5, 11, 20, 2, 15, 24, 7, 18, 1, 10, 21, 3, 14, 25, 8, 19, 6, 16, 23, 12]
    pts = [None] * 27
    for k, pid in enumerate(ids):
        pts[pid] = vtk_xyz[k]
    xml = ('<?xml version="1.0"?><Xdmf Version="3.0"><Domain><Grid Name="mesh" GridType="Uniform">'
           '<Topology TopologyType="Hexahedron_27" NumberOfElements="1" NodesPerElement="27">'
           '<DataItem Dimensions="1 27" NumberType="Int" Format="XML">' + " ".join(str(i) for i in ids) + "</DataItem>"
           '</Topology><Geometry GeometryType="XYZ"><DataItem Dimensions="27 3" NumberType="Float" Format="XML">'
           + " ".join(f"{c}" for p in pts for c in p) + "</DataItem></Geometry></Grid></Domain></Xdmf>")
    path = tmp_path / "one_hex27.xdmf"
    path.write_text(xml)
    mesh, _, _ = read_xdmf_mesh(str(path))
    assert mesh.order == 2 and mesh.geometry.dofmap.shape == (1, 27)
    got = np.asarray(mesh.geometry.x)[mesh.geometry.dofmap[0]]
    want = np.array([[nx / 2, ny / 2, nz / 2] for nz in range(3) for ny in range(3) for nx in range(3)])
    assert np.array_equal(got, want)
