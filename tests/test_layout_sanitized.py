"""The host-side layout builder (csrc/layout.cpp: block partition, conflict-free rounds, internal DOF
numbering, interface-first block order) under AddressSanitizer + UBSan on the CPU (GPU sanitizers are
not available on the MI355X pool): structured / perturbed / unstructured, hexahedra and
quadrilaterals, with and without an interface mask, odd block sizes."""
import os
import subprocess

import numpy as np
import pytest

import fenicsxfus_amd as fa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fenicsx-fus_amd", "csrc")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("san") / "layout_driver"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", CSRC, os.path.join(ROOT, "tests", "cpp", "layout_driver.cpp"),
                           os.path.join(CSRC, "layout.cpp"), "-o", str(exe)])
    return str(exe)


def run(driver, path, tdim, P, dm, cen, be, waves, mask=None):
    cen = np.asarray(cen, dtype=np.float64)
    if cen.shape[1] == 2:
        cen = np.hstack([cen, np.zeros((len(cen), 1))])
    with open(path, "wb") as f:
        np.array([tdim, P, dm.shape[0], int(dm.max()) + 1, be, waves, int(mask is not None)], dtype=np.int64).tofile(f)
        np.ascontiguousarray(dm, dtype=np.int32).tofile(f)
        np.ascontiguousarray(cen).tofile(f)
        if mask is not None:
            np.ascontiguousarray(mask, dtype=np.uint8).tofile(f)
    out = subprocess.run([driver, str(path)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
    return dict(kv.split("=") for kv in out.stdout.split()[1:])


@pytest.mark.parametrize("n,P,be,waves", [((5, 4, 3), 2, 7, 1), ((4, 4, 4), 4, 32, 4), ((3, 2, 2), 7, 5, 2),
                                          ((9, 7), 4, 16, 4), ((6, 5), 7, 64, 8), ((2, 2, 2), 3, 1000, 4)])
def test_layout_builder_clean_under_sanitizers(driver, tmp_path, n, P, be, waves):
    m = fa.BoxMesh([0.0] * len(n), [1.0] * len(n), n, perturb=0.1)
    V = fa.FunctionSpace(m, P)
    r = run(driver, tmp_path / "in.bin", len(n), P, V.tensor_dofmap, m.cell_centroids(), be, waves)
    assert int(r["blocks"]) == -(-m.num_cells // be) and int(r["if"]) == 0


def test_layout_builder_interface_mask_and_unstructured(driver, tmp_path):
    m = fa.BoxMesh([0, 0, 0], [2, 1, 1], (8, 4, 4), rank=1, size=3)          # middle slab: two interface planes
    V = fa.FunctionSpace(m, 3)
    mask = np.zeros(V.num_dofs, bool)
    for _, idx in V.neighbours:
        mask[idx] = True
    r = run(driver, tmp_path / "a.bin", 3, 3, V.tensor_dofmap, m.cell_centroids(), 8, 4, mask)
    assert 0 < int(r["if"]) <= int(r["blocks"])
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_operators2d_mesh.npz"))
    from fenicsxfus_amd.unstructured import VTK_QUAD_TO_TENSOR, QuadMesh
    qm = QuadMesh(g["geometry"], g["topology_vtk"][:, VTK_QUAD_TO_TENSOR])
    Vq = fa.HexFunctionSpace(qm, 5)
    r = run(driver, tmp_path / "b.bin", 2, 5, Vq.tensor_dofmap, qm.cell_centroids(), 37, 4)
    assert int(r["interior"]) + int(r["shared"]) == Vq.num_dofs
