"""CPU-only checks of the C-ABI library and its host logic (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fusmi.h")).read()
    declared = set(re.findall(r"\b(fus_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)
    L = _abi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.fus_version() == 1


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fa.FusError, match="no HIP device"):
        fa.Context(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "fenicsx-fus_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "liboracle" not in src and "oracle/" not in src, f


@pytest.mark.parametrize("n,P,be,w", [((8, 8, 8), 4, 64, 4), ((5, 6, 7), 3, 16, 2), ((3, 3, 3), 2, 5, 1),
                                     ((4, 4, 4), 7, 8, 4), ((6, 4, 2), 5, 64, 4)])
def test_layout_invariants(n, P, be, w):
    # fus_layout_check builds the block partition / rounds / numbering and runs verify_layout:
    # every cell in one block, rounds conflict-free, dof_perm injective, local dofmaps consistent
    m = fa.BoxMesh([0, 0, 0], [1, 1, 1], n, perturb=0.1)
    V = fa.FunctionSpace(m, P)
    info = fa.layout_check(P, V.tensor_dofmap, m.cell_centroids(), block_elems=be, waves=w)
    nblocks, nint, nsh, npairs, maxloc, nshapes, lds, nintl = info
    assert nint + nsh == V.num_dofs
    assert nblocks == -(-m.num_cells // be)
    assert npairs >= 2 * nsh or nblocks == 1
    assert nintl >= V.num_dofs and nintl % 16 == 0


def test_layout_structured_box_is_cubic_blocks():
    # 8^3 cells, 64 per block -> eight 4x4x4 blocks: 17^3 local dofs
    m = fa.BoxMesh([0, 0, 0], [1, 1, 1], (8, 8, 8))
    V = fa.FunctionSpace(m, 4)
    info = fa.layout_check(4, V.tensor_dofmap, m.cell_centroids())
    assert info[0] == 8 and info[4] == 17**3
    # 16^3: 64 blocks but only 27 distinct shapes (3 position classes per axis)
    m = fa.BoxMesh([0, 0, 0], [1, 1, 1], (16, 16, 16))
    V = fa.FunctionSpace(m, 4)
    info = fa.layout_check(4, V.tensor_dofmap, m.cell_centroids())
    assert info[0] == 64 and info[5] == 27


@pytest.mark.parametrize("n,P,be", [((9, 7), 4, 16), ((23, 17), 2, 100), ((5, 4), 7, 6)])
def test_layout_invariants_quadrilaterals(n, P, be):
    m = fa.BoxMesh([0, 0], [1.5, 1.0], n, perturb=0.1)
    V = fa.FunctionSpace(m, P)
    nblocks, nint, nsh, npairs, maxloc, nshapes, lds, nintl = fa.layout_check(P, V.tensor_dofmap, m.cell_centroids(),
                                                                               block_elems=be, waves=2)
    assert nint + nsh == V.num_dofs and nblocks == -(-m.num_cells // be) and nintl % 16 == 0


def test_layout_interface_mask_forces_shared():
    """Multi-rank layout: dofs on the slab's interface plane (held by the neighbour rank too) are
    classified shared even where a single local block touches them."""
    m = fa.BoxMesh([0, 0, 0], [2, 1, 1], (8, 4, 4), rank=0, size=2)
    V = fa.FunctionSpace(m, 3)
    mask = np.zeros(V.num_dofs, bool)
    for _, idx in V.neighbours:
        mask[idx] = True
    assert mask.sum() == (4 * 3 + 1) ** 2
    base = fa.layout_check(3, V.tensor_dofmap, m.cell_centroids(), block_elems=16)
    forced = fa.layout_check(3, V.tensor_dofmap, m.cell_centroids(), block_elems=16, force_shared=mask)
    assert base[1] + base[2] == V.num_dofs
    # shared count includes the alignment gap before the interface range, hence >=
    assert forced[1] < base[1] and forced[1] + forced[2] >= V.num_dofs and forced[0] == base[0]


def test_layout_rejects_bad_dofmap():
    m = fa.BoxMesh([0, 0, 0], [1, 1, 1], (2, 2, 2))
    V = fa.FunctionSpace(m, 2)
    dm = V.tensor_dofmap.copy()
    dm[0, 0] = -1
    out = (C.c_int64 * 8)()
    cen = np.ascontiguousarray(m.cell_centroids())
    rc = _abi.lib().fus_layout_check(2, C.c_int64(8), C.c_int64(V.num_dofs), _abi.ptr(dm), _abi.ptr(cen), 64, 4, out)
    assert rc == -1 and b"out of range" in _abi.lib().fus_last_error()


def test_python_tables_match_oracle(orc):
    for N in range(3, 9):
        p, w = fa.tables.gll(N)
        po, wo = orc.gll(N)
        assert np.allclose(p, po, atol=1e-15) and np.allclose(w, wo, atol=1e-15)
        assert np.allclose(fa.tables.dphi(p), orc.dphi(po), atol=1e-11)


def test_plain_c_consumer_builds_and_runs(tmp_path):
    """include/fusmi.h is consumable from plain C and the library links without torch/Python:
    examples/cabi_min.c runs the host-only layout check (and the device path where a GPU exists)."""
    import subprocess

    exe = tmp_path / "cabi_min"
    libdir = os.path.join(ROOT, "fenicsx-fus_amd", "fenicsxfus_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cabi_min.c"), "-L", libdir, "-lfusmi", "-lm",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "layout ok" in out.stdout
