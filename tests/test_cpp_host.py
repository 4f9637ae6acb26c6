"""The C++ host side above the C ABI (include/fusmi.hpp: StiffnessSpectral3D, MassSpectral3D,
Linear/Lossy/WesterveltSpectral3D with the reference's names and call semantics).
CPU: the header compiles with g++ -std=c++17 and links against libfusmi only; without a device the
example reports FUS_ERR_HIP (no CPU fallback).  GPU: the example's results against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import fenicsxfus_amd as fa
from fenicsxfus_amd import tag_box_boundary
from util import Problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "fenicsx-fus_amd", "fenicsxfus_amd")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "cpp_model_run"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cpp_model_run.cpp"), "-L", LIBDIR, "-lfusmi",
                           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)])
    return str(out)


def write_input(path, pr, tags, kind, nsteps, c, rho, delta, beta, f0, p0, s0, dt, x, coeffs):
    m = pr.mesh
    with open(path, "wb") as f:
        np.array([pr.tdim, pr.P, m.num_cells, pr.ndofs, m.geometry.x.shape[0], len(tags.cells), kind, nsteps],
                 dtype=np.int64).tofile(f)
        np.array([f0, p0, s0, dt], dtype=np.float64).tofile(f)
        pr.dm.astype(np.int32).tofile(f)
        np.asarray(pr.nodes, dtype=np.float64).tofile(f)
        np.ascontiguousarray(m.geometry.x, dtype=np.float64).tofile(f)
        np.ascontiguousarray(m.geometry.dofmap, dtype=np.int32).tofile(f)
        for a in (tags.cells, tags.local_facets, tags.values):
            np.ascontiguousarray(a, dtype=np.int32).tofile(f)
        for a in (c, rho, delta, beta, x, coeffs):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)


def test_cpp_header_builds_and_fails_loudly_without_device(orc, exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present: covered by the gpu test")
    pr = Problem(orc, (2, 2, 2), 2)
    nc = pr.mesh.num_cells
    one = np.ones(nc)
    write_input(tmp_path / "in.bin", pr, tag_box_boundary(pr.mesh), 0, 1, 1500 * one, 1000 * one, one, one, 0.5e6,
                6e4, 1500.0, 1e-8, np.ones(pr.ndofs), one)
    out = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 3 and "fusmi error -2" in out.stderr    # FUS_ERR_HIP, nothing computed


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,P", [(0, (5, 4, 3), 4), (1, (4, 3, 3), 3), (2, (4, 4, 3), 4), (0, (9, 7), 4)])
def test_cpp_host_classes_vs_oracle(orc, exe, tmp_path, kind, n, P):
    t = len(n)
    L = 0.012
    pr = Problem(orc, n, P, hi=[L] * t, perturb=0.1)
    nc = pr.mesh.num_cells
    cx = pr.mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * L) & (cx < 0.6 * L)
    c, rho = np.where(sel, 2800.0, 1500.0), np.where(sel, 1850.0, 1000.0)
    f0, s0 = 0.5e6, 1500.0
    p0 = 6e6 if kind == 2 else 6e4
    w0 = 2 * np.pi * f0
    delta = np.where(sel, fa.compute_diffusivity_of_sound(w0, 2800.0, 400.0 / 20.0 * np.log(10.0)),
                     fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    beta = np.where(sel, 6.0, 3.5)
    tags = tag_box_boundary(pr.mesh)
    rng = np.random.default_rng(kind)
    x, coeffs = rng.standard_normal(pr.ndofs), rng.uniform(0.5, 2.0, nc)
    dt = 0.5 * (L / n[0]) / (c.max() * P**2)
    nsteps = 10
    write_input(tmp_path / "in.bin", pr, tags, kind, nsteps, c, rho, delta, beta, f0, p0, s0, dt, x, coeffs)
    out = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.float64)
    nd = pr.ndofs
    ys, ym, u, v = raw[:nd], raw[nd:2 * nd], raw[2 * nd:3 * nd], raw[3 * nd:4 * nd]
    taken, ndofs = np.frombuffer(raw[4 * nd:].tobytes(), dtype=np.int64)
    assert taken == nsteps and ndofs == nd
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    assert rel(ys, 1.0 + pr.K(x, coeffs)) < 1e-12 and rel(ym, 1.0 + pr.M(x, coeffs)) < 1e-14
    tf = nsteps * dt * (1 - 1e-9)
    # the mesh size and norm the examples' mains report (linear_planewave2d_1/main.cpp:60-68, 151-157)
    words = out.stdout.split()
    hmin, l2 = float(words[words.index("hmin") + 1]), float(words[words.index("norm") + 1])
    X, gdm = pr.mesh.geometry.x, pr.mesh.geometry.dofmap
    assert abs(hmin - min(max(np.linalg.norm(X[a] - X[b]) for a in cell for b in cell) for cell in gdm)) < 1e-15
    assert abs(l2 - np.sqrt(u @ pr.M(u))) < 1e-12 * l2
    uo, vo = np.zeros(nd), np.zeros(nd)
    if kind == 0:
        m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
        orc.linear_rk4(t, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, f0, p0, s0, 0.0, tf, dt, uo, vo)
    else:
        m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
        if kind == 1:
            orc.lossy_rk4(t, pr.N, pr.dm, pr.G, pr.D, lin, att, m, src, absb, src2, f0, p0, s0, 0.0, tf, dt, uo, vo)
        else:
            n1 = -2.0 * beta / rho**2 / c**4
            orc.westervelt_rk4(t, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, f0, p0,
                               s0, 0.0, tf, dt, uo, vo)
    assert np.abs(uo).max() > 0 and rel(u, uo) < 1e-10 and rel(v, vo) < 1e-10
