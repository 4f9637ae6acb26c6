"""Point evaluation of a degree-P GLL field on first-order hexahedra or quadrilaterals
(post-processing, host side).

The reference samples its solutions on lines/planes with DOLFINx's ``Function.eval`` after locating
the cells (``compute_eval_params``, python/src/fenicsxfus/utils.py:10-47;
cpp/mwe/parallel_eval_line/main.cpp:49-84).  Here: candidate cells from a KD-tree on cell centroids,
Newton inversion of the trilinear map, tensor-product Lagrange interpolation with barycentric
weights on the element's GLL nodes.  Works with ``BoxMesh``/``HexMesh`` + their function spaces (or
any object exposing ``mesh.geometry.x/.dofmap``, ``tensor_dofmap``, ``nodes1d``)."""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def _shape(X):
    """Multilinear shape functions and their reference gradients at X [n,t] (t = 2 | 3) ->
    phi [n,2^t], dphi [n,2^t,t]."""
    n, t = X.shape
    nv = 1 << t
    phi = np.ones((n, nv))
    dphi = np.ones((n, nv, t))
    for v in range(nv):
        for d in range(t):
            bit = (v >> d) & 1
            f = X[:, d] if bit else 1.0 - X[:, d]
            phi[:, v] *= f
            for e in range(t):
                dphi[:, v, e] *= (1.0 if bit else -1.0) if e == d else f
    return phi, dphi


def locate(mesh, points, ncand: int = 32, tol: float = 1e-10):
    """For each point: (cell, reference coordinates) of a cell containing it, cell = -1 if none."""
    t = mesh.topology.dim
    cells = np.asarray(mesh.geometry.dofmap)
    if cells.shape[1] != (1 << t):
        raise NotImplementedError("point location needs first-order cells")
    pts = np.atleast_2d(np.asarray(points, dtype=np.float64))[:, :t]
    x = np.asarray(mesh.geometry.x, dtype=np.float64)[:, :t]
    cen = x[cells].mean(axis=1)
    _, cand = cKDTree(cen).query(pts, k=min(ncand, len(cen)))
    cand = cand.reshape(len(pts), -1)
    out_cell = np.full(len(pts), -1, dtype=np.int64)
    out_X = np.zeros((len(pts), t))
    todo = np.arange(len(pts))
    for k in range(cand.shape[1]):
        if len(todo) == 0:
            break
        c = cand[todo, k]
        cd = x[cells[c]]                                   # [m, 2^t, t]
        X = np.full((len(todo), t), 0.5)
        for _ in range(25):                                # Newton on x(X) = p
            phi, dphi = _shape(X)
            r = np.einsum("mv,mvi->mi", phi, cd) - pts[todo]
            J = np.einsum("mvi,mvj->mij", cd, dphi)
            X = X - np.linalg.solve(J, r[..., None])[..., 0]
        phi, _ = _shape(X)
        ok = (np.abs(np.einsum("mv,mvi->mi", phi, cd) - pts[todo]).max(axis=1) < 1e-9 * np.ptp(x, axis=0).max()) \
            & np.all((X > -tol) & (X < 1 + tol), axis=1)
        out_cell[todo[ok]] = c[ok]
        out_X[todo[ok]] = X[ok]
        todo = todo[~ok]
    return out_cell, out_X


def evaluate(V, u, points):
    """u_h(points) for the DOF vector ``u`` (or an object with ``.x.array``) of space ``V``; NaN
    outside the mesh."""
    ua = np.asarray(getattr(getattr(u, "x", None), "array", u), dtype=np.float64)
    cell, X = locate(V.mesh, points)
    nodes = np.asarray(V.nodes1d, dtype=np.float64)
    N = len(nodes)
    diff = nodes[:, None] - nodes[None, :]
    np.fill_diagonal(diff, 1.0)
    lam = 1.0 / diff.prod(axis=1)                          # barycentric weights

    def basis(t):                                          # Lagrange basis values at t [m] -> [m, N]
        d = t[:, None] - nodes[None, :]
        exact = np.abs(d) < 1e-14
        d[exact] = 1.0
        w = lam[None, :] / d
        b = w / w.sum(axis=1, keepdims=True)
        hit = exact.any(axis=1)
        b[hit] = exact[hit].astype(float)
        return b

    out = np.full(len(cell), np.nan)
    ok = cell >= 0
    if X.shape[1] == 3:
        b0, b1, b2 = basis(X[ok, 0]), basis(X[ok, 1]), basis(X[ok, 2])
        dofs = np.asarray(V.tensor_dofmap)[cell[ok]].reshape(-1, N, N, N)
        out[ok] = np.einsum("mi,mj,mk,mijk->m", b0, b1, b2, ua[dofs])
    else:
        b0, b1 = basis(X[ok, 0]), basis(X[ok, 1])
        dofs = np.asarray(V.tensor_dofmap)[cell[ok]].reshape(-1, N, N)
        out[ok] = np.einsum("mi,mj,mij->m", b0, b1, ua[dofs])
    return out
