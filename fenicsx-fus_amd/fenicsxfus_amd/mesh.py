"""Light structured box / rectangle meshes exposing the DOLFINx attributes the adapter reads.

The reference builds its meshes with DOLFINx (``mesh::create_box`` --
cpp/fenicsx-sf/tests/test_operators3d/main.cpp:30-38 -- or XDMF files).  DOLFINx does not exist
in this environment, so this module supplies objects with the same duck-type surface
(``mesh.geometry.x``, ``mesh.geometry.dofmap``, ``mesh.topology.dim``,
``mesh.topology.index_map(d).size_local``, ``V.dofmap.list``, ``Function.x.array``,
``meshtags.indices/values``) that ``operators.py``/``models.py`` consume; real DOLFINx objects are
accepted by the same code.

Conventions: tensor vertex order v = vx + 2 vy + 4 vz; element-local tensor index
i = (i0*N + i1)*N + i2 with i0 <-> x; global DOF index x-slowest, so an x-slab partition owns a
contiguous DOF range and each interface plane is one contiguous index range.
"""
from __future__ import annotations

import numpy as np

from . import tables


class _IndexMap:
    def __init__(self, size_local, num_ghosts=0, size_global=None):
        self.size_local = int(size_local)
        self.num_ghosts = int(num_ghosts)
        self.size_global = int(size_global if size_global is not None else size_local)


class _Geometry:
    def __init__(self, x, dofmap, dim):
        self.x = x
        self.dofmap = dofmap
        self.dim = dim


class _Topology:
    def __init__(self, dim, ncells, ncells_global):
        self.dim = dim
        self._im = {dim: _IndexMap(ncells, 0, ncells_global)}

    def index_map(self, d):
        return self._im[d]


class BoxMesh:
    """``n`` cells per axis on [lo, hi]; tdim = len(n) in {2, 3}.

    ``rank``/``size`` select an x-slab of the cell grid (element-wise partition without ghost
    cells, the reference's ``GhostMode::none`` -- test_operators3d/main.cpp:31).
    """

    def __init__(self, lo, hi, n, rank: int = 0, size: int = 1, dtype=np.float64, perturb: float = 0.0,
                 seed: int = 0, order: int = 1, warp=None):
        self.tdim = len(n)
        assert self.tdim in (2, 3)
        self.order = order
        if order == 2:
            # second-order (27-node) hexahedra, geometry nodes in tensor order n = nx + 3 ny + 9 nz;
            # ``warp(x) -> x'`` (x: [nnodes, 3]) bends the node lattice (curved cells)
            assert self.tdim == 3 and not perturb
            self._init_second_order(lo, hi, n, rank, size, dtype, warp)
            return
        assert warp is None
        self.n = tuple(int(k) for k in n)
        self.lo = np.asarray(lo, dtype=np.float64)
        self.hi = np.asarray(hi, dtype=np.float64)
        self.rank, self.size = rank, size
        self.dtype = np.dtype(dtype)
        self.perturbed = bool(perturb)
        nx = self.n[0]
        assert size <= nx, "more ranks than element layers"
        # contiguous slabs of element layers along x
        self.cx0 = (nx * rank) // size
        self.cx1 = (nx * (rank + 1)) // size
        self.nloc = (self.cx1 - self.cx0,) + self.n[1:]
        nvert = [k + 1 for k in self.nloc]
        axes = []
        for d in range(self.tdim):
            h = (self.hi[d] - self.lo[d]) / self.n[d]
            off = self.cx0 if d == 0 else 0
            axes.append(self.lo[d] + h * (off + np.arange(nvert[d])))
        grid = np.meshgrid(*axes, indexing="ij")
        x = np.zeros((int(np.prod(nvert)), 3), dtype=np.float64)
        for d in range(self.tdim):
            x[:, d] = grid[d].ravel()
        if perturb:
            # deterministic, partition-independent interior-vertex perturbation (non-affine hexes)
            gi = self._global_vertex_ids(nvert)
            hmin = min((self.hi[d] - self.lo[d]) / self.n[d] for d in range(self.tdim))
            interior = np.ones(len(gi), dtype=bool)
            for dd in range(self.tdim):
                interior &= (np.abs(x[:, dd] - self.lo[dd]) > 1e-9 * hmin) & (
                    np.abs(x[:, dd] - self.hi[dd]) > 1e-9 * hmin)
            for d in range(self.tdim):
                r = np.sin(12.9898 * (gi + 1) + 78.233 * (d + 1) + seed) * 43758.5453
                r = r - np.floor(r) - 0.5
                x[:, d] += perturb * hmin * r * interior
        # cell -> vertex map, tensor vertex order
        cidx = np.indices(self.nloc).reshape(self.tdim, -1)
        nv = 1 << self.tdim
        dm = np.empty((cidx.shape[1], nv), dtype=np.int32)
        for v in range(nv):
            vid = np.zeros(cidx.shape[1], dtype=np.int64)
            for d in range(self.tdim):
                vid = vid * nvert[d] + cidx[d] + ((v >> d) & 1)
            dm[:, v] = vid
        self.geometry = _Geometry(x.astype(self.dtype), dm, self.tdim)
        ncg = int(np.prod(self.n))
        self.topology = _Topology(self.tdim, dm.shape[0], ncg)
        self._cidx = cidx

    def _init_second_order(self, lo, hi, n, rank, size, dtype, warp):
        self.n = tuple(int(k) for k in n)
        self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
        self.rank, self.size, self.dtype = rank, size, np.dtype(dtype)
        nx = self.n[0]
        self.cx0, self.cx1 = (nx * rank) // size, (nx * (rank + 1)) // size
        self.nloc = (self.cx1 - self.cx0,) + self.n[1:]
        nn = [2 * k + 1 for k in self.nloc]                     # node lattice, half-cell spacing
        axes = []
        for d in range(3):
            h = (self.hi[d] - self.lo[d]) / self.n[d]
            off = self.cx0 if d == 0 else 0
            axes.append(self.lo[d] + 0.5 * h * (2 * off + np.arange(nn[d])))
        grid = np.meshgrid(*axes, indexing="ij")
        x = np.stack([g.ravel() for g in grid], axis=1)
        if warp is not None:
            x = np.asarray(warp(x), dtype=np.float64)
        cidx = np.indices(self.nloc).reshape(3, -1)
        dm = np.empty((cidx.shape[1], 27), dtype=np.int32)
        for nz in range(3):
            for ny in range(3):
                for nxx in range(3):
                    nid = ((2 * cidx[0] + nxx) * nn[1] + 2 * cidx[1] + ny) * nn[2] + 2 * cidx[2] + nz
                    dm[:, nxx + 3 * ny + 9 * nz] = nid
        self.geometry = _Geometry(x.astype(self.dtype), dm, 3)
        self.topology = _Topology(3, dm.shape[0], int(np.prod(self.n)))
        self._cidx = cidx

    def _global_vertex_ids(self, nvert):
        idx = np.indices(nvert).reshape(self.tdim, -1).astype(np.int64)
        idx[0] += self.cx0
        gid = np.zeros(idx.shape[1], dtype=np.int64)
        for d in range(self.tdim):
            gid = gid * (self.n[d] + 1) + idx[d]
        return gid

    @property
    def num_cells(self):
        return self.geometry.dofmap.shape[0]

    def hmin(self):
        return float(min((self.hi[d] - self.lo[d]) / self.n[d] for d in range(self.tdim)))

    def cell_centroids(self):
        return self.geometry.x[self.geometry.dofmap].mean(axis=1)

    # ---- boundary facets as (cell, local facet) pairs -------------------------------------
    # DOLFINx local facet numbering: hex 0:z=0 1:y=0 2:x=0 3:x=1 4:y=1 5:z=1;
    # quad 0:y=0 1:x=0 2:x=1 3:y=1.
    def exterior_facets(self):
        """Return (cells, local_facets, axis, side) of the GLOBAL domain boundary in this slab."""
        t = self.tdim
        table = {3: [(2, 0), (1, 0), (0, 0), (0, 1), (1, 1), (2, 1)], 2: [(1, 0), (0, 0), (0, 1), (1, 1)]}[t]
        cells, lf, ax, sd = [], [], [], []
        ci = self._cidx
        for f, (a, s) in enumerate(table):
            off = self.cx0 if a == 0 else 0
            on = (ci[a] + off == (self.n[a] - 1 if s else 0))
            ids = np.nonzero(on)[0]
            cells.append(ids)
            lf.append(np.full(len(ids), f, dtype=np.int32))
            ax.append(np.full(len(ids), a, dtype=np.int32))
            sd.append(np.full(len(ids), s, dtype=np.int32))
        return (np.concatenate(cells).astype(np.int32), np.concatenate(lf), np.concatenate(ax),
                np.concatenate(sd))


class FacetTags:
    """Duck-type of ``dolfinx.mesh.MeshTags`` for boundary facets given as (cell, local facet)."""

    def __init__(self, cells, local_facets, values):
        self.cells = np.asarray(cells, dtype=np.int32)
        self.local_facets = np.asarray(local_facets, dtype=np.int32)
        self.values = np.asarray(values, dtype=np.int32)
        self.indices = np.arange(len(self.values), dtype=np.int32)

    def find(self, tag):
        return np.nonzero(self.values == tag)[0]


def tag_box_boundary(mesh: BoxMesh, source_axis: int = 0, source_side: int = 0) -> FacetTags:
    """Tag 1 on the source face (x = lo by default), tag 2 (absorbing) on every other face --
    the reference's convention (forms.py:36-39 of SC1-BM1)."""
    cells, lf, ax, sd = mesh.exterior_facets()
    vals = np.where((ax == source_axis) & (sd == source_side), 1, 2).astype(np.int32)
    return FacetTags(cells, lf, vals)


class _DofMap:
    def __init__(self, lst, index_map):
        self.list = lst
        self.index_map = index_map
        self.index_map_bs = 1


class FunctionSpace:
    """Degree-P GLL Lagrange space on a BoxMesh.  ``dofmap.list`` is emitted directly in tensor
    product order, i.e. what ``reorder_dofmap`` (permute.hpp:15-42) produces in the reference.

    ``node_order`` permutes the 1-D node/DOF order inside each element (default monotone);
    results per global DOF are invariant to it."""

    def __init__(self, mesh: BoxMesh, P: int, node_order=None):
        self.mesh, self.P = mesh, int(P)
        N = P + 1
        pts, _ = tables.gll(N)
        order = np.arange(N) if node_order is None else np.asarray(node_order)
        assert sorted(order.tolist()) == list(range(N))
        self.nodes1d = pts[order]           # nodes1d[i] = coordinate of local 1-D DOF i
        t = mesh.tdim
        nd = [k * P + 1 for k in mesh.nloc]  # local DOF grid
        ndg = [k * P + 1 for k in mesh.n]    # global DOF grid
        self.dof_grid, self.dof_grid_global = nd, ndg
        ci = mesh._cidx
        loc = np.indices((N,) * t).reshape(t, -1)
        ndofs = int(np.prod(nd))
        assert ndofs < 2**31, "local DOF indices are int32 (as in the reference's dofmap)"
        # dof(cell, local) = base(cell) + offset(local): the lexicographic index is linear in both parts, so the
        # [ncells, N^t] table is one int32 broadcast add (a 256^3 p=4 box is 8.4 GB of dofmap; no int64 temporaries)
        base = np.zeros(ci.shape[1], dtype=np.int64)
        off = np.zeros(N**t, dtype=np.int64)
        for d in range(t):
            base = base * nd[d] + ci[d] * P
            off = off * nd[d] + order[loc[d]]
        dm = np.empty((ci.shape[1], N**t), dtype=np.int32)
        np.add(base.astype(np.int32)[:, None], off.astype(np.int32)[None, :], out=dm)
        plane = int(np.prod(nd[1:]))
        self.plane = plane
        self.global_offset = mesh.cx0 * P * plane  # local dof l  <->  global dof l + offset
        self.dofmap = _DofMap(dm, _IndexMap(ndofs, 0, int(np.prod(ndg))))
        self.tensor_dofmap = self.dofmap.list
        # shared interface planes with the slab neighbours: (neighbour rank, local dof indices),
        # both sides list the plane in the same (global id) order
        self.neighbours = []
        if mesh.rank > 0:
            self.neighbours.append((mesh.rank - 1, np.arange(0, plane, dtype=np.int32)))
        if mesh.rank < mesh.size - 1:
            self.neighbours.append((mesh.rank + 1, np.arange(ndofs - plane, ndofs, dtype=np.int32)))

    @property
    def num_dofs(self):
        return self.dofmap.index_map.size_local

    def tabulate_dof_coordinates(self):
        m, P = self.mesh, self.P
        N = P + 1
        if getattr(m, "perturbed", False) or getattr(m, "order", 1) == 2:
            return self._mapped_dof_coordinates()
        pts, _ = tables.gll(N)
        axes = []
        for d in range(m.tdim):
            h = (m.hi[d] - m.lo[d]) / m.n[d]
            off = m.cx0 if d == 0 else 0
            k = np.arange(self.dof_grid[d])
            cell = np.minimum(k // P, m.nloc[d] - 1)
            axes.append(m.lo[d] + h * (off + cell + pts[k - cell * P]))
        grid = np.meshgrid(*axes, indexing="ij")
        x = np.zeros((self.num_dofs, 3))
        for d in range(m.tdim):
            x[:, d] = grid[d].ravel()
        return x


    def _mapped_dof_coordinates(self):
        """Physical positions of the dofs through the cells' geometry map (perturbed / curved meshes):
        tensor Lagrange basis of the geometry order at the element's GLL nodes."""
        m, t = self.mesh, self.mesh.tdim
        g = getattr(m, "order", 1)
        x = np.asarray(m.geometry.x, dtype=np.float64)
        cd = x[m.geometry.dofmap]                                      # [nc, (g+1)^t, 3]
        grids = np.meshgrid(*([np.asarray(self.nodes1d)] * t), indexing="ij")
        X = np.stack([gr.ravel() for gr in grids], axis=1)             # element node i -> reference point

        def basis1d(xv):
            if g == 1:
                return np.stack([1.0 - xv, xv], axis=1)
            return np.stack([(2 * xv - 1) * (xv - 1), 4 * xv * (1 - xv), xv * (2 * xv - 1)], axis=1)

        phi = np.ones((X.shape[0], (g + 1) ** t))
        for n in range((g + 1) ** t):
            for d in range(t):
                phi[:, n] *= basis1d(X[:, d])[:, (n // (g + 1) ** d) % (g + 1)]
        out = np.zeros((self.num_dofs, 3))
        out[self.tensor_dofmap] = np.einsum("qn,cnk->cqk", phi, cd)
        return out


class _Vec:
    def __init__(self, n, dtype):
        self.array = np.zeros(n, dtype=dtype)


class Function:
    """Duck-type of ``dolfinx.fem.Function``: ``.x.array`` is the DOF vector."""

    def __init__(self, V, dtype=None):
        self.function_space = V
        n = V.num_dofs if hasattr(V, "num_dofs") else int(V)
        self.x = _Vec(n, dtype or getattr(getattr(V, "mesh", None), "dtype", np.float64))

    def interpolate(self, f):
        X = self.function_space.tabulate_dof_coordinates()
        self.x.array[:] = f(X.T)


class CellFunction:
    """DG0 per-cell field (``c0``, ``rho0`` in the reference drivers, BM7-SC1/main.cpp:81-109)."""

    def __init__(self, mesh, value=0.0, dtype=None):
        self.mesh = mesh
        self.x = _Vec(mesh.num_cells, dtype or mesh.dtype)
        self.x.array[:] = value
