"""Bridge from real DOLFINx objects to the inputs libfusmi takes.

DOLFINx/Basix are not installed where this repository is built and tested, so this module is
import-guarded and NOT exercised by the test-suite; it restates, call for call, what the
reference does with the same objects:

* tensor-product dofmap: ``reorder_dofmap`` (cpp/fenicsx-sf/common/permute.hpp:15-42) --
  ``perm = argsort(basix.tp_dof_ordering(P, hexahedron, gll_warped))`` and
  ``tensor_dofmap[c, i] = dofmap[c, perm[i]]``;
* 1-D node coordinates in Basix's own order: the GLL quadrature points of the interval
  (spectral_op.hpp:160-162 uses the same rule through ``make_quadrature``); the library accepts
  any node order, so no assumption about that order is made here;
* boundary facets as (cell, local facet) pairs per tag: what
  ``fem::compute_integration_domains(exterior_facet, topology, ft->find(tag), tdim-1)`` returns
  (cpp/fenicsx-sf/common/Linear.hpp:113-118);
* shared DOFs per neighbour rank from the function space's ``IndexMap`` (the data behind
  ``la::Vector::scatter_fwd/scatter_rev``, Linear.hpp:196-206).

Usage where DOLFINx exists::

    from fenicsxfus_amd.dolfinx_adapter import wrap_function_space, wrap_facet_tags
    V = wrap_function_space(dolfinx_V, degree)
    tags = wrap_facet_tags(mesh, mt_facet)
    model = LinearSpectralExplicit(mesh, tags, degree, c0, rho0, f0, p0, s0, 4, dt, V=V)
    u_n, v_n, t = model.rk(t0, tf)            # u_n.x.array is in DOLFINx's own DOF numbering
"""
from __future__ import annotations

import numpy as np

# quadrature degree per polynomial degree, as in the reference (spectral_op.hpp:35-44, _linear.py:333-343)
QDEGREE = {2: 3, 3: 4, 4: 6, 5: 8, 6: 10, 7: 12, 8: 14, 9: 16, 10: 18}


def _require():
    try:
        import basix
        import dolfinx
    except ImportError as e:  # pragma: no cover - environment without DOLFINx
        raise ImportError("fenicsxfus_amd.dolfinx_adapter needs dolfinx and basix") from e
    return basix, dolfinx


class _WrappedSpace:
    """Duck-type consumed by ``SpectralOperatorData``: ``mesh``, ``P``, ``tensor_dofmap``,
    ``nodes1d``, ``dofmap`` (DOLFINx's own, for ``index_map``), ``neighbours``."""

    def __init__(self, V, P, tensor_dofmap, nodes1d, neighbours):
        self.mesh, self.P = V.mesh, P
        self.dofmap = V.dofmap
        self.tensor_dofmap = tensor_dofmap
        self.nodes1d = nodes1d
        self.neighbours = neighbours
        self._V = V

    @property
    def num_dofs(self):
        im = self.dofmap.index_map
        return im.size_local + im.num_ghosts


def wrap_function_space(V, P: int):
    """``V``: a ``dolfinx.fem.FunctionSpace`` of degree-P ``gll_warped`` Lagrange on hexahedra or
    quadrilaterals."""
    basix, dolfinx = _require()
    cell = basix.CellType.hexahedron if V.mesh.topology.dim == 3 else basix.CellType.quadrilateral
    tp_order = np.asarray(basix.tp_dof_ordering(basix.ElementFamily.P, cell, P,
                                                basix.LagrangeVariant.gll_warped, basix.DPCVariant.unset, False))
    perm = np.argsort(tp_order, kind="stable")                    # permute.hpp:27-32
    dm = np.asarray(V.dofmap.list).reshape(-1, len(tp_order))
    ncells = V.mesh.topology.index_map(V.mesh.topology.dim).size_local
    tensor_dofmap = np.ascontiguousarray(dm[:ncells][:, perm], dtype=np.int32)   # permute.hpp:38-41
    pts, _ = basix.make_quadrature(basix.CellType.interval, QDEGREE[P], basix.QuadratureType.gll)
    nodes1d = np.ascontiguousarray(np.asarray(pts).reshape(-1), dtype=np.float64)
    return _WrappedSpace(V, P, tensor_dofmap, nodes1d, _neighbours(V))


def _neighbours(V):
    """(rank, local dof indices) per neighbour rank, each list ordered by global index so both sides
    agree: ghosts owned by rank r are shared with r; owned dofs that are ghosts elsewhere are
    found through the index map's shared-index data."""
    im = V.dofmap.index_map
    if im.num_ghosts == 0 and im.size_global == im.size_local:
        return []
    n_owned = im.size_local
    shared = {}
    owners = np.asarray(im.owners)
    ghosts = np.asarray(im.ghosts)
    for k, (g, r) in enumerate(zip(ghosts, owners)):            # my ghosts: shared with their owner
        shared.setdefault(int(r), []).append((int(g), n_owned + k))
    # my owned dofs that other ranks ghost: index_to_dest_ranks gives, per owned index, the ranks
    dest = im.index_to_dest_ranks()
    offs, ranks = np.asarray(dest.offsets), np.asarray(dest.array)
    lo = im.local_range[0]
    for i in range(n_owned):
        for r in ranks[offs[i]:offs[i + 1]]:
            shared.setdefault(int(r), []).append((lo + i, i))
    out = []
    for r in sorted(shared):
        pairs = sorted(set(shared[r]))
        out.append((r, np.array([p[1] for p in pairs], dtype=np.int32)))
    return out


def wrap_facet_tags(mesh, meshtags):
    """DOLFINx ``MeshTags`` on facets -> object with ``cells``, ``local_facets``, ``values``
    ((cell, local facet) pairs of the tagged exterior facets, Linear.hpp:113-118)."""
    basix, dolfinx = _require()
    from dolfinx import fem

    tdim = mesh.topology.dim
    mesh.topology.create_connectivity(tdim - 1, tdim)
    cells, lfs, vals = [], [], []
    for tag in np.unique(meshtags.values):
        ents = meshtags.find(tag)
        dom = np.asarray(fem.compute_integration_domains(fem.IntegralType.exterior_facet, mesh.topology, ents,
                                                         tdim - 1)).reshape(-1, 2)
        cells.append(dom[:, 0]), lfs.append(dom[:, 1]), vals.append(np.full(len(dom), tag))
    from .mesh import FacetTags

    return FacetTags(np.concatenate(cells), np.concatenate(lfs), np.concatenate(vals))


# Second-order (27-node) hexahedron: DOLFINx/Basix store geometry nodes as vertices, edges, faces,
# interior, following the reference-cell topology (vertex v = vx + 2 vy + 4 vz; edges and faces in
# lexicographic vertex order).  libfusmi takes them in tensor order n = nx + 3 ny + 9 nz.
# NOT verifiable here (no Basix): check `perm` against `basix.geometry/topology(CellType.hexahedron)`
# where DOLFINx exists before relying on it.
_HEX_EDGES = [(0, 1), (0, 2), (0, 4), (1, 3), (1, 5), (2, 3), (2, 6), (3, 7), (4, 5), (4, 6), (5, 7), (6, 7)]
_HEX_FACES = [(0, 1, 2, 3), (0, 1, 4, 5), (0, 2, 4, 6), (1, 3, 5, 7), (2, 3, 6, 7), (4, 5, 6, 7)]


def hex27_dolfinx_to_tensor():
    """perm with tensor_nodes[:, perm[k]] = dolfinx_nodes[:, k]: position (in tensor order) of the
    k-th DOLFINx geometry node of a second-order hexahedron."""
    def pos(v):
        return np.array([v & 1, (v >> 1) & 1, v >> 2]) * 2
    nodes = [pos(v) for v in range(8)]
    nodes += [(pos(a) + pos(b)) // 2 for a, b in _HEX_EDGES]
    nodes += [sum(pos(v) for v in f) // 4 for f in _HEX_FACES]
    nodes.append(np.array([1, 1, 1]))
    return np.array([n[0] + 3 * n[1] + 9 * n[2] for n in nodes], dtype=np.int64)


_QUAD_EDGES = [(0, 1), (0, 2), (1, 3), (2, 3)]


def quad9_dolfinx_to_tensor():
    """The same for the 9-node quadrilateral (vertices, edges, interior -> n = nx + 3 ny)."""
    def pos(v):
        return np.array([v & 1, v >> 1]) * 2
    nodes = [pos(v) for v in range(4)]
    nodes += [(pos(a) + pos(b)) // 2 for a, b in _QUAD_EDGES]
    nodes.append(np.array([1, 1]))
    return np.array([n[0] + 3 * n[1] for n in nodes], dtype=np.int64)


def tensor_geometry_dofmap(mesh):
    """Geometry dofmap of a DOLFINx mesh in the order libfusmi expects (order 1: unchanged)."""
    gd = np.asarray(mesh.geometry.dofmap)
    tdim = mesh.topology.dim
    if gd.shape[1] == (1 << tdim):
        return gd.astype(np.int32)
    if gd.shape[1] != 3 ** tdim:
        raise ValueError("only first- and second-order hexahedral / quadrilateral geometry is supported")
    perm = hex27_dolfinx_to_tensor() if tdim == 3 else quad9_dolfinx_to_tensor()
    out = np.empty_like(gd, dtype=np.int32)
    out[:, perm] = gd
    return out
