"""Bridge from DOLFINx objects to the inputs libfusmi takes.

DOLFINx/Basix are not installed where this repository is built and tested, so the module imports
them lazily; its logic is exercised by tests/test_dolfinx_adapter.py with duck-typed stand-ins for
the handful of attributes it reads (``V.dofmap.list``, ``V.dofmap.index_map.{size_local, num_ghosts,
size_global, local_range, ghosts, owners, index_to_dest_ranks()}``, ``basix.tp_dof_ordering``,
``basix.make_quadrature``).  It restates, call for call, what the reference does with the same objects:

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
  ``la::Vector::scatter_fwd/scatter_rev``, Linear.hpp:196-206).  The library's exchange is
  symmetric -- every rank that holds a DOF adds the partial sums of ALL the other holders, in
  ascending rank order -- while an ``IndexMap`` only links a ghost to its owner: two ranks that both
  ghost a DOF owned by a third (partition edges and corners) do not know of each other.  The owner
  therefore tells every ghosting rank which other ranks hold the DOF (one all-to-all at setup,
  :func:`sharer_messages` / :func:`neighbours_from_index_map`).

Usage where DOLFINx exists::

    from fenicsxfus_amd.dolfinx_adapter import wrap_function_space, wrap_facet_tags
    V = wrap_function_space(dolfinx_V, degree)
    tags = wrap_facet_tags(mesh, mt_facet)
    model = LinearSpectralExplicit(mesh, tags, degree, c0, rho0, f0, p0, s0, 4, dt, V=V)
    u_n, v_n, t = model.rk(t0, tf)            # u_n.x.array is in DOLFINx's own DOF numbering
    copy_to_function(u_n, dolfinx_u)          # ... so VTXWriter / post-processing keep working
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


def tensor_dofmap_of(V, P: int, basix):
    """``reorder_dofmap`` (permute.hpp:15-42): the cell dofmap of the locally owned cells with every row
    permuted by ``argsort(tp_dof_ordering)``."""
    cell = basix.CellType.hexahedron if V.mesh.topology.dim == 3 else basix.CellType.quadrilateral
    tp_order = np.asarray(basix.tp_dof_ordering(basix.ElementFamily.P, cell, P,
                                                basix.LagrangeVariant.gll_warped, basix.DPCVariant.unset, False))
    perm = np.argsort(tp_order, kind="stable")                    # permute.hpp:27-32
    dm = np.asarray(V.dofmap.list).reshape(-1, len(tp_order))
    ncells = V.mesh.topology.index_map(V.mesh.topology.dim).size_local
    return np.ascontiguousarray(dm[:ncells][:, perm], dtype=np.int32)   # permute.hpp:38-41


def wrap_function_space(V, P: int, basix=None, alltoall=None):
    """``V``: a ``dolfinx.fem.FunctionSpace`` of degree-P ``gll_warped`` Lagrange on hexahedra or
    quadrilaterals.  ``alltoall(send: dict rank -> int64 array) -> dict rank -> int64 array`` moves the
    sharer messages between the ranks (default: ``mesh.comm`` through mpi4py)."""
    if basix is None:
        basix, _ = _require()
    tensor_dofmap = tensor_dofmap_of(V, P, basix)
    pts, _ = basix.make_quadrature(basix.CellType.interval, QDEGREE[P], basix.QuadratureType.gll)
    nodes1d = np.ascontiguousarray(np.asarray(pts).reshape(-1), dtype=np.float64)
    return _WrappedSpace(V, P, tensor_dofmap, nodes1d, _neighbours(V, alltoall))


def _dest_ranks(im):
    dest = im.index_to_dest_ranks()
    return np.asarray(dest.offsets), np.asarray(dest.array)


def sharer_messages(im):
    """Owner side.  For every owned DOF that two or more other ranks ghost, the message to each of those
    ranks r: (global index, k, the k other ghosting ranks besides r).  Returns {rank: int64 array}."""
    offs, ranks = _dest_ranks(im)
    lo = int(im.local_range[0])
    cnt = np.diff(offs[:im.size_local + 1])
    out = {}
    for i in np.nonzero(cnt >= 2)[0]:
        R = [int(r) for r in ranks[offs[i]:offs[i + 1]]]
        for r in R:
            out.setdefault(r, []).extend([lo + int(i), len(R) - 1, *[q for q in R if q != r]])
    return {r: np.asarray(v, dtype=np.int64) for r, v in out.items()}


def neighbours_from_index_map(im, received):
    """(rank, local dof indices) per neighbour rank with EVERY pair of holders of a DOF listing each
    other, each list ordered by global index so that both sides agree.  ``received``: the other
    ranks' :func:`sharer_messages` addressed to this rank, {source rank: int64 array}."""
    n_owned = int(im.size_local)
    lo = int(im.local_range[0])
    owners = np.asarray(im.owners).astype(np.int64)
    ghosts = np.asarray(im.ghosts).astype(np.int64)
    shared = {}                                                   # rank -> [(global, local)]
    for k in range(len(ghosts)):                                  # my ghosts: held by their owner
        shared.setdefault(int(owners[k]), []).append((int(ghosts[k]), n_owned + k))
    offs, ranks = _dest_ranks(im)                                 # my owned dofs that other ranks ghost
    cnt = np.diff(offs[:n_owned + 1])
    for i in np.nonzero(cnt > 0)[0]:
        for r in ranks[offs[i]:offs[i + 1]]:
            shared.setdefault(int(r), []).append((lo + int(i), int(i)))
    ghost_local = {int(g): n_owned + k for k, g in enumerate(ghosts)}
    for src, msg in received.items():                             # the other holders of my ghosts
        msg = np.asarray(msg, dtype=np.int64)
        p = 0
        while p < len(msg):
            g, k = int(msg[p]), int(msg[p + 1])
            if g not in ghost_local:
                raise ValueError(f"rank {src} names global dof {g}, which is not a ghost here")
            for q in msg[p + 2:p + 2 + k]:
                shared.setdefault(int(q), []).append((g, ghost_local[g]))
            p += 2 + k
    out = []
    for r in sorted(shared):
        pairs = sorted(set(shared[r]))
        out.append((r, np.array([p[1] for p in pairs], dtype=np.int32)))
    return out


def _mpi_alltoall(comm):
    def alltoall(send):
        size = comm.Get_size()
        got = comm.alltoall([send.get(r, np.zeros(0, np.int64)) for r in range(size)])
        return {r: m for r, m in enumerate(got) if len(m)}
    return alltoall


def _neighbours(V, alltoall=None):
    im = V.dofmap.index_map
    if im.num_ghosts == 0 and im.size_global == im.size_local:
        return []
    if alltoall is None:
        alltoall = _mpi_alltoall(V.mesh.comm)
    return neighbours_from_index_map(im, alltoall(sharer_messages(im)))


def copy_to_function(src, dst):
    """Model output (``.x.array`` in DOLFINx's DOF numbering, owned + ghosts) -> a ``dolfinx.fem.Function``."""
    dst.x.array[:] = np.asarray(src.x.array, dtype=dst.x.array.dtype)
    return dst


def wrap_facet_tags(mesh, meshtags):
    """DOLFINx ``MeshTags`` on facets -> object with ``cells``, ``local_facets``, ``values``
    ((cell, local facet) pairs of the tagged exterior facets, Linear.hpp:113-118)."""
    _require()
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
