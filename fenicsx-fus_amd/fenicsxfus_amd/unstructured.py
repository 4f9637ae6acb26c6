"""Unstructured hexahedral / quadrilateral meshes: XDMF/HDF5 ingestion and the degree-P GLL function
space on them.

The reference reads Gmsh meshes through DOLFINx (``XDMFFile::read_mesh`` + ``read_meshtags``,
cpp/fenicsx-sf/benchmarks/PH1/BM7-SC1/main.cpp:55-64) and lets DOLFINx/Basix number the DOFs.
Here the same files are read with :mod:`hdf5_lite`, cells are brought from XDMF/VTK vertex order to
the tensor order libfusmi uses (v = vx + 2 vy + 4 vz), and the conforming tensor-product dofmap is
built geometrically: every element node is mapped to physical space with the tri-/bilinear map and
nodes that coincide are one DOF (GLL nodes on a shared face/edge coincide from both sides because
the map restricted to the shared entity is the same).  The count is checked against the
topological formula  #V + #E (P-1) + #F (P-1)^2 + #C (P-1)^3  (quadrilaterals: #V + #E (P-1) +
#C (P-1)^2; the reference's 2-D fixtures are cpp/fenicsx-sf-naive/tests/test_operators2d/mesh_1).
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

import numpy as np

from . import tables
from .hdf5_lite import H5File
from .mesh import FacetTags, _DofMap, _Geometry, _IndexMap, _Topology

# XDMF / VTK hexahedron vertex order -> tensor order (tensor[k] = vtk[VTK_TO_TENSOR[k]])
VTK_TO_TENSOR = np.array([0, 1, 3, 2, 4, 5, 7, 6])
# DOLFINx local facet -> its 4 vertices in tensor numbering (facet 0: z=0, 1: y=0, 2: x=0, 3: x=1,
# 4: y=1, 5: z=1)
FACET_VERTS = np.array([[0, 1, 2, 3], [0, 1, 4, 5], [0, 2, 4, 6], [1, 3, 5, 7], [2, 3, 6, 7], [4, 5, 6, 7]])
_EDGES = np.array([[0, 1], [2, 3], [4, 5], [6, 7], [0, 2], [1, 3], [4, 6], [5, 7], [0, 4], [1, 5], [2, 6], [3, 7]])


# quadrilaterals: XDMF/VTK order is counter-clockwise; tensor v = vx + 2 vy.  DOLFINx local facets
# (edges): 0: y=0, 1: x=0, 2: x=1, 3: y=1
VTK_QUAD_TO_TENSOR = np.array([0, 1, 3, 2])
QUAD_FACET_VERTS = np.array([[0, 1], [0, 2], [1, 3], [2, 3]])
# 9-node (biquadratic) quadrilateral: VTK order is 4 corners, 4 mid-edge nodes (edges 0-1, 1-2, 2-3,
# 3-0), centre; tensor order is n = nx + 3 ny
VTK_QUAD9_TO_TENSOR = np.array([0, 4, 1, 7, 8, 5, 3, 6, 2])


def _vtk_hex27_to_tensor():
    """27-node (triquadratic) hexahedron: XDMF ``Hexahedron_27`` files written by DOLFINx / meshio list the
    nodes in VTK order -- 8 corners (VTK hexahedron order), 12 mid-edge nodes (edges 0-1, 1-2, 2-3, 3-0,
    4-5, 5-6, 6-7, 7-4, 0-4, 1-5, 2-6, 3-7), 6 face centres (x-, x+, y-, y+, z-, z+) and the body centre
    (DOLFINx ``io::cells::perm_vtk``); libfusmi takes them in tensor order n = nx + 3 ny + 9 nz.
    Returns perm with tensor[k] = vtk[perm[k]]."""
    corner = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    edges = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    pos = [tuple(2 * c for c in v) for v in corner]
    pos += [tuple(corner[a][d] + corner[b][d] for d in range(3)) for a, b in edges]
    pos += [(0, 1, 1), (2, 1, 1), (1, 0, 1), (1, 2, 1), (1, 1, 0), (1, 1, 2), (1, 1, 1)]
    perm = np.empty(27, dtype=np.int64)
    for i, (nx, ny, nz) in enumerate(pos):
        perm[nx + 3 * ny + 9 * nz] = i
    return perm


VTK_HEX27_TO_TENSOR = _vtk_hex27_to_tensor()


class HexMesh:
    """Unstructured hexahedral mesh, first-order (8 vertices, tensor order v = vx + 2 vy + 4 vz) or
    second-order (27 nodes, tensor order n = nx + 3 ny + 9 nz), with the attributes the adapter reads from
    a DOLFINx mesh (``geometry.x``, ``geometry.dofmap``, ``topology.dim``, ``index_map``)."""

    tdim = 3
    _facet_verts = FACET_VERTS
    _edges = _EDGES

    def __init__(self, x, cells, dtype=np.float64):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape[1] == 2:                                            # XDMF "XY" geometry
            x = np.hstack([x, np.zeros((x.shape[0], 1))])
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.order = 1 if cells.shape[1] == (1 << self.tdim) else 2
        assert x.shape[1] == 3 and cells.shape[1] == (self.order + 1) ** self.tdim
        # positions of the 2^tdim corner vertices (v = vx + 2 vy + 4 vz) inside a cell's node list
        g = self.order + 1
        self._corners = np.array([sum((((v >> d) & 1) * self.order) * g**d for d in range(self.tdim))
                                  for v in range(1 << self.tdim)])
        self.dtype = np.dtype(dtype)
        self.geometry = _Geometry(x.astype(self.dtype), cells, 3)
        self.topology = _Topology(self.tdim, cells.shape[0], cells.shape[0])
        self._x64 = x

    @property
    def num_cells(self):
        return self.geometry.dofmap.shape[0]

    def cell_centroids(self):
        return self._x64[self.vertex_dofmap].mean(axis=1)

    @property
    def vertex_dofmap(self):
        """[ncells, 2^tdim] corner vertices of each cell (the whole node list for first-order cells)."""
        return self.geometry.dofmap[:, self._corners]

    def _facet_keys(self):
        fv = self.vertex_dofmap[:, self._facet_verts]                 # [nc, nfacets, nverts]
        return np.sort(fv, axis=2).reshape(-1, self._facet_verts.shape[1])

    def exterior_facets(self):
        """(cells, local facets) of the facets that belong to exactly one cell."""
        keys = self._facet_keys()
        nf = self._facet_verts.shape[0]
        _, inv, cnt = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
        ext = np.nonzero(cnt[inv.ravel()] == 1)[0]
        return (ext // nf).astype(np.int32), (ext % nf).astype(np.int32)

    def facet_tags(self, facet_vertices, values) -> FacetTags:
        """Tags given per facet as vertex ids (XDMF ``MeshTags`` topology) -> (cell, local facet)
        pairs; only exterior facets are kept (the forms integrate over ``ds``)."""
        cells, lf = self.exterior_facets()
        keys = np.sort(self.vertex_dofmap[cells][np.arange(len(cells))[:, None], self._facet_verts[lf]], axis=1)
        lut = {tuple(k): i for i, k in enumerate(keys)}
        fc, fl, fv = [], [], []
        nfv = self._facet_verts.shape[1]          # higher-order facets list their end vertices first (VTK)
        for verts, val in zip(np.sort(np.asarray(facet_vertices)[:, :nfv], axis=1), np.asarray(values).ravel()):
            i = lut.get(tuple(verts))
            if i is not None:
                fc.append(cells[i]), fl.append(lf[i]), fv.append(val)
        return FacetTags(np.array(fc, np.int32), np.array(fl, np.int32), np.array(fv, np.int32))

    def entity_counts(self):
        """(#vertices used, #edges, #faces, #cells); a quadrilateral mesh has no faces besides its cells."""
        dm = self.vertex_dofmap
        nv = len(np.unique(dm))
        ne = len(np.unique(np.sort(dm[:, self._edges].reshape(-1, 2), axis=1), axis=0))
        nf = len(np.unique(self._facet_keys(), axis=0)) if self.tdim == 3 else 0
        return nv, ne, nf, dm.shape[0]


class QuadMesh(HexMesh):
    """Unstructured quadrilateral mesh: first-order (4 vertices, tensor order v = vx + 2 vy) or
    second-order (9 nodes, tensor order n = nx + 3 ny)."""

    tdim = 2
    _facet_verts = QUAD_FACET_VERTS
    _edges = QUAD_FACET_VERTS


class HexFunctionSpace:
    """Degree-P GLL Lagrange space on a :class:`HexMesh` or :class:`QuadMesh`; ``tensor_dofmap`` in
    tensor order like ``reorder_dofmap`` produces (cpp/fenicsx-sf/common/permute.hpp:15-42)."""

    def __init__(self, mesh: HexMesh, P: int, tol: float = 1e-9):
        self.mesh, self.P = mesh, int(P)
        N = P + 1
        t = mesh.tdim
        pts, _ = tables.gll(N)
        self.nodes1d = pts
        x = mesh._x64
        cd = x[mesh.geometry.dofmap]                                   # [nc, (g+1)^t, 3]
        # tensor Lagrange shape functions of the geometry (order g = 1: multilinear, 2: biquadratic,
        # node n = sum_d n_d (g+1)^d) at the N^t tensor nodes, tensor index (i0*N + i1)*N + i2
        g = getattr(mesh, "order", 1)
        grids = np.meshgrid(*([pts] * t), indexing="ij")
        X = np.stack([gr.ravel() for gr in grids], axis=1)             # [Nd, t]

        def basis1d(xv):
            if g == 1:
                return np.stack([1.0 - xv, xv], axis=1)
            return np.stack([(2 * xv - 1) * (xv - 1), 4 * xv * (1 - xv), xv * (2 * xv - 1)], axis=1)

        phi = np.ones((X.shape[0], (g + 1) ** t))
        for n in range((g + 1) ** t):
            for d in range(t):
                phi[:, n] *= basis1d(X[:, d])[:, (n // (g + 1) ** d) % (g + 1)]
        nodes = np.einsum("qv,cvk->cqk", phi, cd)                      # [nc, Nd, 3]
        self._node_x = nodes
        scale = np.ptp(x, axis=0).max()
        q = np.round(nodes.reshape(-1, 3) / (tol * scale)).astype(np.int64)
        _, first, inv = np.unique(q, axis=0, return_index=True, return_inverse=True)
        ndofs = len(first)
        nv, ne, nf, nc = mesh.entity_counts()
        expect = nv + ne * (P - 1) + nf * (P - 1) ** 2 + nc * (P - 1) ** t
        if ndofs != expect:
            raise ValueError(f"geometric dof matching found {ndofs} dofs, topology says {expect}")
        self.tensor_dofmap = np.ascontiguousarray(inv.reshape(nodes.shape[0], -1), dtype=np.int32)
        self.dofmap = _DofMap(self.tensor_dofmap, _IndexMap(ndofs))
        self._dof_x = nodes.reshape(-1, 3)[first]
        self.neighbours = []

    @property
    def num_dofs(self):
        return self.dofmap.index_map.size_local

    def tabulate_dof_coordinates(self):
        return self._dof_x.copy()


def read_xdmf_mesh(xdmf_path: str, name: str | None = None, dtype=np.float64):
    """Read ``<Grid Name=name>`` (default: the first grid) of an XDMF file written by DOLFINx/meshio
    and its cell / facet mesh tags if present.  Returns (HexMesh | QuadMesh, cell_values | None,
    FacetTags | None)."""
    root = ET.parse(xdmf_path).getroot()
    base = os.path.dirname(os.path.abspath(xdmf_path))
    grids = {g.get("Name"): g for g in root.iter("Grid")}
    grid = grids[name] if name else next(iter(grids.values()))
    name = grid.get("Name")
    files = {}

    def data(item):
        if item.get("Format", "HDF").upper() == "XML":          # values inline in the XDMF file
            dims = [int(k) for k in item.get("Dimensions").split()]
            kind = item.get("NumberType", item.get("DataType", "Float")).lower()
            return np.array(item.text.split(), dtype=np.float64 if kind == "float" else np.int64).reshape(dims)
        fname, dset = item.text.strip().split(":")
        if fname not in files:
            files[fname] = H5File(os.path.join(base, fname))
        return files[fname][dset]

    topo = grid.find("Topology")
    kind = topo.get("TopologyType").lower()
    x = data(grid.find("Geometry").find("DataItem"))
    if kind == "hexahedron" and topo.get("NodesPerElement", "8") == "8":
        mesh = HexMesh(x, data(topo.find("DataItem"))[:, VTK_TO_TENSOR], dtype=dtype)
    elif kind == "quadrilateral" and topo.get("NodesPerElement", "4") == "4":
        mesh = QuadMesh(x, data(topo.find("DataItem"))[:, VTK_QUAD_TO_TENSOR], dtype=dtype)
    elif kind == "quadrilateral_9":
        mesh = QuadMesh(x, data(topo.find("DataItem"))[:, VTK_QUAD9_TO_TENSOR], dtype=dtype)
    elif kind == "hexahedron_27":
        mesh = HexMesh(x, data(topo.find("DataItem"))[:, VTK_HEX27_TO_TENSOR], dtype=dtype)
    else:
        raise NotImplementedError("grids read: first- and second-order hexahedra and quadrilaterals")
    cell_vals, ftags = None, None
    g = grids.get(f"{name}_cells")
    if g is not None:
        cell_vals = data(g.find("Attribute").find("DataItem")).ravel()
    g = grids.get(f"{name}_facets")
    if g is not None:
        fverts = data(g.find("Topology").find("DataItem"))
        fvals = data(g.find("Attribute").find("DataItem")).ravel()
        ftags = mesh.facet_tags(fverts, fvals)
    return mesh, cell_vals, ftags


read_xdmf_hex_mesh = read_xdmf_mesh
