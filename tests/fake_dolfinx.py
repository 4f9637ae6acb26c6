"""Duck-typed stand-ins for the DOLFINx / Basix objects the adapter reads (neither is installed here).

`partition(pr, part, P, seed)` cuts a single-rank `Problem` into ranks by a cell -> rank map and gives every
rank what DOLFINx would: an unstructured local mesh, a function space whose `dofmap.list` is in a
non-tensor ("Basix") local ordering, and an `IndexMap` with owned DOFs first, ghosts after, owners =
lowest holding rank, contiguous global ranges per rank and `index_to_dest_ranks()`.  `FakeBasix` returns a
fixed permutation as `tp_dof_ordering` and the GLL points in Basix's endpoint-first order."""
import numpy as np

from fenicsxfus_amd import tables
from fenicsxfus_amd.unstructured import HexFunctionSpace, HexMesh


class _Enum:
    def __getattr__(self, k):
        return k


class FakeBasix:
    CellType = ElementFamily = LagrangeVariant = DPCVariant = QuadratureType = _Enum()

    def __init__(self, P, tdim, seed=0, endpoints_first=False):
        self.P, self.tdim = P, tdim
        N = P + 1
        self.tp = np.random.default_rng(seed).permutation(N**tdim)
        pts, _ = tables.gll(N)
        # Basix lists interval DOFs / GLL points as the two end points first, then the interior ones
        self.order1d = np.r_[0, N - 1, np.arange(1, N - 1)] if endpoints_first else np.arange(N)
        self.pts = np.asarray(pts)[self.order1d]

    def tp_dof_ordering(self, family, cell, degree, variant, dpc, discontinuous):
        assert degree == self.P and cell == ("hexahedron" if self.tdim == 3 else "quadrilateral")
        assert variant == "gll_warped" and not discontinuous
        return list(self.tp)

    def make_quadrature(self, cell, degree, qtype):
        assert cell == "interval" and qtype == "gll"
        return self.pts.reshape(-1, 1), None


class _Adj:
    def __init__(self, offsets, array):
        self.offsets, self.array = offsets, array


class FakeIndexMap:
    def __init__(self, size_local, ghosts, owners, lo, size_global, dest):
        self.size_local, self.num_ghosts = int(size_local), len(ghosts)
        self.ghosts, self.owners = np.asarray(ghosts, np.int64), np.asarray(owners, np.int32)
        self.local_range = (int(lo), int(lo) + int(size_local))
        self.size_global = int(size_global)
        self._dest = dest

    def index_to_dest_ranks(self):
        return self._dest


class _DofMap:
    def __init__(self, lst, im):
        self.list, self.index_map = lst, im


class FakeSpace:
    def __init__(self, mesh, lst, im):
        self.mesh, self.dofmap = mesh, _DofMap(lst, im)


def partition(pr, part, P, basix):
    """Returns per rank: dict(V=FakeSpace, cells=global cell ids, oracle_ids=[local dolfinx index] -> index
    in the single-rank problem's vectors, tensor_local=the tensor dofmap in DOLFINx-local numbering)."""
    from scipy.spatial import cKDTree

    size = int(part.max()) + 1
    # positions of the single-rank dofs, to identify the local spaces' dofs globally
    Xg = np.zeros((pr.ndofs, 3))
    Xg[pr.dm] = HexFunctionSpace(HexMesh(pr.mesh.geometry.x, pr.mesh.geometry.dofmap), P)._node_x
    tree = cKDTree(Xg)
    hi = np.ptp(pr.mesh.geometry.x, axis=0).max()
    loc = []
    for r in range(size):
        cells = np.nonzero(part == r)[0]
        used, inv = np.unique(pr.mesh.geometry.dofmap[cells], return_inverse=True)
        lmesh = HexMesh(pr.mesh.geometry.x[used], inv.reshape(len(cells), 8))
        Vt = HexFunctionSpace(lmesh, P)
        d, gid = tree.query(Vt.tabulate_dof_coordinates())
        assert d.max() < 1e-9 * hi
        loc.append((cells, lmesh, Vt, gid))
    holders = {}
    for r, (_, _, _, gid) in enumerate(loc):
        for g in gid:
            holders.setdefault(int(g), []).append(r)
    owner = {g: min(h) for g, h in holders.items()}
    owned = [sorted(g for g in set(loc[r][3].tolist()) if owner[g] == r) for r in range(size)]
    off = np.r_[0, np.cumsum([len(o) for o in owned])]
    dglob = {}                                               # oracle id -> DOLFINx global index
    for r in range(size):
        for k, g in enumerate(owned[r]):
            dglob[g] = int(off[r]) + k
    out = []
    N = P + 1
    # 1-D node order: element-local tensor index i = (i0 N + i1) N + i2 with i_d counting the nodes in
    # basix.order1d (any order, as long as dofmap, points and tables agree)
    o1 = np.asarray(basix.order1d)
    t3 = (o1[:, None, None] * N + o1[None, :, None]) * N + o1[None, None, :]
    for r, (cells, lmesh, Vt, gid) in enumerate(loc):
        mine = owned[r]
        ghost_g = sorted((g for g in set(gid.tolist()) if owner[g] != r), key=lambda g: dglob[g])
        newlocal = {g: i for i, g in enumerate(mine)}
        newlocal.update({g: len(mine) + k for k, g in enumerate(ghost_g)})
        renum = np.array([newlocal[int(g)] for g in gid])            # hex-local -> DOLFINx-local
        tensor_local = renum[Vt.tensor_dofmap][:, t3.ravel()]        # tensor order w.r.t. basix's 1-D order
        lst = np.empty_like(tensor_local)
        lst[:, np.argsort(basix.tp)] = tensor_local                  # so that lst[:, argsort(tp)] is tensor order
        offs, arr = [0], []
        for g in mine:
            others = [q for q in holders[g] if q != r]
            arr.extend(others)
            offs.append(len(arr))
        im = FakeIndexMap(len(mine), [dglob[g] for g in ghost_g], [owner[g] for g in ghost_g], off[r], off[-1],
                          _Adj(np.array(offs, np.int32), np.array(arr, np.int32)))
        out.append(dict(V=FakeSpace(lmesh, lst.astype(np.int32), im), cells=cells, mesh=lmesh,
                        oracle_ids=np.array(mine + ghost_g, dtype=np.int64), tensor_local=tensor_local.astype(np.int32)))
    return out


def exchange_all(messages):
    """In-process all-to-all: messages[r] = {dest: array} -> received[r] = {src: array}."""
    size = len(messages)
    return [{s: messages[s][r] for s in range(size) if r in messages[s]} for r in range(size)]
