"""Shared problem builders for the tests (inputs follow SURVEY 8d / the reference tests' recipes)."""
import numpy as np

from fenicsxfus_amd import BoxMesh, FunctionSpace, FacetTags


class Problem:
    """Everything the oracle needs for one mesh: tables, geometry factors, dofmap."""

    def __init__(self, orc, n, P, lo=None, hi=None, perturb=0.0, node_order=None, dtype=np.float64,
                 rank=0, size=1, order=1, warp=None):
        t = len(n)
        lo = [0.0] * t if lo is None else lo
        hi = [1.0] * t if hi is None else hi
        self.mesh = BoxMesh(lo, hi, n, perturb=perturb, dtype=dtype, rank=rank, size=size, order=order, warp=warp)
        self.V = FunctionSpace(self.mesh, P, node_order=node_order)
        self.P, self.N, self.tdim, self.dtype = P, P + 1, t, np.dtype(dtype)
        self.nodes = self.V.nodes1d
        self.wts = orc.gll_weights_at(self.nodes)
        self.D = orc.dphi(self.nodes).astype(dtype)
        self.dm = self.V.tensor_dofmap
        self.ndofs = self.V.num_dofs
        self.G, self.detJ = orc.geometry(t, self.mesh.geometry.x, self.mesh.geometry.dofmap, self.nodes,
                                         self.wts, dtype=dtype)
        self.orc = orc

    def K(self, x, coeffs=None, dense=False, fast=False):
        c = np.ones(self.mesh.num_cells, self.dtype) if coeffs is None else coeffs
        y = np.zeros(self.ndofs, self.dtype)
        return self.orc.stiffness(self.tdim, self.N, self.dm, self.G, self.D, c, x, y, dtype=self.dtype,
                                  dense=dense, fast=fast)

    def M(self, x, coeffs=None):
        c = np.ones(self.mesh.num_cells, self.dtype) if coeffs is None else coeffs
        y = np.zeros(self.ndofs, self.dtype)
        return self.orc.mass(self.tdim, self.N, self.dm, self.detJ, c, x, y, dtype=self.dtype)

    def facet_diag(self, tags: FacetTags, tag, cellcoef):
        sel = tags.find(tag)
        return self.orc.facet_diag(self.tdim, tags.cells[sel], tags.local_facets[sel], cellcoef,
                                   self.mesh.geometry.x, self.mesh.geometry.dofmap, self.nodes, self.wts,
                                   self.dm, self.ndofs, dtype=self.dtype)

    def linear_model_vectors(self, c0, rho0, tags):
        """m, src, absb, coeff of the Linear model (Linear.hpp:127-134,154-155; forms.py:36-39)."""
        nc = self.mesh.num_cells
        c0 = np.broadcast_to(np.asarray(c0, self.dtype), (nc,)).copy()
        rho0 = np.broadcast_to(np.asarray(rho0, self.dtype), (nc,)).copy()
        m = self.M(np.ones(self.ndofs, self.dtype), 1.0 / (rho0 * c0 * c0))
        src = self.facet_diag(tags, 1, 1.0 / rho0)
        absb = self.facet_diag(tags, 2, 1.0 / (rho0 * c0))
        return m, src, absb, (-1.0 / rho0).astype(self.dtype)

    def lossy_model_vectors(self, c0, rho0, delta0, tags):
        """m, src, absb, src2, lin_coeff, att_coeff of the Lossy model (Lossy.hpp:133-141,166-169;
        BM7-SC1/forms.py:37-42): absorbing + delta mass term on every listed boundary facet."""
        nc = self.mesh.num_cells
        c0 = np.broadcast_to(np.asarray(c0, self.dtype), (nc,)).copy()
        rho0 = np.broadcast_to(np.asarray(rho0, self.dtype), (nc,)).copy()
        d0 = np.broadcast_to(np.asarray(delta0, self.dtype), (nc,)).copy()
        allf = FacetTags(tags.cells, tags.local_facets, np.full(len(tags.values), 7))
        m = self.M(np.ones(self.ndofs, self.dtype), 1.0 / (rho0 * c0 * c0))
        m = m + self.facet_diag(allf, 7, d0 / (rho0 * c0**3))
        src = self.facet_diag(tags, 1, 1.0 / rho0)
        absb = self.facet_diag(allf, 7, 1.0 / (rho0 * c0))
        src2 = self.facet_diag(tags, 1, d0 / (rho0 * c0 * c0))
        return m, src, absb, src2, (-1.0 / rho0).astype(self.dtype), (-d0 / (rho0 * c0 * c0)).astype(self.dtype)
