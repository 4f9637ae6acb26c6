"""Host-side mirror of the reference operator classes (cpp/fenicsx-sf/common/spectral_op.hpp):
``StiffnessSpectral3D`` (:132-284) and ``MassSpectral3D`` (:29-107).  Same call shape --
``op(x, coeffs, y)`` accumulates into ``y`` -- with numpy arrays (or anything exposing
``.x.array`` / ``.array``) in place of ``la::Vector``."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._abi import Context, check, lib, ptr

_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _array(v):
    if hasattr(v, "x") and hasattr(v.x, "array"):
        return v.x.array
    if hasattr(v, "array") and not isinstance(v, np.ndarray):
        return v.array
    return v


class SpectralOperatorData:
    """Device-resident operator data shared by the mass and stiffness operators and the models
    (the reference recomputes G/dofmap per operator object, Lossy.hpp:152-153; here it is built
    once).  ``V`` needs ``.mesh.geometry.x/.dofmap``, ``.tensor_dofmap`` (or ``.dofmap.list``
    already in tensor order), ``.nodes1d`` and ``.P``."""

    def __init__(self, V, ctx: Context | None = None, fields: int = 1):
        self.ctx = ctx or default_context()
        self.fields = fields
        mesh = V.mesh
        self.V = V
        self.P = int(V.P)
        xg = np.ascontiguousarray(mesh.geometry.x)
        self.dtype = xg.dtype
        gdm = np.ascontiguousarray(mesh.geometry.dofmap, dtype=np.int32)
        tdm = np.ascontiguousarray(getattr(V, "tensor_dofmap", V.dofmap.list), dtype=np.int32)
        nodes = np.ascontiguousarray(V.nodes1d, dtype=np.float64)
        self.ncells = gdm.shape[0]
        self.ndofs = V.dofmap.index_map.size_local + V.dofmap.index_map.num_ghosts
        self.h = C.c_void_p()
        tdim = self.tdim = mesh.topology.dim
        order = 1 if gdm.shape[1] == (1 << tdim) else 2
        self.ctx.set_option("fields", fields)   # LDS sizing: 2 operator inputs for the lossy model
        try:
            check(lib().fus_op_create(self.ctx.h, C.c_int(tdim), C.c_int(self.P),
                                      C.c_int(_abi.dtype_code(self.dtype)), C.c_int64(self.ncells),
                                      C.c_int64(self.ndofs), ptr(tdm), ptr(nodes), ptr(xg),
                                      C.c_int64(xg.shape[0]), ptr(gdm), C.c_int(order), C.byref(self.h)))
        finally:
            self.ctx.set_option("fields", 1)
        neigh = getattr(V, "neighbours", [])
        if neigh:
            ranks = np.array([r for r, _ in neigh], dtype=np.int32)
            counts = np.array([len(i) for _, i in neigh], dtype=np.int64)
            idx = np.ascontiguousarray(np.concatenate([i for _, i in neigh]), dtype=np.int32)
            check(lib().fus_op_set_neighbours(self.h, C.c_int(len(neigh)), ptr(ranks), ptr(counts), ptr(idx)))

    def _apply(self, fn, x, coeffs, y):
        x = np.ascontiguousarray(_array(x), dtype=self.dtype)
        coeffs = np.ascontiguousarray(_array(coeffs), dtype=self.dtype)
        ya = _array(y)
        assert ya.dtype == self.dtype and ya.flags.c_contiguous and ya.shape[0] == self.ndofs
        assert x.shape[0] == self.ndofs and coeffs.shape[0] == self.ncells
        check(fn(self.h, ptr(x), ptr(coeffs), ptr(ya), C.c_int(_abi.FUS_HOST)))
        return y

    def stiffness(self, x, coeffs, y):
        return self._apply(lib().fus_stiffness_apply, x, coeffs, y)

    def mass(self, x, coeffs, y):
        return self._apply(lib().fus_mass_apply, x, coeffs, y)

    def geometry(self):
        Nd = (self.P + 1) ** self.tdim
        G = np.empty((self.ncells, Nd, 6 if self.tdim == 3 else 3), dtype=self.dtype)
        dJ = np.empty((self.ncells, Nd), dtype=self.dtype)
        check(lib().fus_op_get_geometry(self.h, ptr(G), ptr(dJ)))
        return G, dJ

    def tables(self):
        N = self.P + 1
        w, D = np.empty(N), np.empty((N, N))
        check(lib().fus_op_get_tables(self.h, ptr(w), ptr(D)))
        return w, D

    def info(self):
        out = (C.c_int64 * 8)()
        check(lib().fus_op_info(self.h, out))
        keys = ["nblocks", "interior_dofs", "shared_dofs", "pairs", "max_local_dofs", "shapes", "lds_bytes",
                "internal_len"]
        return dict(zip(keys, list(out)))

    def hmin(self) -> float:
        """Smallest local cell size (largest vertex distance per cell, dolfinx mesh::h; main.cpp:60-64)."""
        out = C.c_double()
        check(lib().fus_op_hmin(self.h, C.byref(out)))
        return out.value

    def norm2(self, x) -> float:
        """Local part of the squared L2 norm, sum over the local cells of the GLL integral of x^2
        (assemble_scalar(u*u*dx), main.cpp:151-157); sum it over the ranks."""
        x = np.ascontiguousarray(_array(x), dtype=self.dtype)
        assert x.shape[0] == self.ndofs
        out = C.c_double()
        check(lib().fus_op_norm2(self.h, ptr(x), C.c_int(_abi.FUS_HOST), C.byref(out)))
        return out.value

    def is_affine(self) -> bool:
        return bool(lib().fus_op_is_affine(self.h))

    def geometry_mode(self) -> str:
        return ("stream", "affine", "trilinear")[lib().fus_op_geometry_mode(self.h)]

    def uses_mfma(self) -> bool:
        return bool(lib().fus_op_uses_mfma(self.h))

    def uses_mfma4(self) -> bool:
        """Index-1 contraction on v_mfma_f64_4x4x4_4b_f64 (degree 7, fp64, trilinear kernel)."""
        return bool(lib().fus_op_uses_mfma4(self.h))

    def uses_diag_metric(self) -> bool:
        """Affine cells with orthogonal edges: stiffness action as three 1-D stiffness contractions (fusmi.h)."""
        return bool(lib().fus_op_uses_diag_metric(self.h))

    def uses_pack32(self) -> bool:
        return bool(lib().fus_op_uses_pack32(self.h))

    def facet_diag(self, cells, local_facets, cellcoef):
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        lf = np.ascontiguousarray(local_facets, dtype=np.int32)
        cc = np.ascontiguousarray(cellcoef, dtype=self.dtype)
        out = np.zeros(self.ndofs, dtype=self.dtype)
        check(lib().fus_facet_diag(self.h, C.c_int64(len(cells)), ptr(cells), ptr(lf), ptr(cc), ptr(out)))
        return out

    def halo_layout(self):
        """External transport: (ranks, counts, offsets) of the neighbours in buffer order (values)."""
        n = C.c_int()
        check(lib().fus_op_halo_layout(self.h, C.byref(n), None, None, None))
        ranks, counts, offs = np.zeros(n.value, np.int32), np.zeros(n.value, np.int64), np.zeros(n.value, np.int64)
        check(lib().fus_op_halo_layout(self.h, C.byref(n), ptr(ranks), ptr(counts), ptr(offs)))
        return ranks, counts, offs

    def halo_buffers(self):
        """External transport: device addresses of the send / receive buffers and their length in values."""
        s, r, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(lib().fus_op_halo_buffers(self.h, C.byref(s), C.byref(r), C.byref(n)))
        return s.value, r.value, n.value

    def close(self):
        if self.h:
            lib().fus_op_destroy(self.h)
            self.h = C.c_void_p()


class StiffnessSpectral3D:
    """``StiffnessSpectral3D<T,P>(V)``; ``op(x, coeffs, y)``: y += K(coeffs) x
    (spectral_op.hpp:135-171, 173-243)."""

    def __init__(self, V, data: SpectralOperatorData | None = None, ctx: Context | None = None):
        self.data = data or SpectralOperatorData(V, ctx)

    def __call__(self, x, coeffs, y):
        return self.data.stiffness(x, coeffs, y)


class MassSpectral3D:
    """``MassSpectral3D<T,P>(V)``; ``op(x, coeffs, y)``: y += M(coeffs) x (spectral_op.hpp:32-86)."""

    def __init__(self, V, data: SpectralOperatorData | None = None, ctx: Context | None = None):
        self.data = data or SpectralOperatorData(V, ctx)

    def __call__(self, x, coeffs, y):
        return self.data.mass(x, coeffs, y)


class StiffnessSpectral2D(StiffnessSpectral3D):
    """``StiffnessSpectral2D<T,P>(V)`` on quadrilaterals
    (cpp/fenicsx-sf-naive/common/spectral_op.hpp:226-359); same call convention."""

    def __init__(self, V, data: SpectralOperatorData | None = None, ctx: Context | None = None):
        super().__init__(V, data, ctx)
        assert self.data.tdim == 2


class MassSpectral2D(MassSpectral3D):
    """``MassSpectral2D<T,P>(V)`` on quadrilaterals (cpp/fenicsx-sf-naive/common/spectral_op.hpp:29-107)."""

    def __init__(self, V, data: SpectralOperatorData | None = None, ctx: Context | None = None):
        super().__init__(V, data, ctx)
        assert self.data.tdim == 2
