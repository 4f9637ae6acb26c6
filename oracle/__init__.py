"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- see oracle/oracle.h).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``fenicsx-fus_amd``) never does.  Parity status: primitives pinned
against the reference header compiled into ``oracle/_ref``; operator/RK4 level pinned by analytic
known-answer tests only (no stored reference vectors exist) -- "parity unpinned" at that level.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DT = {"f64": (np.float64, C.c_double), "f32": (np.float32, C.c_float)}


def build(force: bool = False) -> None:
    """Compile liboracle*.so (and oracle/_ref when the reference tree is present)."""
    if force or not os.path.exists(os.path.join(_HERE, "liboracle.so")) or not os.path.exists(
        os.path.join(_HERE, "liboracle_fast.so")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)


def _load(name: str) -> C.CDLL:
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


_libs: dict = {}


def lib(fast: bool = False) -> C.CDLL:
    key = "liboracle_fast.so" if fast else "liboracle.so"
    if key not in _libs:
        _libs[key] = _load(key)
    return _libs[key]


def ref_lib():
    """The compiled reference header (oracle/_ref), or None if it was never built here."""
    path = os.path.join(_HERE, "_ref", "libref_sumfact.so")
    if "ref" not in _libs:
        _libs["ref"] = C.CDLL(path) if os.path.exists(path) else None
    return _libs["ref"]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _suf(dtype) -> str:
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


# ---- 1-D tables -------------------------------------------------------------------------------
def gll(N: int):
    pts, wts = np.empty(N), np.empty(N)
    lib().orc_gll(C.c_int(N), _p(pts), _p(wts))
    return pts, wts


def gll_weights_at(nodes):
    nodes = _arr(nodes, np.float64)
    w = np.empty_like(nodes)
    lib().orc_gll_weights_at(C.c_int(len(nodes)), _p(nodes), _p(w))
    return w


def dphi(nodes):
    nodes = _arr(nodes, np.float64)
    N = len(nodes)
    D = np.empty((N, N))
    lib().orc_dphi(C.c_int(N), _p(nodes), _p(D))
    return D


# ---- primitives ---------------------------------------------------------------------------------
def contract(A, B, shape, transpose: bool, dtype=np.float64):
    """C[a,{b,c}] = sum_k A[a,k] B[k,b,c] (transpose) / A[k,a] B[k,b,c]; shape=(Nk,Na,Nb,Nc)."""
    Nk, Na, Nb, Nc = shape
    A, B = _arr(A, dtype), _arr(B, dtype)
    Cc = np.zeros(Na * Nb * Nc, dtype=dtype)
    getattr(lib(), "orc_contract_" + _suf(dtype))(
        C.c_int(Nk), C.c_int(Na), C.c_int(Nb), C.c_int(Nc), C.c_int(int(transpose)), _p(A), _p(B), _p(Cc)
    )
    return Cc


def transpose3(A, dims, offs, dtype=np.float64):
    A = _arr(A, dtype)
    B = np.zeros(A.size, dtype=dtype)
    getattr(lib(), "orc_transpose3_" + _suf(dtype))(
        *(C.c_int(d) for d in dims), *(C.c_int(o) for o in offs), _p(A), _p(B)
    )
    return B


# ---- geometry / operators -------------------------------------------------------------------------
def geometry(tdim, xg, xdofmap, pts, wts, dtype=np.float64):
    xg, xd = _arr(xg, dtype), _arr(xdofmap, np.int32)
    pts, wts = _arr(pts, np.float64), _arr(wts, np.float64)
    N = len(pts)
    nc = xd.shape[0]
    Nd = N**tdim
    ng = 6 if tdim == 3 else 3
    G = np.empty((nc, Nd, ng), dtype=dtype)
    detJ = np.empty((nc, Nd), dtype=dtype)
    if xd.shape[1] == 27 or (tdim == 2 and xd.shape[1] == 9):   # second-order cells, tensor node order
        getattr(lib(), ("orc_geometry_q2_" if tdim == 3 else "orc_geometry_q2_2d_") + _suf(dtype))(
            C.c_int64(nc), _p(xg), _p(xd), C.c_int(N), _p(pts), _p(wts), _p(G), _p(detJ))
        return G, detJ
    getattr(lib(), "orc_geometry_" + _suf(dtype))(
        C.c_int(tdim), C.c_int64(nc), _p(xg), _p(xd), C.c_int(N), _p(pts), _p(wts), _p(G), _p(detJ)
    )
    return G, detJ


def mass(tdim, N, tensor_dofmap, detJ, coeffs, x, y, dtype=np.float64, fast=False):
    dm = _arr(tensor_dofmap, np.int32)
    nc = dm.shape[0]
    getattr(lib(fast), "orc_mass_" + _suf(dtype))(
        C.c_int(tdim), C.c_int64(nc), C.c_int(N), _p(dm), _p(_arr(detJ, dtype)), _p(_arr(coeffs, dtype)),
        _p(_arr(x, dtype)), _p(y)
    )
    return y


def stiffness(tdim, N, tensor_dofmap, G, D, coeffs, x, y, dtype=np.float64, fast=False, dense=False):
    """y += K(coeffs) x with the reference's per-cell call sequence (y must be C-contiguous)."""
    dm = _arr(tensor_dofmap, np.int32)
    nc = dm.shape[0]
    name = "orc_stiffness3d_dense_" if dense else ("orc_stiffness3d_" if tdim == 3 else "orc_stiffness2d_")
    assert y.flags.c_contiguous and y.dtype == np.dtype(dtype)
    getattr(lib(fast), name + _suf(dtype))(
        C.c_int64(nc), C.c_int(N), _p(dm), _p(_arr(G, dtype)), _p(_arr(D, dtype)), _p(_arr(coeffs, dtype)),
        _p(_arr(x, dtype)), _p(y)
    )
    return y


def facet_diag(tdim, facet_cell, facet_local, cellcoef, xg, xdofmap, pts, wts, tensor_dofmap, ndofs,
               dtype=np.float64):
    fc, fl = _arr(facet_cell, np.int32), _arr(facet_local, np.int32)
    out = np.zeros(ndofs, dtype=dtype)
    if np.asarray(xdofmap).shape[1] == 27 or (tdim == 2 and np.asarray(xdofmap).shape[1] == 9):
        getattr(lib(), ("orc_facet_diag_q2_" if tdim == 3 else "orc_facet_diag_q2_2d_") + _suf(dtype))(
            C.c_int64(len(fc)), _p(fc), _p(fl), _p(_arr(cellcoef, dtype)), _p(_arr(xg, dtype)),
            _p(_arr(xdofmap, np.int32)), C.c_int(len(pts)), _p(_arr(pts, np.float64)), _p(_arr(wts, np.float64)),
            _p(_arr(tensor_dofmap, np.int32)), _p(out))
        return out
    getattr(lib(), "orc_facet_diag_" + _suf(dtype))(
        C.c_int(tdim), C.c_int64(len(fc)), _p(fc), _p(fl), _p(_arr(cellcoef, dtype)), _p(_arr(xg, dtype)),
        _p(_arr(xdofmap, np.int32)), C.c_int(len(pts)), _p(_arr(pts, np.float64)), _p(_arr(wts, np.float64)),
        _p(_arr(tensor_dofmap, np.int32)), _p(out)
    )
    return out


def linear_rk4(tdim, N, tensor_dofmap, G, D, coeff, m, src, absb, freq, p0, s0, t0, tf, dt, u, v,
               dtype=np.float64, fast=False, order=4):
    """Linear.hpp rk4 restated (order 1-3: the Python reference's other RK tables); u, v updated in
    place; returns the number of steps taken."""
    dm = _arr(tensor_dofmap, np.int32)
    fn = getattr(lib(fast), "orc_linear_rk_" + _suf(dtype))
    fn.restype = C.c_int64
    assert u.flags.c_contiguous and v.flags.c_contiguous
    return fn(
        C.c_int(order), C.c_int(tdim), C.c_int64(dm.shape[0]), C.c_int64(len(u)), C.c_int(N), _p(dm), _p(_arr(G, dtype)),
        _p(_arr(D, dtype)), _p(_arr(coeff, dtype)), _p(_arr(m, dtype)), _p(_arr(src, dtype)),
        _p(_arr(absb, dtype)), C.c_double(freq), C.c_double(p0), C.c_double(s0), C.c_double(t0),
        C.c_double(tf), C.c_double(dt), _p(u), _p(v)
    )


def lossy_rk4(tdim, N, tensor_dofmap, G, D, lin_coeff, att_coeff, m, src, absb, src2, freq, p0, s0, t0, tf, dt,
              u, v, dtype=np.float64, fast=False, source_scale=2.0):
    """Lossy.hpp rk4 restated; u, v updated in place; returns the number of steps taken.
    source_scale 2: Lossy.hpp:216-220; 1: the Python package's unscaled source (_lossy.py:186-189)."""
    dm = _arr(tensor_dofmap, np.int32)
    fn = getattr(lib(fast), "orc_lossy_rk4_s_" + _suf(dtype))
    fn.restype = C.c_int64
    assert u.flags.c_contiguous and v.flags.c_contiguous
    return fn(
        C.c_int(tdim), C.c_int64(dm.shape[0]), C.c_int64(len(u)), C.c_int(N), _p(dm), _p(_arr(G, dtype)),
        _p(_arr(D, dtype)), _p(_arr(lin_coeff, dtype)), _p(_arr(att_coeff, dtype)), _p(_arr(m, dtype)),
        _p(_arr(src, dtype)), _p(_arr(absb, dtype)), _p(_arr(src2, dtype)), C.c_double(freq), C.c_double(p0),
        C.c_double(s0), C.c_double(t0), C.c_double(tf), C.c_double(dt), _p(u), _p(v), C.c_double(source_scale)
    )


def westervelt_rk4(tdim, N, tensor_dofmap, G, detJ, D, lin_coeff, att_coeff, nlin1, nlin2, m0, src, absb, src2,
                   freq, p0, s0, t0, tf, dt, u, v, dtype=np.float64, fast=False, source_scale=2.0):
    """Westervelt.hpp rk4 restated; u, v updated in place; returns the number of steps taken."""
    dm = _arr(tensor_dofmap, np.int32)
    fn = getattr(lib(fast), "orc_westervelt_rk4_s_" + _suf(dtype))
    fn.restype = C.c_int64
    assert u.flags.c_contiguous and v.flags.c_contiguous
    a = lambda x: _p(_arr(x, dtype))  # noqa: E731
    return fn(
        C.c_int(tdim), C.c_int64(dm.shape[0]), C.c_int64(len(u)), C.c_int(N), _p(dm), a(G), a(detJ), a(D),
        a(lin_coeff), a(att_coeff), a(nlin1), a(nlin2), a(m0), a(src), a(absb), a(src2), C.c_double(freq),
        C.c_double(p0), C.c_double(s0), C.c_double(t0), C.c_double(tf), C.c_double(dt), _p(u), _p(v),
        C.c_double(source_scale)
    )


def linear_rk4_mt(N, tensor_dofmap, G, D, coeff, m, src, absb, freq, p0, s0, t0, tf, dt, u, v, slab_cell_off,
                  dtype=np.float64, fast=True):
    """Threaded CPU-baseline variant (3-D): one OpenMP thread per contiguous cell slab."""
    dm = _arr(tensor_dofmap, np.int32)
    off = _arr(slab_cell_off, np.int64)
    fn = getattr(lib(fast), "orc_linear_rk4_mt_" + _suf(dtype))
    fn.restype = C.c_int64
    return fn(
        C.c_int64(dm.shape[0]), C.c_int64(len(u)), C.c_int(N), _p(dm), _p(_arr(G, dtype)), _p(_arr(D, dtype)),
        _p(_arr(coeff, dtype)), _p(_arr(m, dtype)), _p(_arr(src, dtype)), _p(_arr(absb, dtype)), C.c_double(freq),
        C.c_double(p0), C.c_double(s0), C.c_double(t0), C.c_double(tf), C.c_double(dt), _p(u), _p(v),
        C.c_int(len(off) - 1), _p(off)
    )
