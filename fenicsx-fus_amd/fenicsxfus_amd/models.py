"""Host-side mirror of the reference model classes for the offloaded path.

``LinearSpectralExplicit`` keeps the constructor and ``init()`` / ``rk(t0, tf)`` signatures of
python/src/fenicsxfus/_linear.py:258-513 (and is the Python face of the C++ ``LinearSpectral3D``,
cpp/fenicsx-sf/common/Linear.hpp:52-347).  The whole RK4 loop runs on the GPU through
``fus_model_rk4`` with the reference's Runge-Kutta tables (``rk_order`` 1-4, _linear.py:286-311).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._abi import Context, check, lib, ptr
from .mesh import Function, FunctionSpace
from .operators import SpectralOperatorData, _array


class _SpectralExplicit:
    """Shared host side of the offloaded explicit models (constructor plumbing, init, rk)."""

    _kind = _abi.FUS_LINEAR

    def _create(self, mesh, meshtags, k, c0, rho0, delta0, freq0, p0, s0, rk_order, dt, V, ctx, beta0=None,
                forms="cpp"):
        if rk_order not in (1, 2, 3, 4):
            raise _abi.FusError("rk_order must be 1, 2, 3 or 4 (_linear.py:286-311)")
        if forms not in ("cpp", "python"):
            raise _abi.FusError("forms must be 'cpp' or 'python'")
        self.mesh, self.dt = mesh, dt
        self.freq, self.p0, self.s0 = float(freq0), float(p0), float(s0)
        self.V = V or FunctionSpace(mesh, k)
        self.data = SpectralOperatorData(self.V, ctx, fields=2 if delta0 is not None else 1)
        self.ctx = self.data.ctx
        dt_ = self.data.dtype
        c0a = np.ascontiguousarray(_array(c0), dtype=dt_)
        rhoa = np.ascontiguousarray(_array(rho0), dtype=dt_)
        dla = None if delta0 is None else np.ascontiguousarray(_array(delta0), dtype=dt_)
        bta = None if beta0 is None else np.ascontiguousarray(_array(beta0), dtype=dt_)
        cells = np.ascontiguousarray(meshtags.cells, dtype=np.int32)
        lf = np.ascontiguousarray(meshtags.local_facets, dtype=np.int32)
        tags = np.ascontiguousarray(meshtags.values, dtype=np.int32)
        self.h = C.c_void_p()
        self.ctx.set_option("forms", 1 if forms == "python" else 0)
        try:
            check(lib().fus_model_create(self.ctx.h, C.c_int(self._kind), self.data.h, ptr(c0a), ptr(rhoa),
                                         ptr(dla), ptr(bta), C.c_int64(len(cells)), ptr(cells), ptr(lf), ptr(tags),
                                         C.c_double(self.freq), C.c_double(self.p0), C.c_double(self.s0),
                                         C.byref(self.h)))
        finally:
            self.ctx.set_option("forms", 0)
        check(lib().fus_model_set_rk_order(self.h, C.c_int(rk_order)))
        self.u_n = Function(self.V, dt_)
        self.v_n = Function(self.V, dt_)
        self._nrecv = 0

    # ---- external transport (the caller exchanges the interface values; fusmi.h) ----
    def setup_count(self) -> int:
        return int(lib().fus_model_setup_count(self.h))

    def setup_pack(self, k: int):
        check(lib().fus_model_setup_pack(self.h, C.c_int(k)))

    def setup_unpack(self, k: int):
        check(lib().fus_model_setup_unpack(self.h, C.c_int(k)))

    def setup_finish(self):
        check(lib().fus_model_setup_finish(self.h))

    def stage_begin(self, i: int, t: float, dt: float):
        check(lib().fus_model_stage_begin(self.h, C.c_int(i), C.c_double(t), C.c_double(dt)))

    def stage_end(self, i: int, t: float, dt: float):
        check(lib().fus_model_stage_end(self.h, C.c_int(i), C.c_double(t), C.c_double(dt)))

    def init(self):
        """u_n = v_n = 0 (_linear.py:363-369, Linear.hpp:161-164)."""
        self.u_n.x.array[:] = 0.0
        self.v_n.x.array[:] = 0.0
        check(lib().fus_model_init(self.h))

    def set_state(self, u=None, v=None):
        for which, a in ((_abi.FUS_U, u), (_abi.FUS_V, v)):
            if a is not None:
                a = np.ascontiguousarray(_array(a), dtype=self.data.dtype)
                check(lib().fus_model_set(self.h, C.c_int(which), ptr(a), C.c_int(_abi.FUS_HOST)))

    def _pull(self):
        check(lib().fus_model_get(self.h, C.c_int(_abi.FUS_U), ptr(self.u_n.x.array), C.c_int(_abi.FUS_HOST)))
        check(lib().fus_model_get(self.h, C.c_int(_abi.FUS_V), ptr(self.v_n.x.array), C.c_int(_abi.FUS_HOST)))

    def rk(self, t0: float, tf: float):
        """Runge-Kutta solve from t0 to tf; returns (u_n, v_n, t) like _linear.py:430-513."""
        n = C.c_int64()
        check(lib().fus_model_rk4(self.h, C.c_double(t0), C.c_double(tf), C.c_double(self.dt), C.byref(n)))
        self.nsteps = n.value
        self._pull()
        return self.u_n, self.v_n, tf

    # C++-style aliases (Linear.hpp:228, 316, 318)
    def rk4(self, t0, tf, dt):
        self.dt = dt
        return self.rk(t0, tf)

    def rk4_steps(self, t0: float, dt: float, nsteps: int, sync: bool = True):
        """Exactly nsteps steps, asynchronous unless ``sync`` (benchmark entry)."""
        check(lib().fus_model_rk4_steps(self.h, C.c_double(t0), C.c_double(dt), C.c_int64(nsteps)))
        if sync:
            self.ctx.synchronize()

    def external_setup(self, exchange):
        """External transport, setup phase: sums the lumped mass (and Westervelt's mass diagonal) over
        the sharers through ``exchange()``, a callable that moves every neighbour's range of the send
        buffer into the matching range of the neighbour's receive buffer (see fusmi.h)."""
        for k in range(self.setup_count()):
            self.setup_pack(k)
            exchange()
            self.setup_unpack(k)
        self.setup_finish()

    def external_rk_steps(self, t0: float, dt: float, nsteps: int, exchange, rk_order: int = 4):
        """External transport: nsteps steps with the two halves of every stage around ``exchange()``."""
        t = t0
        for _ in range(nsteps):
            for i in range(rk_order):
                self.stage_begin(i, t, dt)
                exchange()
                self.stage_end(i, t, dt)
            t += dt

    # ---- receivers: point samples of the resident solution (utils.py:10-47 compute_eval_params + Function.eval) ----
    def set_receivers(self, points):
        """Locate ``points`` [n, tdim|3] in the local mesh (host, once) and hand (cell, reference coordinates) to the
        library; returns the indices of the points found on this rank (the reference's ``points_on_proc``)."""
        from .evaluate import locate

        cell, X = locate(self.mesh, points)
        on = np.flatnonzero(cell >= 0)
        cells = np.ascontiguousarray(cell[on], dtype=np.int32)
        Xr = np.ascontiguousarray(X[on], dtype=np.float64)
        check(lib().fus_model_set_receivers(self.h, C.c_int64(len(on)), ptr(cells), ptr(Xr)))
        self._nrecv = len(on)
        return on

    def sample(self, which: str = "u"):
        """u_h (or v_h) at the receivers, evaluated on the device from the resident vector."""
        out = np.zeros(self._nrecv, dtype=self.data.dtype)
        w = _abi.FUS_U if which == "u" else _abi.FUS_V
        check(lib().fus_model_sample(self.h, C.c_int(w), ptr(out), C.c_int(_abi.FUS_HOST)))
        return out

    def record(self, every: int, capacity: int, which: str = "u"):
        """Sample the receivers after every ``every``-th step of rk() / rk4_steps() into a device buffer."""
        w = _abi.FUS_U if which == "u" else _abi.FUS_V
        check(lib().fus_model_record(self.h, C.c_int(w), C.c_int(every), C.c_int64(capacity)))

    def records(self):
        """(times [nrec], samples [nrec, npts]) recorded so far."""
        n = C.c_int64()
        check(lib().fus_model_get_records(self.h, None, None, C.byref(n)))
        out = np.zeros((n.value, self._nrecv), dtype=self.data.dtype)
        times = np.zeros(n.value)
        check(lib().fus_model_get_records(self.h, ptr(out), ptr(times), C.byref(n)))
        return times, out

    def u_sol(self):
        self._pull()
        return self.u_n

    def number_of_dofs(self):
        return self.V.dofmap.index_map.size_global

    def mass_vector(self):
        out = np.empty(self.data.ndofs, dtype=self.data.dtype)
        check(lib().fus_model_get_mass(self.h, ptr(out)))
        return out

    def close(self):
        """Frees the model and the operator data it created (G, dofmaps, partial slab)."""
        if self.h:
            lib().fus_model_destroy(self.h)
            self.h = C.c_void_p()
        self.data.close()


class LinearSpectralExplicit(_SpectralExplicit):
    """``LinearSpectralExplicit(mesh, meshtags, k, c0, rho0, freq0, p0, s0, rk_order, dt)``
    (_linear.py:267).  ``meshtags`` carries boundary facets as (cell, local facet) pairs with
    ``values`` 1 = source, 2 = absorbing (what ``compute_integration_domains`` yields for the
    tagged facets, Linear.hpp:113-118)."""

    _kind = _abi.FUS_LINEAR

    def __init__(self, mesh, meshtags, k, c0, rho0, freq0, p0, s0, rk_order=4, dt=None, V=None,
                 ctx: Context | None = None):
        self._create(mesh, meshtags, k, c0, rho0, None, freq0, p0, s0, rk_order, dt, V, ctx)


class LossySpectralExplicit(_SpectralExplicit):
    """``LossySpectralExplicit(mesh, meshtags, k, c0, rho0, delta0, freq0, p0, s0, rk_order, dt)``
    (python/src/fenicsxfus/_lossy.py:21-23; C++ ``LossySpectral3D``, Lossy.hpp:56-62).
    ``delta0`` = diffusivity of sound (DG0).  The reference has two variants of the boundary forms:
    ``forms="cpp"`` (default) is the C++ one -- absorbing and delta-mass terms on every listed
    boundary facet (BM7-SC1/forms.py:37-42) and the source doubled (Lossy.hpp:216-220);
    ``forms="python"`` is the Python package's -- those terms on tag 2 only, source not doubled
    (_lossy.py:107-128, :186-189)."""

    _kind = _abi.FUS_LOSSY

    def __init__(self, mesh, meshtags, k, c0, rho0, delta0, freq0, p0, s0, rk_order=4, dt=None, V=None,
                 ctx: Context | None = None, forms="cpp"):
        self._create(mesh, meshtags, k, c0, rho0, delta0, freq0, p0, s0, rk_order, dt, V, ctx, forms=forms)


class WesterveltSpectralExplicit(_SpectralExplicit):
    """``WesterveltSpectralExplicit(mesh, meshtags, k, c0, rho0, delta0, beta0, freq0, p0, s0,
    rk_order, dt)`` (python/src/fenicsxfus/_westervelt.py:21-23; C++ ``WesterveltSpectral3D``,
    Westervelt.hpp:58-67).  ``beta0`` = coefficient of nonlinearity (DG0); ``forms`` as in
    :class:`LossySpectralExplicit` (_westervelt.py:107-142, :215)."""

    _kind = _abi.FUS_WESTERVELT

    def __init__(self, mesh, meshtags, k, c0, rho0, delta0, beta0, freq0, p0, s0, rk_order=4, dt=None, V=None,
                 ctx: Context | None = None, forms="cpp"):
        self._create(mesh, meshtags, k, c0, rho0, delta0, freq0, p0, s0, rk_order, dt, V, ctx, beta0=beta0,
                     forms=forms)


def compute_diffusivity_of_sound(w0: float, c0: float, alpha: float) -> float:
    """delta = 2 alpha c0^3 / w0^2 with alpha in Np/m (the C++ helper, Lossy.hpp:375-379).  The
    Python package's helper of the same name takes dB/m: :func:`fenicsxfus_amd.utils.compute_diffusivity_of_sound`."""
    return 2 * alpha * c0**3 / w0**2


def group_finish_setup(models):
    """In-process transport: add the sharers' parts of the mass / boundary vectors (tests)."""
    arr = (C.c_void_p * len(models))(*[m.h for m in models])
    check(lib().fus_group_finish_setup(arr, C.c_int(len(models))))


def group_rk4_steps(models, t0: float, dt: float, nsteps: int):
    """In-process transport: advance all slab models in lock-step (tests)."""
    arr = (C.c_void_p * len(models))(*[m.h for m in models])
    check(lib().fus_group_rk4_steps(arr, C.c_int(len(models)), C.c_double(t0), C.c_double(dt), C.c_int64(nsteps)))
