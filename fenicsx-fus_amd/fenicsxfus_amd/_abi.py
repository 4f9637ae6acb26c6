"""ctypes binding of libfusmi's C ABI (include/fusmi.h).  Fails loudly when the HIP extension is
missing or no device is present -- there is no CPU path behind these calls."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FUSMI_LIB", os.path.join(_HERE, "libfusmi.so"))  # override: experiments only

FUS_F32, FUS_F64 = 0, 1
FUS_HOST, FUS_DEVICE = 0, 1
FUS_LINEAR, FUS_LOSSY, FUS_WESTERVELT = 0, 1, 2
FUS_U, FUS_V = 0, 1

# every symbol include/fusmi.h declares
SYMBOLS = [
    "fus_last_error", "fus_version", "fus_init", "fus_finalize", "fus_synchronize", "fus_set_option",
    "fus_comm_unique_id", "fus_comm_init", "fus_comm_selftest", "fus_op_create", "fus_op_destroy", "fus_stiffness_apply",
    "fus_mass_apply", "fus_op_get_geometry", "fus_op_get_tables", "fus_op_info", "fus_op_is_affine", "fus_op_geometry_mode", "fus_op_uses_mfma", "fus_op_uses_diag_metric", "fus_op_uses_mfma4", "fus_op_uses_pack32", "fus_op_hmin", "fus_op_norm2", "fus_comm_allreduce", "fus_facet_diag",
    "fus_op_set_neighbours", "fus_model_create", "fus_model_destroy", "fus_model_set_rk_order", "fus_model_init", "fus_model_rk4",
    "fus_model_rk4_steps", "fus_model_get", "fus_model_set", "fus_model_get_mass", "fus_model_ndofs",
    "fus_profile_enable", "fus_profile_get", "fus_measure_bandwidth", "fus_layout_check", "fus_layout_check_ex", "fus_comm_init_local",
    "fus_group_finish_setup", "fus_group_rk4_steps", "fus_op_halo_layout", "fus_op_halo_buffers",
    "fus_model_setup_count", "fus_model_setup_pack", "fus_model_setup_unpack", "fus_model_setup_finish",
    "fus_model_stage_begin", "fus_model_stage_end",
    "fus_model_set_receivers", "fus_model_sample", "fus_model_record", "fus_model_get_records",
]


class FusError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FusError(
                f"{LIB_PATH} not found: build the HIP extension first (python fenicsx-fus_amd/build.py)")
        _lib = C.CDLL(LIB_PATH)
        _lib.fus_last_error.restype = C.c_char_p
        _lib.fus_model_ndofs.restype = C.c_int64
    return _lib


def check(code: int) -> None:
    if code != 0:
        raise FusError(f"libfusmi error {code}: {lib().fus_last_error().decode()}")


def ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float64:
        return FUS_F64
    if dt == np.float32:
        return FUS_F32
    raise FusError(f"unsupported scalar type {dt}")


class Context:
    """One per GPU (fus_init)."""

    def __init__(self, device: int = 0, block_elems: int | None = None, waves: int | None = None,
                 deterministic: bool | None = None, geometry: str | None = None):
        self.h = C.c_void_p()
        check(lib().fus_init(C.c_int(device), C.byref(self.h)))
        if block_elems is not None:
            self.set_option("block_elems", block_elems)
        if waves is not None:
            self.set_option("waves", waves)
        if deterministic is not None:
            self.set_option("deterministic", int(deterministic))
        if geometry is not None:   # "auto": per-cell factors on affine meshes; "stream": always stream G;
            # "trilinear": J and G recomputed per point from the cell's trilinear map (first-order hexahedra)
            self.set_option("geometry", {"auto": 0, "stream": 1, "trilinear": 2}[geometry])
        self.rank, self.nranks = 0, 1

    def set_option(self, key: str, value: int):
        check(lib().fus_set_option(self.h, key.encode(), C.c_int64(value)))

    def synchronize(self):
        check(lib().fus_synchronize(self.h))

    @staticmethod
    def _prefer_resident_rccl():
        """If PyTorch (which bundles its own librccl) is loaded, bind to that copy."""
        import sys

        torch = sys.modules.get("torch")
        if torch is not None and "FUSMI_RCCL" not in os.environ:
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(cand):
                os.environ["FUSMI_RCCL"] = cand

    def comm_init(self, rank: int, nranks: int, unique_id: bytes | None):
        self._prefer_resident_rccl()
        buf = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id else None
        check(lib().fus_comm_init(self.h, C.c_int(rank), C.c_int(nranks), buf))
        self.rank, self.nranks = rank, nranks

    def allreduce(self, values, op: str = "sum"):
        """In-place all-reduce of host doubles over the RCCL ranks (MPI_Reduce + MPI_Bcast of the examples'
        mains, linear_planewave2d_1/main.cpp:67-68); returns the reduced numpy array."""
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        check(lib().fus_comm_allreduce(self.h, v.ctypes.data_as(C.c_void_p), C.c_int(len(v)),
                                       C.c_int({"sum": 0, "min": 1, "max": 2}[op])))
        return v

    def comm_selftest(self, n: int = 1 << 16):
        self._prefer_resident_rccl()
        check(lib().fus_comm_selftest(self.h, C.c_int64(n)))

    @staticmethod
    def unique_id() -> bytes:
        Context._prefer_resident_rccl()
        buf = (C.c_char * 128)()
        check(lib().fus_comm_unique_id(buf))
        return bytes(buf)

    def init_external(self, rank: int, nranks: int):
        """External transport: the caller exchanges the packed interface values itself (GPU-aware MPI in
        the reference's setting); see fusmi.h.  No RCCL communicator is created."""
        self.set_option("external_transport", 1)
        check(lib().fus_comm_init(self.h, C.c_int(rank), C.c_int(nranks), None))
        self.rank, self.nranks = rank, nranks

    @staticmethod
    def init_local_group(ctxs):
        """In-process transport: these contexts (one GPU) act as ranks 0..n-1 (tests only)."""
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        check(lib().fus_comm_init_local(arr, C.c_int(len(ctxs))))
        for i, c in enumerate(ctxs):
            c.rank, c.nranks = i, len(ctxs)

    def profile_enable(self, on: bool = True):
        check(lib().fus_profile_enable(self.h, C.c_int(int(on))))   # 1: all kernels, 2: block operator only

    def measure_bandwidth(self, nbytes: int = 1 << 29, reps: int = 5) -> float:
        """Streaming copy bandwidth of the device in GB/s (16-byte non-temporal copy of nbytes; read + written)."""
        g = C.c_double()
        check(lib().fus_measure_bandwidth(self.h, C.c_int64(nbytes), C.c_int(reps), C.byref(g)))
        return g.value

    def profile_get(self, name: str):
        ms, n = C.c_double(), C.c_int64()
        check(lib().fus_profile_get(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self.h:
            lib().fus_finalize(self.h)
            self.h = C.c_void_p()


def layout_check(P, tensor_dofmap, centroids, block_elems=64, waves=4, tdim=None, force_shared=None):
    """Host-only: run the block partitioner + verifier; returns the 8 statistics of fus_op_info.
    tdim defaults to what the dofmap width says; force_shared: optional bool mask of interface dofs."""
    dm = np.ascontiguousarray(tensor_dofmap, dtype=np.int32)
    if tdim is None:
        tdim = 3 if dm.shape[1] == (P + 1) ** 3 else 2
    cen = np.ascontiguousarray(centroids, dtype=np.float64)
    if cen.shape[1] == 2:
        cen = np.hstack([cen, np.zeros((cen.shape[0], 1))])
    ndofs = int(dm.max()) + 1
    fs = None if force_shared is None else np.ascontiguousarray(force_shared, dtype=np.uint8)
    assert fs is None or fs.shape[0] == ndofs
    out = (C.c_int64 * 8)()
    check(lib().fus_layout_check_ex(C.c_int(tdim), C.c_int(P), C.c_int64(dm.shape[0]), C.c_int64(ndofs), ptr(dm),
                                    ptr(cen), C.c_int(block_elems), C.c_int(waves), ptr(fs), out))
    return list(out)
