"""Final-state output (post-processing, host side): a degree-P GLL field written as a VTK
unstructured grid (.vtu, XML, ASCII or raw-appended binary) that ParaView / VisIt open directly.

The reference writes its solutions with DOLFINx's ``VTXWriter`` (ADIOS2; e.g.
cpp/fenicsx-sf/benchmarks/PH1/BM7-SC1/main.cpp:140-147, python/examples); ADIOS2 is not available
here, so the same data -- the field values at the element's GLL nodes -- goes out in the plain VTK
XML format: every degree-P element is split into P^tdim first-order sub-cells on its GLL node
lattice (exact at the nodes, which is all a nodal field holds), points are the global DOF
coordinates, point data are the DOF values.  Works for hexahedra and quadrilaterals, ``BoxMesh`` /
``HexMesh`` / ``QuadMesh`` function spaces (anything with ``tensor_dofmap``, ``nodes1d``, ``P``,
``tabulate_dof_coordinates()``)."""
from __future__ import annotations

import base64

import numpy as np

VTK_QUAD, VTK_HEXAHEDRON = 9, 12


def subcell_connectivity(V) -> np.ndarray:
    """[ncells * P^tdim, 2^tdim] global DOF ids of the first-order sub-cells in VTK vertex order."""
    N = V.P + 1
    tdim = V.mesh.topology.dim
    dm = np.asarray(V.tensor_dofmap)
    order = np.argsort(np.asarray(V.nodes1d), kind="stable")      # local 1-D index of the k-th smallest node
    idx = np.arange(N ** tdim).reshape((N,) * tdim)
    idx = idx[np.ix_(*([order] * tdim))]                           # lattice in increasing coordinates
    if tdim == 3:
        a, b, c = np.meshgrid(*([np.arange(N - 1)] * 3), indexing="ij")
        a, b, c = a.ravel(), b.ravel(), c.ravel()
        # VTK hexahedron: bottom face counter-clockwise, then top face (x = tensor index 0)
        loc = np.stack([idx[a, b, c], idx[a + 1, b, c], idx[a + 1, b + 1, c], idx[a, b + 1, c],
                        idx[a, b, c + 1], idx[a + 1, b, c + 1], idx[a + 1, b + 1, c + 1], idx[a, b + 1, c + 1]], axis=1)
    else:
        a, b = np.meshgrid(*([np.arange(N - 1)] * 2), indexing="ij")
        a, b = a.ravel(), b.ravel()
        loc = np.stack([idx[a, b], idx[a + 1, b], idx[a + 1, b + 1], idx[a, b + 1]], axis=1)
    return dm[:, loc].reshape(-1, loc.shape[1])


def write_vtu(path: str, V, fields: dict, binary: bool = True) -> None:
    """``fields``: name -> array [ndofs] (or object with ``.x.array``, like the models' ``u_n``)."""
    x = np.asarray(V.tabulate_dof_coordinates(), dtype=np.float64)
    if x.shape[1] == 2:
        x = np.hstack([x, np.zeros((x.shape[0], 1))])
    conn = subcell_connectivity(V).astype(np.int64)
    nv = conn.shape[1]
    offsets = (np.arange(conn.shape[0], dtype=np.int64) + 1) * nv
    types = np.full(conn.shape[0], VTK_HEXAHEDRON if nv == 8 else VTK_QUAD, dtype=np.uint8)
    data = {k: np.asarray(getattr(getattr(v, "x", v), "array", v)) for k, v in fields.items()}
    for k, v in data.items():
        if v.shape[0] != x.shape[0]:
            raise ValueError(f"field {k!r} has {v.shape[0]} values, the space has {x.shape[0]} dofs")

    vtk_type = {"float64": "Float64", "float32": "Float32", "int64": "Int64", "uint8": "UInt8"}

    def array(name, a, ncomp=1):
        a = np.ascontiguousarray(a)
        head = f'<DataArray type="{vtk_type[str(a.dtype)]}" Name="{name}" NumberOfComponents="{ncomp}" '
        if binary:   # inline base64 with a UInt64 byte-count header (header_type="UInt64")
            raw = np.uint64(a.nbytes).tobytes() + a.tobytes()
            return head + 'format="binary">' + base64.b64encode(raw).decode() + "</DataArray>\n"
        return head + 'format="ascii">' + " ".join(repr(t) for t in a.ravel().tolist()) + "</DataArray>\n"

    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian" '
                'header_type="UInt64">\n<UnstructuredGrid>\n')
        f.write(f'<Piece NumberOfPoints="{x.shape[0]}" NumberOfCells="{conn.shape[0]}">\n<Points>\n')
        f.write(array("Points", x, 3))
        f.write("</Points>\n<Cells>\n")
        f.write(array("connectivity", conn.ravel()))
        f.write(array("offsets", offsets))
        f.write(array("types", types))
        f.write("</Cells>\n<PointData>\n")
        for k, v in data.items():
            f.write(array(k, v))
        f.write("</PointData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")


def read_vtu(path: str):
    """Minimal reader of what :func:`write_vtu` writes (round-trip tests): points, connectivity,
    types, point data."""
    import xml.etree.ElementTree as ET

    np_type = {"Float64": np.float64, "Float32": np.float32, "Int64": np.int64, "UInt8": np.uint8}
    root = ET.parse(path).getroot()
    out = {}
    for da in root.iter("DataArray"):
        dt = np_type[da.get("type")]
        if da.get("format") == "binary":
            raw = base64.b64decode(da.text.strip())
            n = int(np.frombuffer(raw[:8], dtype=np.uint64)[0])
            a = np.frombuffer(raw[8:8 + n], dtype=dt)
        else:
            a = np.array(da.text.split(), dtype=dt)
        nc = int(da.get("NumberOfComponents", "1"))
        out[da.get("Name")] = a.reshape(-1, nc) if nc > 1 else a
    return out
