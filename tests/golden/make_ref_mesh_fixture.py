"""Extracts the reference's own operator-test mesh (a data file its tests hold:
cpp/fenicsx-sf/tests/test_operators3d/mesh.xdmf + mesh.h5, 6312 Gmsh hexahedra of the unit cube,
read there by io::XDMFFile in the commented block of main.cpp:40-48) into
tests/golden/ref_test_operators3d_mesh.npz with this repository's own XDMF/HDF5 reader.  The
fixture holds DATA only (vertex coordinates, cell-vertex lists in the file's XDMF/VTK order, facet
vertex lists and tag values); run where /root/reference exists."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "fenicsx-fus_amd"))
from fenicsxfus_amd.hdf5_lite import H5File  # noqa: E402

src = "/root/reference/cpp/fenicsx-sf/tests/test_operators3d/mesh.h5"
f = H5File(src)
np.savez_compressed(
    os.path.join(HERE, "ref_test_operators3d_mesh.npz"),
    geometry=f["/Mesh/hex/geometry"],
    topology_vtk=f["/Mesh/hex/topology"].astype(np.int32),
    cell_values=f["/MeshTags/hex_cells/Values"].ravel().astype(np.int32),
    facet_topology=f["/MeshTags/hex_facets/topology"].astype(np.int32),
    facet_values=f["/MeshTags/hex_facets/Values"].ravel().astype(np.int32),
)
print("wrote ref_test_operators3d_mesh.npz")

# The naive 2-D operator test's mesh (cpp/fenicsx-sf-naive/tests/test_operators2d/mesh_1: 265 Gmsh
# quadrilaterals of the unit square, read by main.cpp:46-53), again data only.
src2 = "/root/reference/cpp/fenicsx-sf-naive/tests/test_operators2d/mesh_1/mesh.h5"
f2 = H5File(src2)
np.savez_compressed(
    os.path.join(HERE, "ref_test_operators2d_mesh.npz"),
    geometry=f2["/Mesh/quad/geometry"],
    topology_vtk=f2["/Mesh/quad/topology"].astype(np.int32),
    cell_values=f2["/MeshTags/quad_cells/Values"].ravel().astype(np.int32),
    facet_topology=f2["/MeshTags/quad_facets/topology"].astype(np.int32),
    facet_values=f2["/MeshTags/quad_facets/Values"].ravel().astype(np.int32),
)
print("wrote ref_test_operators2d_mesh.npz")

# ... and its second-order twin (mesh_2: the same 265 quadrilaterals with 9 nodes each, G = 2 in
# main.cpp:31), data only.
src3 = "/root/reference/cpp/fenicsx-sf-naive/tests/test_operators2d/mesh_2/mesh.h5"
f3 = H5File(src3)
np.savez_compressed(
    os.path.join(HERE, "ref_test_operators2d_mesh2.npz"),
    geometry=f3["/Mesh/quad/geometry"],
    topology_vtk=f3["/Mesh/quad/topology"].astype(np.int32),
    cell_values=f3["/MeshTags/quad_cells/Values"].ravel().astype(np.int32),
    facet_topology=f3["/MeshTags/quad_facets/topology"].astype(np.int32),
    facet_values=f3["/MeshTags/quad_facets/Values"].ravel().astype(np.int32),
)
print("wrote ref_test_operators2d_mesh2.npz")
