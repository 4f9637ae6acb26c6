"""fenicsxfus_amd -- MI355X-native drop-in for the fenicsx-fus spectral-element hot path
(sum-factorised operators + RK4 stage update + halo exchange).  The compute path is libfusmi.so
(hand-written HIP for gfx950 behind the C ABI in include/fusmi.h); this package is the host-side
mirror of the reference's operator/model interface."""
from . import output, tables, utils  # noqa: F401
from ._abi import Context, FusError, layout_check  # noqa: F401
from .mesh import BoxMesh, CellFunction, FacetTags, Function, FunctionSpace, tag_box_boundary  # noqa: F401
from .models import (LinearSpectralExplicit, LossySpectralExplicit, WesterveltSpectralExplicit,  # noqa: F401
                     compute_diffusivity_of_sound,
                     group_finish_setup, group_rk4_steps)
from .operators import (MassSpectral2D, MassSpectral3D, SpectralOperatorData, StiffnessSpectral2D,  # noqa: F401
                        StiffnessSpectral3D)
from .unstructured import HexFunctionSpace, HexMesh, QuadMesh, read_xdmf_hex_mesh, read_xdmf_mesh  # noqa: F401,E402
