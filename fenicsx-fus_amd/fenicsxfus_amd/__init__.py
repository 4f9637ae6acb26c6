"""fenicsxfus_amd -- MI355X-native drop-in for the fenicsx-fus spectral-element hot path."""
from . import tables  # noqa: F401
from .mesh import BoxMesh, CellFunction, FacetTags, Function, FunctionSpace, tag_box_boundary  # noqa: F401
