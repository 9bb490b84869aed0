"""Drop-in alias: ``from gpsa import VariationalGPSA, rbf_kernel, ...`` resolves to the MI355X
implementation (spatial_alignment_amd), mirroring the reference's export list (gpsa/__init__.py:1-10)."""
from spatial_alignment_amd import *  # noqa: F401,F403
from spatial_alignment_amd import __all__  # noqa: F401
