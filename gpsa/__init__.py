"""Drop-in alias: ``from gpsa import VariationalGPSA, rbf_kernel, ...`` resolves to the MI355X
implementation (spatial_alignment_amd), mirroring the reference's export list (gpsa/__init__.py:1-17); the
sub-packages ``gpsa.models``, ``gpsa.util``, ``gpsa.plotting`` resolve too (the four plotting callbacks as names only)."""
from spatial_alignment_amd import *  # noqa: F401,F403
from spatial_alignment_amd import __all__ as _core
from gpsa.plotting import (  # noqa: F401
    callback_oned,
    callback_twod,
    callback_twod_aligned_only,
    callback_twod_multimodal,
)

__all__ = list(_core) + ["callback_oned", "callback_twod", "callback_twod_aligned_only", "callback_twod_multimodal"]
