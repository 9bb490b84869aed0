"""alias of ``gpsa.models.gpsa`` (gpsa/models/gpsa.py:9): ``from gpsa.models.gpsa import GPSA``"""
from spatial_alignment_amd.models.gpsa import GPSA  # noqa: F401
