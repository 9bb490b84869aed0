"""alias of ``gpsa.models.vgpsa`` (gpsa/models/vgpsa.py:14): ``from gpsa.models.vgpsa import VariationalGPSA``"""
from spatial_alignment_amd.models.vgpsa import VariationalGPSA  # noqa: F401
