"""alias of the reference's ``gpsa.models`` (gpsa/models/__init__.py): the MI355X classes"""
from spatial_alignment_amd.models import GPSA, VariationalGPSA  # noqa: F401
