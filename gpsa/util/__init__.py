"""alias of the reference's ``gpsa.util`` (gpsa/util/__init__.py:1 exports ``rbf_kernel_numpy``)"""
from spatial_alignment_amd.util.util import rbf_kernel_numpy  # noqa: F401
