"""GPSA (Gaussian Process Spatial Alignment) training hot path for AMD MI355X (gfx950).

Same public names as the reference package (gpsa/__init__.py:1-10) for the parts on the hot path.
"""
from .kernels import matern12_kernel, matern32_kernel, rbf_kernel
from .models import GPSA, VariationalGPSA
from .util import (
    ConvergenceChecker,
    LossNotDecreasingChecker,
    get_st_coordinates,
    polar_warp,
    rbf_kernel_numpy,
)

__all__ = [
    "GPSA",
    "VariationalGPSA",
    "rbf_kernel",
    "matern12_kernel",
    "matern32_kernel",
    "rbf_kernel_numpy",
    "polar_warp",
    "get_st_coordinates",
    "LossNotDecreasingChecker",
    "ConvergenceChecker",
]
__version__ = "0.1.0"
