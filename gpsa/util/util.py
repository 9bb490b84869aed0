"""alias of ``gpsa.util.util`` for the names on or next to the hot path (gpsa/util/util.py:8-87, 112-153, 257-278):
the three covariance plug-ins, their numpy twin, the two coordinate helpers and the two convergence checkers.  The
count-data helpers of that file (size factors, deviance / Pearson residuals, ``make_pinwheel``: util.py:91-255) are
preprocessing for the experiment scripts - out of scope - and are not aliased."""
from spatial_alignment_amd.kernels import matern12_kernel, matern32_kernel, rbf_kernel  # noqa: F401
from spatial_alignment_amd.util.util import (  # noqa: F401
    ConvergenceChecker,
    LossNotDecreasingChecker,
    compute_distance,
    get_st_coordinates,
    polar_warp,
    rbf_kernel_numpy,
)
