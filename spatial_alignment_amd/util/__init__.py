from .util import (  # noqa: F401
    ConvergenceChecker,
    LossNotDecreasingChecker,
    compute_distance,
    get_st_coordinates,
    polar_warp,
    rbf_kernel_numpy,
)
