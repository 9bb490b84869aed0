"""ctypes binding of libgpsa_hip.so (the C ABI declared in include/gpsa_hip.h).

The product path has no CPU fallback: if the shared library is missing, or no HIP device is
visible when an op is called, this raises.  (tests/ may inject a fake ``ops`` object to exercise the
host logic on CPU; nothing in this package does.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpsa_hip.so")

_vp, _i, _ll, _d = C.c_void_p, C.c_int, C.c_longlong, C.c_double

# name -> (restype, argtypes); mirrors include/gpsa_hip.h exactly
SIGNATURES = {
    "gpsa_version": (_i, []),
    "gpsa_build_arch": (C.c_char_p, []),
    "gpsa_kmat": (_i, [_i, _i, _i, _vp, _i, _vp, _ll, _i, _vp, _vp, _d, _vp, _vp]),
    "gpsa_kmat_bwd_workspace": (_ll, [_i, _i, _ll, _i]),
    "gpsa_kmat_bwd": (_i, [_i, _i, _i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_gemm_workspace": (_ll, [_i, _i, _i, _i, _i]),
    "gpsa_gemm": (_i, [_i, _i, _i, _i, _i, _ll, _d, _vp, _ll, _ll, _vp, _ll, _ll, _d, _vp, _ll, _ll,
                       _i, _i, _vp, _ll, _vp]),
    "gpsa_omega_fwd": (_i, [_vp, _i, _i, _d, _vp, _vp]),
    "gpsa_omega_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gpsa_chol_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "gpsa_tri_inv_f64": (_i, [_vp, _vp, _i, _i, _vp]),
    "gpsa_chol_inv_f64": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "gpsa_chol_inv_blocked_workspace": (_ll, [_i, _i]),
    "gpsa_chol_inv_blocked_f64": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_workspace": (_ll, [_i, _i, _ll, _i]),
    "gpsa_quadform_fwd": (_i, [_i, _i, _vp, _vp, _i, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_bwd_alpha": (_i, [_i, _i, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_fwd_keep": (_i, [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpsa_quadform_bwd_alpha_kept": (_i, [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp]),
    "gpsa_quadform_bwd_omega": (_i, [_i, _i, _vp, _vp, _i, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_whiten_workspace": (_ll, [_i]),
    "gpsa_whiten_f64": (_i, [_vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_panel_mm": (_i, [_i, _i, _i, _vp, _vp, _i, _ll, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_col_axpy": (_i, [_i, _vp, _vp, _vp, _d, _i, _ll, _vp, _vp]),
    "gpsa_data_sample_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp]),
    "gpsa_data_sample_bwd": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_warp_sample_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpsa_warp_sample_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _ll, _vp]),
    "gpsa_mean_resid_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp]),
    "gpsa_mean_resid_bwd": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "gpsa_loglik_fwd": (_i, [_vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_loglik_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_bdot": (_i, [_i, _vp, _ll, _vp, _ll, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_add_diag": (_i, [_i, _vp, _i, _i, _d, _vp]),
    "gpsa_mvn_kl_fwd": (_i, [_vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _i, _i, _vp, _vp]),
    "gpsa_mvn_kl_bwd": (_i, [_vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "gpsa_mvn_kl_grouped_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "gpsa_mvn_kl_grouped_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp,
                                     _vp]),
    "gpsa_elbo_fwd": (_i, [_vp, _i, _vp, _i, _d, _vp, _vp]),
    "gpsa_elbo_bwd": (_i, [_vp, _i, _i, _d, _vp, _vp, _vp]),
    "gpsa_kmeans_workspace": (_ll, [_ll, _i, _i]),
    "gpsa_kmeans_assign": (_i, [_vp, _ll, _i, _vp, _i, _vp, _vp, _vp]),
    "gpsa_kmeans_update": (_i, [_vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _ll, _vp]),
}

_lib = None


class GpsaHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach argtypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpsaHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


GPSA_EINVAL, GPSA_EWORKSPACE, GPSA_EUNSUPPORTED = -1, -2, -3  # include/gpsa_hip.h


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "workspace too small", -3: "unsupported size"}.get(
            rc, f"hipError_t {rc}"
        )
        raise GpsaHipError(f"{what} failed: {kind}")
