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
    "gpsa_source_hash": (C.c_char_p, []),
    "gpsa_kmat": (_i, [_i, _i, _i, _vp, _i, _vp, _ll, _i, _vp, _vp, _d, _vp, _vp]),
    "gpsa_kmat_bwd_workspace": (_ll, [_i, _i, _ll, _i]),
    "gpsa_kmat_bwd": (_i, [_i, _i, _i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_gemm_workspace": (_ll, [_i, _i, _i, _i, _i]),
    "gpsa_gemm": (_i, [_i, _i, _i, _i, _i, _ll, _d, _vp, _ll, _ll, _vp, _ll, _ll, _d, _vp, _ll, _ll,
                       _i, _i, _vp, _ll, _vp]),
    "gpsa_omega_fwd": (_i, [_vp, _i, _i, _d, _vp, _vp]),
    "gpsa_omega_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gpsa_omega_fwd2": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _d, _vp]),
    "gpsa_omega_bwd2": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "gpsa_chol_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "gpsa_tri_inv_f64": (_i, [_vp, _vp, _i, _i, _vp]),
    "gpsa_chol_inv_f64": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "gpsa_experiment_split_bf16_product": (_i, [_vp, _vp, _i, _ll, _i, _i, _vp, _vp]),
    "gpsa_experiment_split_bf16_rate": (_i, [_i, _i, _vp, _vp]),
    "gpsa_chol_inv_sel_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gpsa_chol_inv_blocked_workspace": (_ll, [_i, _i]),
    "gpsa_chol_inv_blocked_f64": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_workspace": (_ll, [_i, _i, _ll, _i]),
    "gpsa_quadform_fwd": (_i, [_i, _i, _vp, _vp, _i, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_bwd_alpha": (_i, [_i, _i, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_fwd_keep": (_i, [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpsa_quadform_bwd_alpha_kept": (_i, [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp]),
    "gpsa_quadform_bwd_omega": (_i, [_i, _i, _vp, _vp, _i, _ll, _i, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_bwd_omega_takes_delta": (_i, [_i, _ll]),
    "gpsa_quadform_bwd_omega_delta_f32": (_i, [_i, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _d, _vp, _ll, _vp]),
    "gpsa_whiten_workspace": (_ll, [_i]),
    "gpsa_whiten_f64": (_i, [_vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_panel_mm": (_i, [_i, _i, _i, _vp, _vp, _i, _ll, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_col_axpy": (_i, [_i, _vp, _vp, _vp, _d, _i, _ll, _vp, _vp]),
    "gpsa_data_sample_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp]),
    "gpsa_data_sample_bwd": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _ll, _vp]),
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

MAX_MODS = 4  # GPSA_MAX_MODS


class StepDesc(C.Structure):
    """gpsa_step_desc (include/gpsa_hip.h)"""
    _fields_ = [
        ("n_views", _i), ("n_dims", _i), ("n_mods", _i), ("n_samples", _i),
        ("m_x", _i), ("m_g", _i), ("kind_warp", _i), ("kind_data", _i),
        ("n_latent", _i * MAX_MODS), ("n_out", _i * MAX_MODS), ("has_lmc", _i * MAX_MODS),
        ("n_rows", _ll * MAX_MODS), ("s_test", _i), ("n_test", _ll * MAX_MODS), ("want_kl", _i),
        ("view_fixed", C.POINTER(_i)), ("view_rows", C.POINTER(_ll)), ("keep_budget_bytes", _ll),
        ("exact_inducing_grad", _i), ("kl_own_lo", _i), ("kl_own_hi", _i),
    ]


class StepParams(C.Structure):
    """gpsa_step_params / gpsa_step_param_grads: the same field order, device pointers"""
    _fields_ = [
        ("Xtilde", _vp), ("delta_G", _vp), ("Omega_sqt_G", _vp), ("warp_ls", _vp), ("warp_var", _vp),
        ("slopes", _vp), ("intercepts", _vp), ("Gtilde", _vp), ("data_ls", _vp), ("data_var", _vp),
        ("Omega_sqt_F", _vp * MAX_MODS), ("delta_F", _vp * MAX_MODS), ("W", _vp * MAX_MODS),
    ]


class StepParamGrads(C.Structure):
    _fields_ = [
        ("Xtilde", _vp), ("delta_G", _vp), ("Omega_sqt_G", _vp), ("warp_ls", _vp), ("warp_var", _vp),
        ("Gtilde", _vp), ("data_ls", _vp), ("data_var", _vp),
        ("Omega_sqt_F", _vp * MAX_MODS), ("delta_F", _vp * MAX_MODS), ("W", _vp * MAX_MODS),
    ]


class StepIO(C.Structure):
    _fields_ = [
        ("X", _vp * MAX_MODS), ("eps_G", _vp), ("eps_F", _vp * MAX_MODS), ("G_test", _vp * MAX_MODS),
        ("eps_F_test", _vp * MAX_MODS), ("G_means", _vp * MAX_MODS), ("G_samples", _vp * MAX_MODS),
        ("F_latent", _vp * MAX_MODS), ("F_obs", _vp * MAX_MODS), ("F_latent_test", _vp * MAX_MODS),
        ("F_obs_test", _vp * MAX_MODS), ("mu_z", _vp), ("kl", _vp), ("flag", _vp), ("keep_products", _i),
        ("reuse_mm", _i), ("fuse_elbo", _i), ("Y", _vp * MAX_MODS), ("noise_u", _vp * MAX_MODS),
        ("ll_part", _vp * MAX_MODS), ("F_fused_T", _vp * MAX_MODS), ("bwd_acc", _vp), ("bwd_acc_mode", _i),
        ("f_event", _vp),
    ]


class StepOutGrads(C.Structure):
    _fields_ = [
        ("dG_means", _vp * MAX_MODS), ("dG_samples", _vp * MAX_MODS), ("dF_latent", _vp * MAX_MODS),
        ("dF_obs", _vp * MAX_MODS), ("dF_latent_test", _vp * MAX_MODS), ("dF_obs_test", _vp * MAX_MODS),
        ("dkl", _vp), ("gloss", _vp),
    ]


_pp = C.POINTER(_vp)  # host array of device pointers

SIGNATURES.update({
    "gpsa_kmat_batched": (_i, [_i, _vp, _ll, _i, _vp, _ll, _ll, _i, _vp, _vp, _i, C.POINTER(_ll), _i, _d, _vp, _ll, _vp]),
    "gpsa_kmat_bwd_batched_workspace": (_ll, [_i, _ll, _i, _i]),
    "gpsa_kmat_bwd_batched": (_i, [_i, _vp, _ll, _i, _vp, _ll, _ll, _i, _vp, _vp, _i, C.POINTER(_ll), _i, _vp, _ll, _i,
                                   _vp, _ll, _vp, _vp, _ll, _vp]),
    "gpsa_kmat_bwd_x64": (_i, [_i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_kmat_bwd_x64_f64": (_i, [_i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_kmat_bwd_x64_f64_axpy": (_i, [_i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _ll,
                                        _vp]),
    "gpsa_exact_dkuu_workspace": (_ll, [_i, _ll]),
    "gpsa_exact_dkuu_f64": (_i, [_vp, _vp, _vp, _i, _ll, _vp, _vp, _ll, _vp]),
    "gpsa_longk_f64_workspace": (_ll, [_i, _ll, _i]),
    "gpsa_longk_f64": (_i, [_i, _vp, _vp, _vp, _i, _i, _ll, _ll, _i, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_kmat_bwd_x64_axpy": (_i, [_i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _ll,
                                    _vp]),
    "gpsa_thin_update_f32": (_i, [_vp, _i, _i, _vp, _ll, _vp, _vp]),
    "gpsa_whiten_f64_dual": (_i, [_vp, _vp, _i, _ll, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_whiten_gen_f64_dual": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _ll, _vp, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_whiten_axpy_f32": (_i, [_vp, _vp, _i, _ll, _vp, _vp, _d, _vp, _vp, _ll, _vp]),
    "gpsa_whiten_batched_f64": (_i, [_vp, _ll, _vp, _i, _ll, _ll, _vp, _vp, _i, _vp, _ll, _vp]),
    "gpsa_quadform_fwd_keep_batched_f64": (_i, [_vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "gpsa_quadform_bwd_alpha_kept_batched_f64": (_i, [_vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _i, _vp]),
    "gpsa_col_axpy_batched_f64": (_i, [_vp, _vp, _vp, _d, _i, _ll, _vp, _i, _vp]),
    "gpsa_gram_batched_workspace": (_ll, [_i, _ll, _i, _i]),
    "gpsa_gram_batched_f64": (_i, [_vp, _vp, _i, _ll, _i, _vp, _i, _vp, _ll, _vp]),
    "gpsa_mvn_kl_grouped_bwd_acc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i,
                                         _vp]),
    "gpsa_step_create": (_vp, [C.POINTER(StepDesc)]),
    "gpsa_step_describe": (_i, [C.POINTER(StepDesc), C.POINTER(_ll)]),
    "gpsa_step_destroy": (None, [_vp]),
    "gpsa_step_saved_bytes": (_ll, [_vp]),
    "gpsa_step_saved_bytes_nokeep": (_ll, [_vp]),
    "gpsa_quadform_keep_f32_workspace": (_ll, [_i, _i]),
    "gpsa_quadform_fwd_keep_f32": (_i, [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_keep_f32_bytes": (_ll, [_i, _ll, _i]),
    "gpsa_quadform_bwd_alpha_kept_f32": (_i, [_vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp]),
    "gpsa_lmc_loglik_workspace": (_ll, [_ll, _i, _i, _i]),
    "gpsa_lmc_loglik_fused_f32": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _i, _vp, _i, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_quadform_elbo_parts": (_i, []),
    "gpsa_quadform_elbo_f32_workspace": (_ll, [_i, _ll, _i]),
    "gpsa_quadform_elbo_f32": (_i, [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _ll, _vp]),
    "gpsa_quadform_elbo_takes_delta": (_i, [_i]),
    "gpsa_quadform_elbo_delta_f32": (_i, [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _ll, _vp]),
    "gpsa_step_scratch_bytes": (_ll, [_vp]),
    "gpsa_step_bwd_acc_bytes": (_ll, [_vp]),
    "gpsa_step_n_kl": (_i, [_vp]),
    "gpsa_step_n_factorised": (_i, [_vp]),
    "gpsa_step_eps_g_numel": (_ll, [_vp]),
    "gpsa_step_batch_layout": (_i, [_vp, C.POINTER(_ll)]),
    "gpsa_step_graph": (_i, [_vp, _i, C.POINTER(_ll)]),
    "gpsa_step_early_backwards": (_ll, [_vp]),
    "gpsa_stream_capture_id": (C.c_ulonglong, [_vp]),
    "gpsa_step_timing": (_i, [_vp, _i]),
    "gpsa_step_timing_read": (_i, [_vp, _vp, _i]),
    "gpsa_step_forward": (_i, [_vp, C.POINTER(StepParams), C.POINTER(StepIO), _vp, _vp, _i, _vp]),
    "gpsa_step_backward": (_i, [_vp, C.POINTER(StepParams), C.POINTER(StepIO), C.POINTER(StepOutGrads), _vp, _vp,
                                C.POINTER(StepParamGrads), _vp]),
    "gpsa_elbo_loss_fwd": (_i, [_i, _pp, _pp, _pp, C.POINTER(_i), C.POINTER(_ll), C.POINTER(_i), _vp, _i, _d, _vp,
                                _vp, _vp, _ll, _vp]),
    "gpsa_elbo_loss_bwd": (_i, [_i, _pp, _pp, _pp, C.POINTER(_i), C.POINTER(_ll), C.POINTER(_i), _vp, _i, _d, _pp, _pp,
                                _vp, _i, _vp, _vp, _ll, _vp]),
    "gpsa_elbo_loss_fused_fwd": (_i, [_i, _pp, _pp, _pp, C.POINTER(_i), C.POINTER(_ll), C.POINTER(_i), _pp, _i, _vp, _i,
                                      _d, _vp, _vp, _vp, _ll, _vp]),
    "gpsa_elbo_loss_fused_bwd": (_i, [_i, _pp, _pp, _pp, C.POINTER(_i), C.POINTER(_ll), C.POINTER(_i), _pp, _i, _vp, _i,
                                      _d, _pp, _pp, _vp, _i, _vp, _vp, _ll, _vp]),
    "gpsa_elbo_fused_post": (_i, [_vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _i, _vp, _vp, _ll, _vp]),
    "gpsa_step_fused": (_i, [_vp, _i]),
    "gpsa_adam_step": (_i, [_i, _pp, _pp, _pp, _pp, C.POINTER(_ll), _d, _d, _d, _d, _vp, _vp]),
})

_lib = None


def source_hash():
    """sha256 over the library's sources (csrc/*.hip, *.hpp, include/gpsa_hip.h: names and contents) - what
    __graft_entry__.build() stamps into the library; None when the sources are not next to the package"""
    import glob
    import hashlib

    csrc = os.path.join(_HERE, "csrc")
    header = os.path.join(os.path.dirname(_HERE), "include", "gpsa_hip.h")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")))
    if not files or not os.path.exists(header):
        return None
    h = hashlib.sha256()
    for f in files + [header]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def library_hash(path=None):
    """the stamp inside a built library, read from its bytes (no dlopen), or None"""
    import re

    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    m = re.search(rb"GPSA_SOURCE_HASH=([0-9a-f]{64}|unstamped)", open(path, "rb").read())
    return m.group(1).decode() if m else None


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
    want = source_hash()
    got = lib.gpsa_source_hash().decode().split("=", 1)[1]
    if want is not None and got != want and os.environ.get("GPSA_ALLOW_STALE_LIB") != "1":
        raise GpsaHipError(
            f"{LIB_PATH} was built from other sources (stamp {got[:12]}, sources {want[:12]}): rebuild it with "
            "`python -c 'import __graft_entry__ as g; g.build()'`"
        )
    _lib = lib
    return lib


GPSA_EINVAL, GPSA_EWORKSPACE, GPSA_EUNSUPPORTED = -1, -2, -3  # include/gpsa_hip.h


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "workspace too small", -3: "unsupported size"}.get(
            rc, f"hipError_t {rc}"
        )
        raise GpsaHipError(f"{what} failed: {kind}")
