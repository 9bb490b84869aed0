"""The hot path's entry points as PyTorch-ROCm custom ops: ``torch.ops.gpsa.*``.

SURVEY.md §8(b) asks for the replacement to be reachable "through PyTorch-ROCm custom ops": every op family of
that row is registered with the dispatcher here (``torch.library.custom_op``, device type "cuda" = HIP on ROCm),
each with

* an implementation that forwards to the SAME C ABI the ctypes binding calls (``ops.HipOps`` ->
  ``libgpsa_hip.so``; the ctypes layer stays the documented non-torch binding),
* a fake-tensor (meta) function, so shapes and dtypes propagate under ``FakeTensorMode`` / ``torch.compile``
  tracing without a device,
* an autograd formula made of the family's own backward op where the family is differentiable.

   family                       op(s)
   (1) covariance matrices                  gpsa::kmat, gpsa::kmat_bwd                 (util.py:8-66)
   (2) factorisation / whitening            gpsa::chol_inv, gpsa::whiten               (vgpsa.py:177-180, 257, 320, 394)
   (3) quadratic form (the star kernel)     gpsa::quadform, gpsa::quadform_bwd_alpha,  (vgpsa.py:192-196)
                                            gpsa::quadform_bwd_omega
   (4) reparameterised draws                gpsa::gauss_sample_F (+ _bwd)              (vgpsa.py:423-426)
   (5) KL between Gaussians                 gpsa::mvn_kl                                (vgpsa.py:498-530)
   (6) Gaussian log-likelihood              gpsa::gauss_loglik_sum (+ _bwd)             (vgpsa.py:532-538)
   the whole step                           gpsa::step_forward, gpsa::step_backward,   (vgpsa.py:212-540,
                                            gpsa::elbo_loss_fwd / _bwd, gpsa::adam_step  grid_example.py:59-78)

The step-engine ops are the ones ``VariationalGPSA.forward`` / ``loss_fn`` / ``FusedAdam`` go through
(step_engine.py, optim.py): they mutate caller-allocated tensors (outputs, arenas, the flat gradient buffer) and
take the non-tensor part of the call (plan handle, pointer structs) as an integer key into ``CALLS``.
"""
import ctypes as C

import torch

from . import _lib
from . import ops as _ops_mod

_raw_stream = torch._C._cuda_getCurrentRawStream
KIND_NAMES = ("rbf", "matern12", "matern32")

# non-tensor arguments of an in-flight step-engine call (ctypes structs cannot cross the dispatcher)
CALLS = {}
_next = [0]


def stash(obj):
    _next[0] += 1
    CALLS[_next[0]] = obj
    return _next[0]


def _o():
    return _ops_mod.get_ops()


# ---------------------------------------------------------------------------------------------------------
# (1) covariance matrices
# ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op("gpsa::kmat", mutates_args=(), device_types="cuda")
def kmat(Z: torch.Tensor, X: torch.Tensor, ls_u: torch.Tensor, var_u: torch.Tensor, kind: str,
         jitter: float = 0.0) -> torch.Tensor:
    """K[m, c] = k(Z[m], X[c]) + jitter [Z is X]; kind in rbf | matern12 | matern32 (the plugin API's built-ins)"""
    return _o().kmat(kind, Z, X, ls_u, var_u, jitter)


@kmat.register_fake
def _(Z, X, ls_u, var_u, kind, jitter=0.0):
    return Z.new_empty(Z.shape[0], X.shape[0])


@torch.library.custom_op("gpsa::kmat_bwd", mutates_args=(), device_types="cuda")
def kmat_bwd(Z: torch.Tensor, X: torch.Tensor, ls_u: torch.Tensor, var_u: torch.Tensor, Kbar: torch.Tensor,
             kind: str) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    dZ, dX, dpar = _o().kmat_bwd(kind, Z, X, ls_u, var_u, Kbar, need_dX=True)
    return dZ, dX, dpar


@kmat_bwd.register_fake
def _(Z, X, ls_u, var_u, Kbar, kind):
    return Z.new_empty(Z.shape), X.new_empty(X.shape), Z.new_empty(2)


def _kmat_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:4])
    ctx.kind = inputs[4]


def _kmat_backward(ctx, Kbar):
    Z, X, ls_u, var_u = ctx.saved_tensors
    dZ, dX, dpar = torch.ops.gpsa.kmat_bwd(Z, X, ls_u, var_u, Kbar.contiguous(), ctx.kind)
    return dZ, dX, dpar[0].reshape(ls_u.shape), dpar[1].reshape(var_u.shape), None, None


kmat.register_autograd(_kmat_backward, setup_context=_kmat_setup)


# ---------------------------------------------------------------------------------------------------------
# (2) factorisation and whitening
# ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op("gpsa::chol_inv", mutates_args=(), device_types="cuda")
def chol_inv(A: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """A [B, M, M] fp64 -> (L^-1, logdet A, info): the fused batched Cholesky + inverse of the factor"""
    return _o().chol_inv(A)


@chol_inv.register_fake
def _(A):
    return A.new_empty(A.shape), A.new_empty(A.shape[0]), A.new_empty(A.shape[0], dtype=torch.int32)


@torch.library.custom_op("gpsa::whiten", mutates_args=(), device_types="cuda")
def whiten(Kinv: torch.Tensor, Kuf: torch.Tensor, fp32_out: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    """alpha = K^-1 K_uf (fp64 matrix cores), q[c] = K_uf[:, c] . alpha[:, c]; M <= 256"""
    r = _o().whiten(Kinv, Kuf, torch.float32 if fp32_out else torch.float64)
    if r is None:
        raise _lib.GpsaHipError("gpsa::whiten: M beyond the projection kernel (chain gpsa_panel_mm / gpsa_gemm)")
    return r


@whiten.register_fake
def _(Kinv, Kuf, fp32_out=True):
    return (Kuf.new_empty(Kuf.shape, dtype=torch.float32 if fp32_out else torch.float64),
            Kuf.new_empty(Kuf.shape[1], dtype=torch.float64))


# ---------------------------------------------------------------------------------------------------------
# (3) the quadratic form  v[l, c] = alpha_c^T Omega_l alpha_c
# ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op("gpsa::quadform", mutates_args=(), device_types="cuda")
def quadform(alpha: torch.Tensor, Omega: torch.Tensor) -> torch.Tensor:
    return _o().quadform_fwd(alpha, Omega)


@quadform.register_fake
def _(alpha, Omega):
    return alpha.new_empty(Omega.shape[0], alpha.shape[1])


@torch.library.custom_op("gpsa::quadform_bwd_alpha", mutates_args=(), device_types="cuda")
def quadform_bwd_alpha(alpha: torch.Tensor, Omega: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    return _o().quadform_bwd_alpha(alpha, Omega, g)


@quadform_bwd_alpha.register_fake
def _(alpha, Omega, g):
    return alpha.new_empty(alpha.shape)


@torch.library.custom_op("gpsa::quadform_bwd_omega", mutates_args=(), device_types="cuda")
def quadform_bwd_omega(alpha: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    return _o().quadform_bwd_omega(alpha, g)


@quadform_bwd_omega.register_fake
def _(alpha, g):
    return alpha.new_empty(g.shape[0], alpha.shape[0], alpha.shape[0])


def _qf_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _qf_backward(ctx, g):
    alpha, Omega = ctx.saved_tensors
    g = g.contiguous()
    da = torch.ops.gpsa.quadform_bwd_alpha(alpha, Omega, g) if ctx.needs_input_grad[0] else None
    dO = torch.ops.gpsa.quadform_bwd_omega(alpha, g).to(Omega.dtype) if ctx.needs_input_grad[1] else None
    return da, dO


quadform.register_autograd(_qf_backward, setup_context=_qf_setup)


# ---------------------------------------------------------------------------------------------------------
# (4) reparameterised draw of the data GP: F = mean + sqrt(sigma^2 - q + v + 2e-5) eps
# ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op("gpsa::gauss_sample_F", mutates_args=(), device_types="cuda")
def gauss_sample_F(meanT: torch.Tensor, v: torch.Tensor, q: torch.Tensor, var_u: torch.Tensor,
                   eps: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """meanT, v [L, C]; q [C]; eps [C, L] -> (F [C, L], Sigma [L, C])"""
    return _o().data_sample_fwd(meanT, v, q, var_u, eps)


@gauss_sample_F.register_fake
def _(meanT, v, q, var_u, eps):
    return meanT.new_empty(meanT.shape[1], meanT.shape[0]), meanT.new_empty(meanT.shape)


@torch.library.custom_op("gpsa::gauss_sample_F_bwd", mutates_args=(), device_types="cuda")
def gauss_sample_F_bwd(dF: torch.Tensor, eps: torch.Tensor, Sigma: torch.Tensor,
                       var_u: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (g_ext [L+1, C]: d/dv rows and qbar = d/dq, dmeanT [L, C], dvar_u [1])"""
    return _o().data_sample_bwd(dF, eps, Sigma, var_u)


@gauss_sample_F_bwd.register_fake
def _(dF, eps, Sigma, var_u):
    L, Cn = Sigma.shape
    return Sigma.new_empty(L + 1, Cn), Sigma.new_empty(L, Cn), Sigma.new_empty(1)


def _gs_setup(ctx, inputs, output):
    meanT, v, q, var_u, eps = inputs
    ctx.save_for_backward(eps, output[1], var_u)
    ctx.qdtype = q.dtype


def _gs_backward(ctx, dF, dSigma):
    eps, Sigma, var_u = ctx.saved_tensors
    g_ext, dmeanT, dvar = torch.ops.gpsa.gauss_sample_F_bwd(dF.contiguous(), eps, Sigma, var_u)
    L = Sigma.shape[0]
    return dmeanT, g_ext[:L], g_ext[L].to(ctx.qdtype), dvar.reshape(var_u.shape).to(var_u.dtype), None


gauss_sample_F.register_autograd(_gs_backward, setup_context=_gs_setup)


# ---------------------------------------------------------------------------------------------------------
# (5) KL(N(d, Omega) || N(0, K)) for a batch sharing one prior
# ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op("gpsa::mvn_kl", mutates_args=(), device_types="cuda")
def mvn_kl(Kinv: torch.Tensor, logdetK: torch.Tensor, Omega: torch.Tensor, logdetO: torch.Tensor,
           Dm: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (kl [B], K^-1 d [B, M]); Kinv [M, M], Omega [B, M, M], Dm [B, M] mean differences (fp64)"""
    return _o().mvn_kl_fwd(Kinv, logdetK, Omega, logdetO, Dm)


@mvn_kl.register_fake
def _(Kinv, logdetK, Omega, logdetO, Dm):
    return Dm.new_empty(Dm.shape[0]), Dm.new_empty(Dm.shape)


# ---------------------------------------------------------------------------------------------------------
# (6) Gaussian log-likelihood  sum log N(Y; F, scale = exp(noise_u) + 1e-5) / S
# ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op("gpsa::gauss_loglik_sum", mutates_args=(), device_types="cuda")
def gauss_loglik_sum(F: torch.Tensor, Y: torch.Tensor, noise_u: torch.Tensor) -> torch.Tensor:
    return _o().loglik_fwd(F, Y, noise_u)


@gauss_loglik_sum.register_fake
def _(F, Y, noise_u):
    return F.new_empty(1, dtype=torch.float64)


@torch.library.custom_op("gpsa::gauss_loglik_sum_bwd", mutates_args=(), device_types="cuda")
def gauss_loglik_sum_bwd(F: torch.Tensor, Y: torch.Tensor, noise_u: torch.Tensor,
                         gout: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    return _o().loglik_bwd(F, Y, noise_u, gout)


@gauss_loglik_sum_bwd.register_fake
def _(F, Y, noise_u, gout):
    return F.new_empty(F.shape), F.new_empty(1)


def _ll_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _ll_backward(ctx, gout):
    F, Y, noise_u = ctx.saved_tensors
    dF, dn = torch.ops.gpsa.gauss_loglik_sum_bwd(F, Y, noise_u, gout.contiguous())
    return dF, None, dn.reshape(noise_u.shape).to(noise_u.dtype)


gauss_loglik_sum.register_autograd(_ll_backward, setup_context=_ll_setup)


# ---------------------------------------------------------------------------------------------------------
# the step engine (what the model classes call)
#
# These five run once or twice per training step, also on problems whose whole step is under a millisecond, so they
# are registered through the low-level ``torch.library.Library`` interface (schema string + CUDA kernel + fake
# function): ~10 us of dispatch per call instead of the ~35 us of a ``custom_op`` object (the reference example's
# own 2 x 100-spot problem ran 920 -> 720 steps/s with the latter).  Mutated arguments are declared in the schema.
# ---------------------------------------------------------------------------------------------------------
_ENGINE = torch.library.Library("gpsa", "FRAGMENT")


def _engine_op(schema, fn):
    name = schema.split("(", 1)[0]
    _ENGINE.define(schema)
    _ENGINE.impl(name, fn, "CUDA")
    torch.library.register_fake(f"gpsa::{name}", lambda *a, **k: None, lib=_ENGINE)
    return fn


def _step_forward(params, ins, outs, saved, scratch, call, stages):
    """gpsa_step_forward: warp GPs (stage 1) and data GPs (stage 2) of VariationalGPSA.forward into ``outs``"""
    c = CALLS[call]
    _lib.check(c["lib"].gpsa_step_forward(c["handle"], C.byref(c["prm"]), C.byref(c["io"]), saved.data_ptr(),
                                          scratch.data_ptr(), int(stages), _raw_stream(saved.device.index)),
               "gpsa_step_forward")


_engine_op("step_forward(Tensor[] params, Tensor[] ins, Tensor(a!)[] outs, Tensor(b!) saved, Tensor(c!) scratch, "
           "int call, int stages) -> ()", _step_forward)


def _step_backward(params, grads_out, saved, flat, scratch, call):
    """gpsa_step_backward: every parameter gradient of the step into the flat buffer ``flat``"""
    c = CALLS[call]
    _lib.check(c["lib"].gpsa_step_backward(c["handle"], C.byref(c["prm"]), C.byref(c["io"]), C.byref(c["og"]),
                                           saved.data_ptr(), scratch.data_ptr(), C.byref(c["grads"]),
                                           _raw_stream(saved.device.index)), "gpsa_step_backward")


_engine_op("step_backward(Tensor[] params, Tensor[] grads_out, Tensor saved, Tensor(a!) flat, Tensor(b!) scratch, "
           "int call) -> ()", _step_backward)


def _ll_arrays(Fs, Ys, noise, noise_idx):
    n = len(Fs)
    arr = lambda vals: (C.c_void_p * n)(*vals)
    return (n, arr([f.data_ptr() for f in Fs]), arr([y.data_ptr() for y in Ys]),
            arr([noise.data_ptr() + 4 * j for j in noise_idx]), (C.c_int * n)(*[int(f.shape[0]) for f in Fs]),
            (C.c_longlong * n)(*[int(f.shape[1]) for f in Fs]), (C.c_int * n)(*[int(f.shape[2]) for f in Fs]))


def _elbo_loss_fwd(Fs, Ys, noise, noise_idx, kl, kl_scale, loss, ll, ws):
    """loss = -(sum_i LL_i) + kl_scale * sum(kl): log-likelihood partials, finish and the ELBO glue
    (gpsa_elbo_loss_fwd; contiguous fp32 F / Y / noise, fp64 kl)"""
    n, Fp, Yp, Np, Sa, Na, Pa = _ll_arrays(Fs, Ys, noise, noise_idx)
    _lib.check(_lib.load().gpsa_elbo_loss_fwd(n, Fp, Yp, Np, Sa, Na, Pa, 0 if kl is None else kl.data_ptr(),
                                              0 if kl is None else kl.numel(), float(kl_scale), loss.data_ptr(),
                                              ll.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _raw_stream(loss.device.index)), "gpsa_elbo_loss_fwd")


_engine_op("elbo_loss_fwd(Tensor[] Fs, Tensor[] Ys, Tensor noise, int[] noise_idx, Tensor? kl, float kl_scale, "
           "Tensor(a!) loss, Tensor(b!) ll, Tensor(c!) ws) -> ()", _elbo_loss_fwd)


def _elbo_loss_bwd(Fs, Ys, noise, noise_idx, gloss, n_kl, kl_scale, dFs, dnoise, dkl, ws):
    n, Fp, Yp, Np, Sa, Na, Pa = _ll_arrays(Fs, Ys, noise, noise_idx)
    dFp = (C.c_void_p * n)(*[t.data_ptr() for t in dFs])
    dNp = (C.c_void_p * n)(*[dnoise.data_ptr() + 4 * j for j in noise_idx])
    _lib.check(_lib.load().gpsa_elbo_loss_bwd(n, Fp, Yp, Np, Sa, Na, Pa, gloss.data_ptr(), int(n_kl), float(kl_scale),
                                              dFp, dNp, dnoise.data_ptr(), dnoise.numel(),
                                              0 if dkl is None else dkl.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _raw_stream(gloss.device.index)), "gpsa_elbo_loss_bwd")


_engine_op("elbo_loss_bwd(Tensor[] Fs, Tensor[] Ys, Tensor noise, int[] noise_idx, Tensor gloss, int n_kl, "
           "float kl_scale, Tensor(a!)[] dFs, Tensor(b!) dnoise, Tensor(c!)? dkl, Tensor(d!) ws) -> ()", _elbo_loss_bwd)


def _ll_arrays_fused(Fs, Ys, noise, noise_idx, shapes, fused):
    """as _ll_arrays with explicit [S, N, P] per term; a fused term's "F" is its partial-sum vector (gpsa_step_io.ll_part)"""
    n = len(Fs)
    arr = lambda vals: (C.c_void_p * n)(*vals)
    nparts = max([int(f.numel()) for f, z in zip(Fs, fused) if z] or [0])
    return (n, arr([0 if z else f.data_ptr() for f, z in zip(Fs, fused)]), arr([y.data_ptr() for y in Ys]),
            arr([noise.data_ptr() + 4 * j for j in noise_idx]), (C.c_int * n)(*[int(shapes[3 * i]) for i in range(n)]),
            (C.c_longlong * n)(*[int(shapes[3 * i + 1]) for i in range(n)]),
            (C.c_int * n)(*[int(shapes[3 * i + 2]) for i in range(n)]),
            arr([f.data_ptr() if z else 0 for f, z in zip(Fs, fused)]), nparts)


def _elbo_loss_fused_fwd(Fs, Ys, noise, noise_idx, shapes, fused, kl, kl_scale, loss, ll, ws):
    """gpsa_elbo_loss_fused_fwd: the loss with some likelihood terms already reduced to partial sums by the step"""
    n, Fp, Yp, Np, Sa, Na, Pa, Zp, nparts = _ll_arrays_fused(Fs, Ys, noise, noise_idx, shapes, fused)
    _lib.check(_lib.load().gpsa_elbo_loss_fused_fwd(n, Fp, Yp, Np, Sa, Na, Pa, Zp, nparts,
                                                    0 if kl is None else kl.data_ptr(), 0 if kl is None else kl.numel(),
                                                    float(kl_scale), loss.data_ptr(), ll.data_ptr(), ws.data_ptr(),
                                                    ws.numel(), _raw_stream(loss.device.index)),
               "gpsa_elbo_loss_fused_fwd")


_engine_op("elbo_loss_fused_fwd(Tensor[] Fs, Tensor[] Ys, Tensor noise, int[] noise_idx, int[] shapes, int[] fused, "
           "Tensor? kl, float kl_scale, Tensor(a!) loss, Tensor(b!) ll, Tensor(c!) ws) -> ()", _elbo_loss_fused_fwd)


def _elbo_loss_fused_bwd(Fs, Ys, noise, noise_idx, shapes, fused, gloss, n_kl, kl_scale, dFs, dnoise, dkl, ws):
    n, Fp, Yp, Np, Sa, Na, Pa, Zp, nparts = _ll_arrays_fused(Fs, Ys, noise, noise_idx, shapes, fused)
    dFp = (C.c_void_p * n)(*[0 if z else t.data_ptr() for t, z in zip(dFs, fused)])
    dNp = (C.c_void_p * n)(*[dnoise.data_ptr() + 4 * j for j in noise_idx])
    _lib.check(_lib.load().gpsa_elbo_loss_fused_bwd(n, Fp, Yp, Np, Sa, Na, Pa, Zp, nparts, gloss.data_ptr(), int(n_kl),
                                                    float(kl_scale), dFp, dNp, dnoise.data_ptr(), dnoise.numel(),
                                                    0 if dkl is None else dkl.data_ptr(), ws.data_ptr(), ws.numel(),
                                                    _raw_stream(gloss.device.index)), "gpsa_elbo_loss_fused_bwd")


_engine_op("elbo_loss_fused_bwd(Tensor[] Fs, Tensor[] Ys, Tensor noise, int[] noise_idx, int[] shapes, int[] fused, "
           "Tensor gloss, int n_kl, float kl_scale, Tensor(a!)[] dFs, Tensor(b!) dnoise, Tensor(c!)? dkl, "
           "Tensor(d!) ws) -> ()", _elbo_loss_fused_bwd)


def _lmc_loglik_fused(F, W, Y, noise, noise_idx, zpart, dF, dW, ws):
    """gpsa_lmc_loglik_fused_f32: an LMC modality's likelihood partial sums, dLoss/dF_latent and dLoss/dW in one pass
    over (F_latent [S,N,L], W [L,P], Y [N,P]) - F_obs = F_latent W is never formed"""
    S, N, L = (int(d) for d in F.shape)
    _lib.check(_lib.load().gpsa_lmc_loglik_fused_f32(F.data_ptr(), W.data_ptr(), Y.data_ptr(),
                                                     noise.data_ptr() + 4 * int(noise_idx), S, N, L, int(W.shape[1]),
                                                     zpart.data_ptr(), zpart.numel(), dF.data_ptr(), dW.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), _raw_stream(F.device.index)),
               "gpsa_lmc_loglik_fused_f32")


_engine_op("lmc_loglik_fused(Tensor F, Tensor W, Tensor Y, Tensor noise, int noise_idx, Tensor(a!) zpart, Tensor(b!) dF, "
           "Tensor(c!) dW, Tensor(d!) ws) -> ()", _lmc_loglik_fused)


def _adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps):
    """torch.optim.Adam's update over all tensors in one launch, step counter on the device (gpsa_adam_step)"""
    n = len(params)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    numel = (C.c_longlong * n)(*[p.numel() for p in params])
    _lib.check(_lib.load().gpsa_adam_step(n, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel, float(lr),
                                          float(beta1), float(beta2), float(eps), step.data_ptr(),
                                          _raw_stream(step.device.index)), "gpsa_adam_step")


_engine_op("adam_step(Tensor(a!)[] params, Tensor[] grads, Tensor(b!)[] exp_avg, Tensor(c!)[] exp_avg_sq, "
           "Tensor(d!) step, float lr, float beta1, float beta2, float eps) -> ()", _adam_step)
