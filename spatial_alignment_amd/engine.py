"""Autograd nodes of the GPSA hot path.  Every forward AND backward below is a sequence of
hand-written HIP kernels (through ``ops``); autograd is used only to chain the nodes.

Math (one sparse-GP layer; reference gpsa/models/vgpsa.py:174-204, columns c = spots or (sample, spot)):

    K = k(Z,Z) + 1e-5 I = L L^T          k_c = k(Z, x_c)
    beta_c = L^-1 k_c ,  alpha_c = L^-T beta_c = K^-1 k_c ,  q_c = |beta_c|^2 = k_c^T K^-1 k_c
    mean[l,c] = alpha_c^T dc[:,l]                       (dc = delta - mu_z)
    v[l,c]    = alpha_c^T Omega_l alpha_c               (Omega_l = A_l A_l^T + 1e-5 I;  the reference
                                                          computes |Omega_tril_l^T alpha_c|^2 — identical)
    var[l,c]  = sigma^2 - q_c + v[l,c] + 2e-5

Backward, with abar = d/d alpha, qbar = d/dq  (derivation in DESIGN.md §4):
    abar_c  = dc dmean[:,c] + 2 sum_l g[l,c] Omega_l alpha_c
    gamma_c = K^-1 abar_c ,   W_c = gamma_c + qbar_c alpha_c
    dK_uf[:,c] = W_c + qbar_c alpha_c ,   dK_uu = - W alpha^T
    d dc = alpha dmean^T ,    dOmega_l = sum_c g[l,c] alpha_c alpha_c^T
The M x M factorisations run in fp64 (cond(K_uu) ~ 2e7 for the warp GP); the warp
layer runs entirely in fp64, the data layer's N-scaled work in fp32 on the MFMA units.
"""
import torch

from . import ops as _ops_mod

JITTER = 1e-5  # gpsa/models/gpsa.py:153


def ops():
    return _ops_mod.get_ops()


class Factor:
    """fp64 factorisation of one symmetric positive-definite M x M matrix (a prior covariance K_uu,
    shared by the layer and its KL term, or a variational covariance Omega_l)."""

    __slots__ = ("Linv", "Kinv", "logdet", "info")

    def __init__(self, Kuu64=None, parts=None):
        if parts is None:
            parts = factor_batch([Kuu64.detach().unsqueeze(0)])[0]
        Linv, Kinv, logdet, info = parts
        self.Linv = Linv[0] if Linv.dim() == 3 and Linv.shape[0] == 1 else Linv
        self.Kinv = Kinv[0] if Kinv.dim() == 3 and Kinv.shape[0] == 1 else Kinv
        self.logdet = logdet
        self.info = info


def factor_stack(stack):
    """(Linv, inverse, logdet, info) of a [B,M,M] batch of SPD matrices: ONE fused Cholesky + triangular
    inverse launch and ONE batched L^-T L^-1 product for the whole batch."""
    o = ops()
    Linv, logdet, info = o.chol_inv(stack)
    return Linv, o.gemm(Linv, Linv, transA=True), logdet, info


def factor_batch(mats, stack=None):
    """Factorise many SPD matrices with as few launches as possible.

    ``mats``: list of fp64 tensors [b_i, M_i, M_i].  Matrices of equal size share one factor_stack call
    (one workgroup per matrix), so the launch-latency-bound factorisations of a whole step cost one
    kernel's latency.  ``stack``: the matrices already laid out back to back in this [sum b_i, M, M]
    buffer (then nothing is copied).
    Returns, per input, (Linv [b,M,M], inverse [b,M,M], logdet [b], info [b]); with ``stack`` also the
    four whole-batch tensors as a second value.
    """
    out = [None] * len(mats)
    by_size = {}
    for i, m in enumerate(mats):
        by_size.setdefault(m.shape[-1], []).append(i)
    whole = None
    for M, idxs in by_size.items():
        if stack is not None and len(by_size) == 1:
            batch = stack
        else:
            batch = torch.cat([mats[i].detach().reshape(-1, M, M) for i in idxs], 0)
        Linv, inv, logdet, info = whole = factor_stack(batch)
        off = 0
        for i in idxs:
            b = mats[i].reshape(-1, M, M).shape[0]
            out[i] = (Linv[off : off + b], inv[off : off + b], logdet[off : off + b], info[off : off + b])
            off += b
    if stack is not None and len(by_size) == 1:
        return out, whole
    return out


def _cov_inputs(Z, X, ls_u, var_u, dtype):
    """Detached covariance inputs in one storage dtype the kernels accept next to compute dtype
    ``dtype``: (fp32 | fp64 storage, fp64 compute) or (fp32, fp32)."""
    sd = Z.dtype if Z.dtype in (torch.float32, torch.float64) else torch.float32
    if dtype == torch.float32:
        sd = torch.float32
    fix = lambda t: t.detach() if t.dtype == sd else t.detach().to(sd)
    return fix(Z), fix(X), fix(ls_u).reshape(1), fix(var_u).reshape(1)


class KmatFn(torch.autograd.Function):
    """K = k(Z, X) (+ jitter on the diagonal), computed in ``dtype``; fused HIP forward and backward.
    Replaces the built-in plugins gpsa/util/util.py:8-66."""

    @staticmethod
    def forward(ctx, kind, Z, X, ls_u, var_u, jitter, dtype, same, bwd_dtype=None, out=None):
        o = ops()
        # the kernels read the (fp32) parameters as stored and compute in ``dtype``: no cast launches
        Zc, Xc, ls, var = _cov_inputs(Z, X, ls_u, var_u, dtype)
        K = o.kmat(kind, Zc, Xc, ls, var, jitter, dtype=dtype, out=out)
        ctx.kind, ctx.same = kind, bool(same)
        ctx.bwd_dtype = bwd_dtype or dtype  # gradient-only precision for the backward
        if ctx.bwd_dtype == torch.float32 and Zc.dtype != torch.float32:
            Zc, Xc, ls, var = (t.float() for t in (Zc, Xc, ls, var))
        ctx.save_for_backward(Zc, Xc, ls, var)
        ctx.meta = (Z.dtype, X.dtype, ls_u.dtype, var_u.dtype, ls_u.shape, var_u.shape)
        return K

    @staticmethod
    def backward(ctx, Kbar):
        o = ops()
        Zc, Xc, ls, var = ctx.saved_tensors
        zdt, xdt, ldt, vdt, lshape, vshape = ctx.meta
        dZ, dX, dpar = o.kmat_bwd(ctx.kind, Zc, Xc, ls, var, Kbar.to(ctx.bwd_dtype),
                                  need_dX=ctx.needs_input_grad[2], same=ctx.same)
        gZ = dZ.to(zdt) if ctx.needs_input_grad[1] else None
        gX = dX.to(xdt) if (dX is not None and ctx.needs_input_grad[2]) else None
        gl = dpar[0].to(ldt).reshape(lshape) if ctx.needs_input_grad[3] else None
        gv = dpar[1].to(vdt).reshape(vshape) if ctx.needs_input_grad[4] else None
        return None, gZ, gX, gl, gv, None, None, None, None, None


class OmegaFn(torch.autograd.Function):
    """Omega = A A^T + 1e-5 I in fp64 (vgpsa.py:206-210).  A [B,M,M] fp32 parameter rows.
    ``symmetric_grad``: the caller promises that every gradient that reaches Omega is symmetric (true
    for all consumers in this package: mirrored Gram sums and differences of inverses of symmetric
    matrices, symmetric to rounding), so the adjoint (G + G^T) A is evaluated as 2 G A."""

    @staticmethod
    def forward(ctx, A, out=None, symmetric_grad=False):
        Ad = A.detach()
        Om = ops().omega_fwd(Ad, JITTER, out=out)  # ``out``: a slice of the step's factorisation batch
        ctx.save_for_backward(Ad)
        ctx.adt, ctx.sym = A.dtype, bool(symmetric_grad)
        return Om

    @staticmethod
    def backward(ctx, dOm):
        (Ad,) = ctx.saved_tensors
        return ops().omega_bwd(dOm, Ad, symmetric=ctx.sym).to(ctx.adt), None, None


def _project(o, fac, Kuf, T):
    """alpha = K_uu^-1 K_uf (stored as T) and q = k^T K_uu^-1 k in K_uf's precision.  fp64: ONE pass of
    the fp64-MFMA kernel with the explicit inverse; otherwise the two triangular passes (their fp32
    error grows with cond(L) = sqrt(cond(K)), not cond(K)).  The fp64 factor is read as stored."""
    fused = o.whiten(fac.Kinv, Kuf, T) if Kuf.dtype == torch.float64 else None
    if fused is not None:
        return fused
    beta, q = o.panel_mm(fac.Linv, Kuf, want_colsq=True)
    alpha_w, _ = o.panel_mm(fac.Linv, beta, transP=True)
    return (alpha_w if alpha_w.dtype == T else alpha_w.to(T)), q


def _solve_K(o, Linv, Kinv, abar):
    """gamma = K_uu^-1 abar in abar's precision: ONE pass of the fp64-MFMA kernel with the explicit fp64
    inverse (the panel is widened on the fly, fp32 or fp64), the two triangular passes beyond its size"""
    fused = o.whiten(Kinv, abar, abar.dtype, want_q=False)
    if fused is not None:
        return fused[0]
    t, _ = o.panel_mm(Linv, abar)
    return o.panel_mm(Linv, t, transP=True)[0]


def _layer_backward(o, alpha, dcT, Om, Linv, Kinv, dmeanT, g, qbar, need_dOm, g_ext=None, W=None):
    """shared backward of the sparse-GP layer: returns dKuu, dKuf, ddc, dOm (alpha's precision, dKuu
    possibly fp64).  ``g_ext`` [L+1, C] = g rows followed by the row qbar = -sum_l g in ONE buffer (the
    fused data layer; its presence vouches for that identity, which the dK_uu shortcut below uses)."""
    M, Cn = alpha.shape
    L = Om.shape[0]
    T = alpha.dtype
    zeros = lambda *sh: torch.zeros(*sh, dtype=T, device=alpha.device)
    dmeanT = zeros(L, Cn) if dmeanT is None else dmeanT.to(T).contiguous()
    if g_ext is not None:
        g, qbar = g_ext[:L], g_ext[L]
    else:
        g = zeros(L, Cn) if g is None else g.to(T).contiguous()
        qbar = zeros(Cn) if qbar is None else qbar.to(T).contiguous()
    if W is None:
        abar = o.quadform_bwd_alpha(alpha, Om, g)
        o.gemm(dcT, dmeanT, beta=1.0, out=abar)
    else:  # the products Omega_l alpha kept by the forward (few-output layers): one streaming pass, the mean
        abar = o.quadform_bwd_alpha_kept(W, g, dcT, dmeanT)  # term's share dcT dmeanT included
    ddc = o.gemm(alpha, dmeanT, transB=True, splitk=o.pick_splitk(Cn, M, L))
    gamma = _solve_K(o, Linv, Kinv, abar)
    if g_ext is not None and need_dOm and Cn >= 4 * L * M:
        # dK_uu = -(gamma + qbar a) a^T without a second C-long product:
        #   gamma a^T = K^-1 (abar a^T),   abar a^T = dc ddc^T + 2 sum_l Omega_l dOmega_l,   dOmega_l = sum_c g a a^T
        #   (qbar a) a^T = sum_c qbar_c a_c a_c^T = - sum_l dOmega_l          (qbar = - sum_l g_l, exactly)
        # i.e. one product with K = L M (fp64, on what the Gram kernel already produced) instead of K = C
        f64 = torch.float64
        dOm64 = o.quadform_bwd_omega(alpha, g, out_dtype=f64)  # [L, M, M]; the caller wants fp64 anyway
        P = o.gemm(Om.reshape(L * M, M), dOm64.reshape(L * M, M), transA=True, alpha=2.0,
                   splitk=o.pick_splitk(L * M, M, M))
        o.gemm(dcT.to(f64), ddc.to(f64), transB=True, beta=1.0, out=P)
        dKuu = o.gemm(Kinv, P, alpha=-1.0, beta=1.0, out=dOm64.sum(0))  # -(K^-1 P) + sum_l dOmega_l
        dKuf = o.col_axpy(gamma, alpha, qbar, 2.0, out=gamma)
        return dKuu, dKuf, ddc, dOm64
    if need_dOm and T == torch.float64 and M <= 256 and Cn % 4 == 0 and Cn >= 4096:
        # fp64 layer (the warp GP): dOmega is a plain gradient (it takes no part in the sigma^2
        # cancellation that forces fp64 on dK_uu / dK_uf), so it runs on the fp32 MFMA Gram kernel on the
        # rounded alpha: 3 launches instead of L x (scale + C-long fp64 product + split-K reduce).  Only
        # where it pays: the LDS-DMA staging needs 16-byte aligned rows, and a small C is latency either way
        dOm = o.quadform_bwd_omega(alpha.float(), g.float(), out_dtype=T)
    else:
        dOm = o.quadform_bwd_omega(alpha, g) if need_dOm else None
    W = o.col_axpy(gamma, alpha, qbar, 1.0, out=gamma)
    if T == torch.float32:
        # fp32 layer, few columns (the shortcut above does not pay): the C-long product still has to be
        # ADDED in fp64 - dK_uu meets dK_uf in the covariance parameters' gradients, where the two cancel to
        # ~1e-4 of their size (measured at M = 1000: data_kernel_variance 4.9e-2 -> 1.1e-5 off the oracle)
        dKuu = o.gemm(W.double(), alpha.double(), transB=True, alpha=-1.0, splitk=o.pick_splitk(Cn, M, M))
    else:
        dKuu = o.gemm(W, alpha, transB=True, alpha=-1.0, splitk=o.pick_splitk(Cn, M, M))
    dKuf = o.col_axpy(W, alpha, qbar, 1.0, out=W)
    return dKuu, dKuf, ddc, dOm


class SGPCoreFn(torch.autograd.Function):
    """(K_uf, dc, Omega; factor of K_uu) -> meanT [L,C], v [L,C], q [C]   (vgpsa.py:174-204).

    The whitening (beta = L^-1 K_uf, q = |beta|^2, alpha = L^-T beta) runs in K_uf's precision (fp64:
    sigma^2 - q cancels to ~1e-3 sigma^2 when the inducing points are dense, and cond(L) ~ 1e3);
    mean and the dominant quadratic form run in ``main_dtype`` (fp32 MFMA for the data layer) on the
    rounded alpha.  q keeps K_uf's precision so that the sampler can form sigma^2 - q before rounding.
    """

    @staticmethod
    def forward(ctx, Kuu, Kuf, dc, Omega, fac, main_dtype):
        o = ops()
        T = main_dtype
        alpha, q = _project(o, fac, Kuf.detach(), T)
        dcT = dc.detach().to(T).contiguous()
        Om = Omega.detach()  # read as stored (fp64); rounded to T while packed for the matrix cores
        meanT = o.gemm(dcT, alpha, transA=True)
        v = o.quadform_fwd(alpha, Om)
        ctx.save_for_backward(alpha, dcT, Om, fac.Linv, fac.Kinv)
        ctx.meta = (Kuu.dtype, Kuf.dtype, dc.dtype, Omega.dtype)
        return meanT, v, q

    @staticmethod
    def backward(ctx, dmeanT, g, qbar):
        o = ops()
        alpha, dcT, Om, Linv, Kinv = ctx.saved_tensors
        kdt, fdt, ddt, odt = ctx.meta
        dKuu, dKuf, ddc, dOm = _layer_backward(o, alpha, dcT, Om, Linv, Kinv, dmeanT, g, qbar,
                                               ctx.needs_input_grad[3])
        return (
            dKuu.to(kdt),
            dKuf.to(fdt),
            ddc.to(ddt),
            dOm.to(odt) if dOm is not None else None,
            None,
            None,
        )


class SGPDataLayerFn(torch.autograd.Function):
    """The whole data GP of one modality as ONE node: covariance, projection, mean / variance forms and
    the reparameterised draw F = mean + sqrt(var) eps (SGPCoreFn + DataSampleFn with the covariance fused in
    and without the autograd edges between them: no fp64 <-> fp32 round trip of q and its gradient, g and qbar in one buffer).
    fp64 covariance + projection, fp32 MFMA contractions, fp32 backward.  Returns F [C, L] fp32."""

    @staticmethod
    def forward(ctx, kind, Z, X, ls_u, var_u, Kuu, dc, Omega, fac, eps, X64=None):
        o = ops()
        f64, T = torch.float64, torch.float32
        Zs, Xs, lss, vars_ = _cov_inputs(Z, X, ls_u, var_u, f64)
        # X64: the warp GP's draws before their rounding to the fp32 API tensor X (same values otherwise):
        # the covariance is built from those, as in the reference's fp64 run; the backward keeps fp32 X
        Kuf = o.kmat(kind, Zs, Xs if X64 is None else X64.detach(), lss, vars_, 0.0, dtype=f64)
        alpha, q = _project(o, fac, Kuf, T)
        del Kuf
        dcT = dc.detach().to(T).contiguous()
        Om = Omega.detach()
        meanT = o.gemm(dcT, alpha, transA=True)
        v = o.quadform_fwd(alpha, Om)
        var32 = vars_ if vars_.dtype == T else vars_.float()
        F, Sigma = o.data_sample_fwd(meanT, v, q, var32, eps)
        if Zs.dtype != T:
            Zs, Xs, lss = Zs.float(), Xs.float(), lss.float()
        ctx.save_for_backward(alpha, dcT, Om, fac.Linv, fac.Kinv, Zs, Xs, lss, var32, eps, Sigma)
        ctx.kind = kind
        ctx.meta = (Z.dtype, X.dtype, ls_u.dtype, ls_u.shape, var_u.dtype, var_u.shape, Kuu.dtype,
                    dc.dtype, Omega.dtype)
        ctx.via_x64 = X64 is not None and X64.requires_grad
        return F

    @staticmethod
    def backward(ctx, dF):
        o = ops()
        alpha, dcT, Om, Linv, Kinv, Zb, Xb, lsb, var32, eps, Sigma = ctx.saved_tensors
        zdt, xdt, ldt, lshape, vdt, vshape, kdt, ddt, odt = ctx.meta
        g_ext, dmeanT, dvar_s = o.data_sample_bwd(dF.contiguous(), eps, Sigma, var32)
        dKuu, dKuf, ddc, dOm = _layer_backward(o, alpha, dcT, Om, Linv, Kinv, dmeanT, None, None,
                                               ctx.needs_input_grad[7], g_ext=g_ext)
        # the coordinates' gradient goes back to the warp GP along the fp64 tensor when there is one: its
        # backward amplifies the rounding noise of an fp32 gradient by cond(K_uu) (measured: 1e-4 -> 1e-7 on
        # the warp GP's parameter gradients at M = 1000)
        need_x = ctx.needs_input_grad[2] or ctx.via_x64
        # covariance backward: fp32 panel and inputs as stored, fp64 arithmetic, partial sums and results (the
        # hyper-parameter and coordinate gradients are sums over M x C terms that cancel against dK_uu's)
        dZ, dX, dpar = o.kmat_bwd(ctx.kind, Zb, Xb, lsb, var32, dKuf, need_dX=need_x, out_dtype=torch.float64)
        dvar = dpar[1:2] + dvar_s.double()  # sigma^2 enters the covariance and var = sigma^2 - q + v
        dX64 = None
        if ctx.via_x64 and dX is not None:
            dX64, dX = dX, None
        return (
            None,
            dZ.to(zdt) if ctx.needs_input_grad[1] else None,
            dX.to(xdt) if (ctx.needs_input_grad[2] and dX is not None) else None,
            dpar[0].to(ldt).reshape(lshape) if ctx.needs_input_grad[3] else None,
            dvar.to(vdt).reshape(vshape) if ctx.needs_input_grad[4] else None,
            dKuu.to(kdt),
            ddc.to(ddt),
            dOm.to(odt) if dOm is not None else None,
            None, None, dX64,
        )


class SGPWarpLayerFn(torch.autograd.Function):
    """The warp GP of one view as ONE node, all fp64: covariance, projection, mean / variance forms, the
    view's linear mean function and the reparameterised draws G = mean + var eps (SGPCoreFn + WarpSampleFn
    with the covariance fused in and without the autograd edges between them).  Returns G_mean [n,D], G_samples [S,n,D]
    (fp32) and the per-block variance flags."""

    @staticmethod
    def forward(ctx, kind, Z, X, ls_u, var_u, Kuu, dc, Omega, fac, slopes, intercept, eps):
        o = ops()
        f64 = torch.float64
        Zs, Xs, lss, vars_ = _cov_inputs(Z, X, ls_u, var_u, f64)
        Kuf = o.kmat(kind, Zs, Xs, lss, vars_, 0.0, dtype=f64)
        alpha, q = _project(o, fac, Kuf, f64)
        del Kuf
        dcT = dc.detach().to(f64).contiguous()
        Om = Omega.detach()
        # D <= 3 outputs: keep W_j = Omega_j alpha (D x M x n fp64) for the backward instead of
        # recomputing D products there; the mean term rides on the pass over alpha that closes the form
        v, W, meanT = o.quadform_fwd_keep(alpha, Om, dcT)
        sl, ic = slopes.detach(), intercept.detach()
        Gmean, Gs, bad, Gs64 = o.warp_sample_fwd(meanT, v, q, vars_, Xs, sl, ic, eps)
        ctx.save_for_backward(alpha, dcT, Om, fac.Linv, fac.Kinv, Zs, Xs, lss, vars_, eps, W)
        ctx.kind = kind
        ctx.meta = (Z.dtype, X.dtype, ls_u.dtype, ls_u.shape, var_u.dtype, var_u.shape, Kuu.dtype,
                    dc.dtype, Omega.dtype, slopes.dtype, intercept.dtype)
        ctx.mark_non_differentiable(bad)
        ctx.set_materialize_grads(False)  # G_mean usually takes no part in the loss: no zero-fill launches
        return Gmean, Gs, bad, Gs64

    @staticmethod
    def backward(ctx, dGmean, dGs, _dbad, dGs64):
        o = ops()
        alpha, dcT, Om, Linv, Kinv, Zb, Xb, lsb, varb, eps, W = ctx.saved_tensors
        zdt, xdt, ldt, lshape, vdt, vshape, kdt, ddt, odt, sdt, idt = ctx.meta
        # the draws feed the data GP through their fp64 copy (gradient dGs64) and a caller's own losses
        # through the fp32 API tensor (dGs): the sampler's backward adds the two
        dmeanT, g, qbar, dvar_s, dslopes, dint = o.warp_sample_bwd(dGmean, dGs, eps, varb, Xb, dGs64)
        dKuu, dKuf, ddc, dOm = _layer_backward(o, alpha, dcT, Om, Linv, Kinv, dmeanT, g, qbar,
                                               ctx.needs_input_grad[7], W=W)
        need_x = ctx.needs_input_grad[2]
        dZ, dX, dpar = o.kmat_bwd(ctx.kind, Zb, Xb, lsb, varb, dKuf, need_dX=need_x)
        dvar = dpar[1:2] + dvar_s.to(dpar.dtype)  # variance enters the covariance and the sampler
        return (
            None,
            dZ.to(zdt) if ctx.needs_input_grad[1] else None,
            dX.to(xdt) if (need_x and dX is not None) else None,
            dpar[0].to(ldt).reshape(lshape) if ctx.needs_input_grad[3] else None,
            dvar.to(vdt).reshape(vshape) if ctx.needs_input_grad[4] else None,
            dKuu.to(kdt),
            ddc.to(ddt),
            dOm.to(odt) if dOm is not None else None,
            None,
            dslopes.to(sdt),
            dint.to(idt),
            None,
        )


class MeanResidFn(torch.autograd.Function):
    """Linear mean function at the inducing points and the variational residual of one view
    (vgpsa.py:283-289, 296): (Z, slopes, intercept, delta) -> mu_z = scale (Z slopes + intercept)
    [fp32, not differentiable: it is only kept as the reference's ``mu_z_G`` attribute] and
    resid = delta - mu_z [fp64], one launch each way."""

    @staticmethod
    def forward(ctx, Z, slopes, intercept, delta, scale):
        o = ops()
        mu, resid = o.mean_resid_fwd(Z.detach(), slopes.detach(), intercept.detach(), delta.detach(), scale)
        ctx.save_for_backward(Z.detach(), slopes.detach())
        ctx.scale = scale
        ctx.meta = (Z.dtype, slopes.dtype, intercept.dtype, delta.dtype)
        ctx.mark_non_differentiable(mu)
        ctx.set_materialize_grads(False)
        return mu, resid

    @staticmethod
    def backward(ctx, _dmu, dresid):
        o = ops()
        Z, slopes = ctx.saved_tensors
        zdt, sdt, idt, ddt = ctx.meta
        if dresid is None:
            return None, None, None, None, None
        ddelta, dZ, dslopes, dint = o.mean_resid_bwd(dresid, Z, slopes, ctx.scale)
        return dZ.to(zdt), dslopes.to(sdt), dint.to(idt), ddelta.to(ddt), None


class WarpSampleFn(torch.autograd.Function):
    """G_mean, G_samples of the warp GP (vgpsa.py:186-191, 334-351) with the view's linear mean function
    evaluated in the same kernel.  var is used as the std.  Outputs fp32; internals
    fp64; the fp32 parameters are read as stored."""

    @staticmethod
    def forward(ctx, meanT, v, q, var_u, X, slopes, intercept, eps):
        o = ops()
        Xd, var_d = X.detach(), var_u.detach().reshape(1)
        Gmean, Gs, bad, Gs64 = o.warp_sample_fwd(meanT.detach(), v.detach(), q.detach(), var_d, Xd,
                                                 slopes.detach(), intercept.detach(), eps)
        ctx.save_for_backward(eps, var_d, Xd)
        ctx.vmeta = (var_u.dtype, var_u.shape, slopes.dtype, intercept.dtype)
        ctx.mark_non_differentiable(bad)
        ctx.set_materialize_grads(False)
        return Gmean, Gs, bad, Gs64

    @staticmethod
    def backward(ctx, dGmean, dGs, _dbad, dGs64):
        o = ops()
        eps, var_d, Xd = ctx.saved_tensors
        dmeanT, g, qbar, dvar, dslopes, dint = o.warp_sample_bwd(dGmean, dGs, eps, var_d, Xd, dGs64)
        vdt, vshape, sdt, idt = ctx.vmeta
        return (dmeanT, g, qbar, dvar.to(vdt).reshape(vshape), None, dslopes.to(sdt), dint.to(idt), None)


class DataSampleFn(torch.autograd.Function):
    """F = mean + sqrt(var) eps of the data GP (vgpsa.py:197-204, 423-426)."""

    @staticmethod
    def forward(ctx, meanT, v, q, var_u, eps):
        o = ops()
        var32 = var_u.detach().float().reshape(1)
        F, Sigma = o.data_sample_fwd(meanT.detach(), v.detach(), q.detach().double(), var32, eps)
        ctx.save_for_backward(eps, Sigma, var32)
        ctx.vmeta = (var_u.dtype, var_u.shape, q.dtype)
        return F

    @staticmethod
    def backward(ctx, dF):
        o = ops()
        eps, Sigma, var32 = ctx.saved_tensors
        g_ext, dmeanT, dvar = o.data_sample_bwd(dF.contiguous(), eps, Sigma, var32)
        vdt, vshape, qdt = ctx.vmeta
        return dmeanT, g_ext[:-1], g_ext[-1].to(qdt), dvar.to(vdt).reshape(vshape), None


class MatmulFn(torch.autograd.Function):
    """F_obs = F_latent @ W (LMC, vgpsa.py:428-432) on the HIP gemm."""

    @staticmethod
    def forward(ctx, F, W):
        o = ops()
        F2 = F.detach().reshape(-1, F.shape[-1]).contiguous()
        Wc = W.detach().contiguous()
        ctx.save_for_backward(F2, Wc)
        ctx.fshape = F.shape
        return o.gemm(F2, Wc).reshape(*F.shape[:-1], W.shape[1])

    @staticmethod
    def backward(ctx, dO):
        o = ops()
        F2, Wc = ctx.saved_tensors
        dO2 = dO.contiguous().reshape(-1, dO.shape[-1])
        dF = o.gemm(dO2, Wc, transB=True).reshape(ctx.fshape) if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            dW = o.gemm(F2, dO2, transA=True, splitk=o.pick_splitk(F2.shape[0], Wc.shape[0], Wc.shape[1]))
        return dF, dW


class LogLikFn(torch.autograd.Function):
    """sum log N(Y; F, scale) / S with scale = exp(noise_u) + 1e-5 used as a std (vgpsa.py:532-538)."""

    @staticmethod
    def forward(ctx, F, Y, noise_u):
        o = ops()
        nu = noise_u.detach().float().reshape(1)
        Fc, Yc = F.detach().contiguous(), Y.detach().contiguous()
        ctx.save_for_backward(Fc, Yc, nu)
        ctx.nmeta = (noise_u.dtype, noise_u.shape)
        return o.loglik_fwd(Fc, Yc, nu).reshape(())

    @staticmethod
    def backward(ctx, gout):
        o = ops()
        Fc, Yc, nu = ctx.saved_tensors
        dF, dn = o.loglik_bwd(Fc, Yc, nu, gout.detach().double().reshape(1))
        ndt, nshape = ctx.nmeta
        return dF, None, dn.to(ndt).reshape(nshape)


class ElboFn(torch.autograd.Function):
    """loss = -(sum of the modalities' log-likelihoods) + kl_scale * (sum of the KL terms), fp32 scalar
    (vgpsa.py:540): one launch each way instead of the sum / neg / scale / add / cast chain."""

    @staticmethod
    def forward(ctx, ll, kl, kl_scale):
        o = ops()
        llv, klv = ll.detach().double().reshape(-1), kl.detach().double().reshape(-1)
        ctx.shapes = (ll.shape, kl.shape, ll.dtype, kl.dtype)
        ctx.kl_scale = float(kl_scale)
        return o.elbo_fwd(llv, klv, kl_scale).reshape(())

    @staticmethod
    def backward(ctx, gloss):
        o = ops()
        lshape, kshape, ldt, kdt = ctx.shapes
        n_ll, n_kl = 1, 1
        for d in lshape:
            n_ll *= d
        for d in kshape:
            n_kl *= d
        dll, dkl = o.elbo_bwd(gloss.detach(), n_ll, n_kl, ctx.kl_scale)
        return dll.reshape(lshape).to(ldt), dkl.reshape(kshape).to(kdt), None


class KLPlan:
    """Static pairing of the KL terms of a model with the matrices of its factorisation batch
    (priors at batch positions 0..P-1, every variational covariance behind them, in forward's order).
    ``prior_of_term[t]``: position of term t's prior in the batch, or -1 for an absent term (fixed view)."""

    def __init__(self, prior_of_term, n_priors, device):
        T = len(prior_of_term)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
        self.T, self.P = T, n_priors
        self.om_idx = i32([n_priors + t for t in range(T)])
        self.pr_idx = i32(list(prior_of_term))
        order, off = [], [0]
        for p in list(range(n_priors)) + [-1]:  # last group: the absent terms
            order += [t for t in range(T) if prior_of_term[t] == p]
            off.append(len(order))
        self.pr_list, self.grp_off, self.order = i32(list(range(n_priors))), i32(off), i32(order)


class MvnKLGroupedFn(torch.autograd.Function):
    """Every KL term of the step in one launch each way (same formulas as MvnKLFn).  Inputs: D [T,M]
    (d_t rows, fp64), the P prior matrices, then the variational covariance tensors in batch order;
    ``batch`` = (mats, inv, logdet) of the whole factorisation batch.  Returns kl [T]."""

    @staticmethod
    def forward(ctx, plan, batch, D, *mats_in):
        o = ops()
        mats, inv, logdet = batch
        Dc = D.detach().contiguous()
        kl, KD = o.mvn_kl_grouped_fwd(mats, inv, logdet, plan, Dc)
        ctx.save_for_backward(mats, inv, Dc, KD)
        ctx.plan = plan
        ctx.shapes = [tuple(m.shape) for m in mats_in]
        return kl

    @staticmethod
    def backward(ctx, gkl):
        o = ops()
        mats, inv, Dc, KD = ctx.saved_tensors
        plan = ctx.plan
        dOm, dD, S = o.mvn_kl_grouped_bwd(mats, inv, plan, Dc, KD, gkl.double().contiguous())
        Kinv = inv[: plan.P]
        dK = o.gemm(o.gemm(Kinv, S), Kinv, alpha=0.5)
        grads, used = [], 0
        for i, shp in enumerate(ctx.shapes):
            if i < plan.P:
                grads.append(dK[i].reshape(shp))
            else:
                n = shp[0] if len(shp) == 3 else 1
                grads.append(dOm[used : used + n].reshape(shp))
                used += n
        return (None, None, dD) + tuple(grads)


class MvnKLFn(torch.autograd.Function):
    """KL( N(delta_l, Omega_l) || N(mu_l, K) ) for l = 1..L, fp64 (vgpsa.py:498-530; torch's MVN-MVN
    formula): 0.5 [ logdet K - logdet Omega_l + tr(K^-1 Omega_l) + d_l^T K^-1 d_l - M ], d = delta - mu.
    Inputs: Kuu [M,M] (for the gradient path), Dm [M,L], Omega [L,M,M]; ``fac`` = Factor(Kuu);
    ``ofac`` = (Omega^-1 [L,M,M], logdet Omega [L]) from factor_batch."""

    @staticmethod
    def forward(ctx, Kuu, Dm, Omega, fac, ofac):
        o = ops()
        Om = Omega.detach()
        D64 = Dm.detach().double().contiguous()
        Oinv, logdetO = ofac
        kl, KD = o.mvn_kl_fwd(fac.Kinv, fac.logdet, Om, logdetO, D64)
        ctx.save_for_backward(Kuu.detach(), Om, Oinv, KD, fac.Kinv, D64)
        ctx.meta = (Dm.dtype,)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        o = ops()
        Kuu, Om, Oinv, KD, Kinv, D64 = ctx.saved_tensors
        dOm, dDm, Sp = o.mvn_kl_bwd(Kuu, Kinv, Om, Oinv, D64, KD, gkl.double().contiguous())
        dK = o.gemm(o.gemm(Kinv, Sp), Kinv, alpha=0.5)  # 0.5 K^-1 [(sum g) K - sum_l g_l (Omega_l + d d^T)] K^-1
        return dK, dDm.to(ctx.meta[0]), dOm, None, None
