"""Tensor-level wrappers over the C ABI (include/gpsa_hip.h).

PyTorch is used here only as plumbing: it owns device memory (torch.empty), the HIP stream
(torch.cuda.current_stream) and autograd bookkeeping.  Every numerical kernel below is a
hand-written HIP kernel in csrc/, reached through ctypes with raw device pointers.
"""
import os

import torch

from . import _lib

F32, F64 = 0, 1
KINDS = {"rbf": 0, "matern12": 1, "matern32": 2}


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = torch._C._cuda_getCurrentRawStream  # (device index) -> hipStream_t as int


class HipOps:
    """All ops run on the current HIP stream of the tensors' device; outputs are freshly allocated."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self._scratch = {}
        self._wsq_cache = {}
        if not torch.cuda.is_available():
            raise _lib.GpsaHipError("no HIP device visible: the GPSA hot path has no CPU fallback")

    # ------------------------------------------------------------------ plumbing
    # (host time per launch matters: a 1/8 shard of the headline step is bound by the ~250 launches'
    #  host cost, not by the GPU - so no Stream objects and no allocator round trip per call)
    @staticmethod
    def _stream(t):
        """raw hipStream_t of the current stream of t's device"""
        return _raw_stream(t.get_device())

    def _wsq(self, name, *args):
        """memoised *_workspace(...) query: the sizes depend on the arguments only, and a step asks the same
        two dozen questions every time (an FFI call each otherwise)"""
        key = (name,) + args
        v = self._wsq_cache.get(key)
        if v is None:
            v = self._wsq_cache[key] = int(getattr(self.lib, name)(*args))
        return v

    def _ws(self, nbytes, like):
        """scratch of >= nbytes for ONE launch sequence on the current stream.  One buffer per
        (device, stream), grown on demand: launches on a stream are ordered, scratch never outlives its
        call, so consecutive calls can share it."""
        dev = like.get_device()
        stream = _raw_stream(dev)
        # Inside a stream capture the buffer is the CAPTURE's own (round 6): a block allocated while capturing lives in
        # that graph's private pool; cached per stream alone, the next capture on the same stream baked it into a second
        # graph, and once the first graph was destroyed the allocator handed the block out again under the second
        # one's replays.  Nor may a capture bake the eager buffer: a later, larger request frees that one.  Entries of
        # finished captures are dropped here - that only returns their blocks to their own graphs' pools.
        cid = int(self.lib.gpsa_stream_capture_id(stream))
        key = (dev, stream, cid)
        if cid and key not in self._scratch:
            for k in [k for k in self._scratch if k[2] not in (0, cid)]:
                del self._scratch[k]
            self.__dict__["_retired"] = []
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < nbytes:
            # doubling only while that is cheap: the large-M plans ask for tens of GB
            grow = max(int(nbytes), 1 << 20, 0 if buf is None else min(2 * buf.numel(), int(nbytes) + (256 << 20)))
            if buf is not None:
                if cid:
                    # a block must not go back to the allocator inside a capture (its stream bookkeeping records
                    # events): the old buffer outlives the capture in this list
                    self.__dict__.setdefault("_retired", []).append(buf)
                else:
                    self._scratch[key] = None  # returned before the larger one is requested
                del buf
            buf = self._scratch[key] = torch.empty(grow, dtype=torch.uint8, device=like.device)
        return buf

    @staticmethod
    def _c(t):
        return t if t.is_contiguous() else t.contiguous()

    # ------------------------------------------------------------------ covariance matrices
    @staticmethod
    def _cov_args(Z, X, ls_u, var_u):
        """coordinates and hyper-parameters in ONE storage dtype (fp32 parameters pass through)"""
        dt = Z.dtype
        fix = lambda t: t if t.dtype == dt else t.to(dt)
        return HipOps._c(Z), HipOps._c(fix(X)), fix(ls_u), fix(var_u)

    def kmat(self, kind, Z, X, ls_u, var_u, jitter=0.0, dtype=None, out=None):
        """K = k(Z, X) computed and stored in ``dtype`` (default: Z's) from inputs of Z's dtype"""
        in_dt = None
        if X.dtype == torch.float64 and Z.dtype == torch.float32 and dtype == torch.float64:
            # fp32 parameters next to the warp GP's unrounded fp64 draws: read both as stored
            Z, X, in_dt = self._c(Z), self._c(X), 2  # GPSA_F32_X64
        else:
            Z, X, ls_u, var_u = self._cov_args(Z, X, ls_u, var_u)
        dtype = dtype or Z.dtype
        M, D = Z.shape
        Cn = X.shape[0]
        K = torch.empty(M, Cn, dtype=dtype, device=Z.device) if out is None else out
        assert K.shape == (M, Cn) and K.dtype == dtype and K.is_contiguous()
        rc = self.lib.gpsa_kmat(_dt(K), _dt(Z) if in_dt is None else in_dt, KINDS[kind], _p(Z), M, _p(X), Cn, D,
                                _p(ls_u), _p(var_u), float(jitter), _p(K), self._stream(Z))
        _lib.check(rc, "gpsa_kmat")
        return K

    def kmat_bwd(self, kind, Z, X, ls_u, var_u, Kbar, need_dX=True, same=False, out_dtype=None):
        """gradients in Z's dtype (``out_dtype`` = fp64 with fp32 inputs and an fp64 Kbar: stored as fp64);
        the arithmetic and partial sums run in Kbar's dtype.  ``same``: Z and X are the same points (K_uu):
        returns dZ + dX as dZ, and None for dX."""
        Z, X, ls_u, var_u = self._cov_args(Z, X, ls_u, var_u)
        Kbar = self._c(Kbar)
        M, D = Z.shape
        Cn = X.shape[0]
        odt, in_dt, kdt = Z.dtype, _dt(Z), _dt(Kbar)
        if out_dtype == torch.float64 and Z.dtype == torch.float32:
            # fp64 arithmetic and gradients from fp32 inputs; Kbar in either precision
            odt, in_dt, kdt = torch.float64, (3 if Kbar.dtype == torch.float64 else 4), F64
        dZ = torch.empty(Z.shape, dtype=odt, device=Z.device)
        dX = torch.empty(X.shape, dtype=odt, device=Z.device) if (need_dX and not same) else None
        dpar = torch.empty(2, dtype=odt, device=Z.device)
        wsb = self._wsq("gpsa_kmat_bwd_workspace", kdt, M, Cn, D)
        ws = self._ws(wsb, Z)
        rc = self.lib.gpsa_kmat_bwd(kdt, in_dt, KINDS[kind], _p(Z), M, _p(X), Cn, D, _p(ls_u),
                                    _p(var_u), _p(Kbar), int(bool(same)), _p(dZ), _p(dX), _p(dpar), _p(ws),
                                    ws.numel(), self._stream(Z))
        _lib.check(rc, "gpsa_kmat_bwd")
        return dZ, dX, dpar

    # ------------------------------------------------------------------ dense products
    def gemm(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None, splitk=1):
        """out = alpha * op(A) @ op(B) + beta * out.  A, B: 2-D, or 3-D batched (a 2-D operand is
        broadcast over the batch).  Last-dim stride must be 1."""
        batched = A.dim() == 3 or B.dim() == 3
        nb = (A.shape[0] if A.dim() == 3 else B.shape[0]) if batched else 1
        A = A if A.stride(-1) == 1 else A.contiguous()
        B = B if B.stride(-1) == 1 else B.contiguous()
        ar, ac = A.shape[-2], A.shape[-1]
        br, bc = B.shape[-2], B.shape[-1]
        m, k = (ac, ar) if transA else (ar, ac)
        k2, n = (bc, br) if transB else (br, bc)
        assert k == k2, (A.shape, B.shape, transA, transB)
        if out is None:
            assert beta == 0.0
            out = torch.empty((nb, m, n) if batched else (m, n), dtype=A.dtype, device=A.device)
        if splitk == 1 and nb == 1 and 128 <= k <= 1024 and ((m + 63) // 64) * ((n + 63) // 64) <= 32:
            splitk = min(4, k // 48)  # small M x M x M products are latency-bound: spread the k loop
        sA = A.stride(0) if A.dim() == 3 else 0
        sB = B.stride(0) if B.dim() == 3 else 0
        sC = out.stride(0) if out.dim() == 3 else 0
        wsb = self._wsq("gpsa_gemm_workspace", _dt(A), m, n, nb, splitk) if splitk > 1 else 0
        ws = self._ws(wsb, A) if splitk > 1 else None
        rc = self.lib.gpsa_gemm(_dt(A), int(transA), int(transB), m, n, k, float(alpha), _p(A),
                                A.stride(-2), sA, _p(B), B.stride(-2), sB, float(beta), _p(out),
                                out.stride(-2), sC, nb, splitk, _p(ws), wsb if ws is not None else 0,
                                self._stream(A))
        _lib.check(rc, "gpsa_gemm")
        return out

    @staticmethod
    def pick_splitk(k, m, n):
        """split the reduction dim so that a skinny product still fills the chip"""
        tiles = ((m + 63) // 64) * ((n + 63) // 64)
        s = max(1, min(256, -(-512 // max(tiles, 1)), k // 64))
        return int(s)

    # ------------------------------------------------------------------ variational covariances
    def omega_fwd(self, A, jitter, out=None):
        """A [B,M,M] fp32 -> A A^T + jitter I [B,M,M] fp64 (written into ``out`` when given)"""
        A = self._f32(A)
        Bn, M = A.shape[0], A.shape[-1]
        if out is None:
            out = torch.empty(Bn, M, M, dtype=torch.float64, device=A.device)
        assert out.is_contiguous() and out.dtype == torch.float64 and out.shape == (Bn, M, M)
        rc = self.lib.gpsa_omega_fwd(_p(A), M, Bn, float(jitter), _p(out), self._stream(A))
        _lib.check(rc, "gpsa_omega_fwd")
        return out

    def omega_bwd(self, G, A, symmetric=False):
        """dA = (G + G^T) A : G [B,M,M] fp64, A [B,M,M] fp32 -> fp32; ``symmetric`` promises G = G^T"""
        G = self._c(G if G.dtype == torch.float64 else G.double())
        A = self._f32(A)
        Bn, M = A.shape[0], A.shape[-1]
        dA = torch.empty(Bn, M, M, dtype=torch.float32, device=A.device)
        rc = self.lib.gpsa_omega_bwd(_p(G), _p(A), M, Bn, int(bool(symmetric)), _p(dA), self._stream(A))
        _lib.check(rc, "gpsa_omega_bwd")
        return dA

    # ------------------------------------------------------------------ factorisations (fp64)
    def chol(self, A):
        """A [B,M,M] fp64 (not modified) -> L, logdet [B], info [B] (int32)"""
        L = A.contiguous().clone()
        Bn, M = L.shape[0], L.shape[-1]
        logdet = torch.empty(Bn, dtype=torch.float64, device=A.device)
        info = torch.empty(Bn, dtype=torch.int32, device=A.device)
        rc = self.lib.gpsa_chol_f64(_p(L), M, Bn, _p(logdet), _p(info), self._stream(A))
        _lib.check(rc, "gpsa_chol_f64")
        return L, logdet, info

    def tri_inv(self, L):
        L = self._c(L)
        out = torch.empty_like(L)
        rc = self.lib.gpsa_tri_inv_f64(_p(L), _p(out), L.shape[-1], L.shape[0], self._stream(L))
        _lib.check(rc, "gpsa_tri_inv_f64")
        return out

    def chol_inv(self, A):
        """A [B,M,M] fp64 (not modified) -> L^-1 [B,M,M], logdet [B], info [B]; one fused register-resident
        sweep for M <= 256, the blocked factorisation (that sweep per diagonal block + fp64-MFMA
        products) above that."""
        A = self._c(A)
        Bn, M = A.shape[0], A.shape[-1]
        Linv = torch.empty_like(A)
        logdet = torch.empty(Bn, dtype=torch.float64, device=A.device)
        info = torch.empty(Bn, dtype=torch.int32, device=A.device)
        if M > 256:
            ws = self._ws(self._wsq("gpsa_chol_inv_blocked_workspace", M, Bn), A)
            rc = self.lib.gpsa_chol_inv_blocked_f64(_p(A), _p(Linv), M, Bn, _p(logdet), _p(info), _p(ws),
                                                    ws.numel(), self._stream(A))
            _lib.check(rc, "gpsa_chol_inv_blocked_f64")
            return Linv, logdet, info
        rc = self.lib.gpsa_chol_inv_f64(_p(A), _p(Linv), M, Bn, _p(logdet), _p(info), self._stream(A))
        _lib.check(rc, "gpsa_chol_inv_f64")
        return Linv, logdet, info

    def chol_inv_sel(self, A, n_always, keep_lo, keep_hi, Linv=None, logdet=None, info=None):
        """``chol_inv`` for a SELECTION of the batch in one launch (gpsa_chol_inv_sel_f64; M <= 256): the matrices
        b < n_always and keep_lo <= b < keep_hi; the other entries of Linv / logdet / info are left as they are"""
        A = self._c(A)
        Bn, M = A.shape[0], A.shape[-1]
        Linv = torch.empty_like(A) if Linv is None else Linv
        logdet = torch.empty(Bn, dtype=torch.float64, device=A.device) if logdet is None else logdet
        info = torch.empty(Bn, dtype=torch.int32, device=A.device) if info is None else info
        rc = self.lib.gpsa_chol_inv_sel_f64(_p(A), _p(Linv), M, Bn, int(n_always), int(keep_lo), int(keep_hi),
                                            _p(logdet), _p(info), self._stream(A))
        _lib.check(rc, "gpsa_chol_inv_sel_f64")
        return Linv, logdet, info

    # ------------------------------------------------------------------ quadratic forms
    def _qf_ws(self, alpha, L):
        M, Cn = alpha.shape
        return self._ws(self._wsq("gpsa_quadform_workspace", _dt(alpha), M, Cn, L), alpha)

    def quadform_fwd(self, alpha, Omega):
        alpha, Omega = self._c(alpha), self._c(Omega)
        M, Cn = alpha.shape
        L = Omega.shape[0]
        v = torch.empty(L, Cn, dtype=alpha.dtype, device=alpha.device)
        ws = self._qf_ws(alpha, L)
        rc = self.lib.gpsa_quadform_fwd(_dt(alpha), _dt(Omega), _p(alpha), _p(Omega), M, Cn, L, _p(v),
                                        _p(ws), ws.numel(), self._stream(alpha))
        _lib.check(rc, "gpsa_quadform_fwd")
        return v

    def quadform_bwd_alpha(self, alpha, Omega, g):
        alpha, Omega, g = self._c(alpha), self._c(Omega), self._c(g)
        M, Cn = alpha.shape
        L = Omega.shape[0]
        out = torch.empty_like(alpha)
        ws = self._qf_ws(alpha, L)
        rc = self.lib.gpsa_quadform_bwd_alpha(_dt(alpha), _dt(Omega), _p(alpha), _p(Omega), _p(g), M, Cn,
                                              L, _p(out), _p(ws), ws.numel(), self._stream(alpha))
        _lib.check(rc, "gpsa_quadform_bwd_alpha")
        return out

    def quadform_fwd_keep(self, alpha, Omega, dcT=None):
        """(v, W[, meanT]): the form, the products W[l] = Omega[l] alpha it is made of (few-output layers)
        and, with ``dcT`` [M,L], the mean term meanT = dcT^T alpha from the same pass over alpha"""
        alpha, Omega = self._c(alpha), self._c(Omega.to(alpha.dtype))
        M, Cn = alpha.shape
        L = Omega.shape[0]
        v = torch.empty(L, Cn, dtype=alpha.dtype, device=alpha.device)
        W = torch.empty(L, M, Cn, dtype=alpha.dtype, device=alpha.device)
        meanT = None
        if dcT is not None:
            dcT = self._c(dcT.to(alpha.dtype))
            assert dcT.shape == (M, L)
            meanT = torch.empty(L, Cn, dtype=alpha.dtype, device=alpha.device)
        rc = self.lib.gpsa_quadform_fwd_keep(_dt(alpha), _p(alpha), _p(Omega), M, Cn, L, _p(v), _p(W),
                                             _p(dcT), _p(meanT), self._stream(alpha))
        _lib.check(rc, "gpsa_quadform_fwd_keep")
        return (v, W) if dcT is None else (v, W, meanT)

    def quadform_elbo(self, alpha, Omega, meanT, q, var_u, eps, Y, noise_u, want_draws=False, delta=None):
        """The data GP's variance, draw, Gaussian likelihood and the backward's abar in one pass over the products
        Omega_l alpha (gpsa_quadform_elbo_f32; alpha fp32 [M,C], Omega [L,M,M], meanT [L,C], q fp64 [C], eps [S,N,L] or
        [C,L], Y [N,L]).  Returns (g [L,C], dmeanT [L,C], abar [M,C], z2 = sum ((Y - F)/s)^2 as an fp64 0-dim tensor), all at
        upstream gradient 1 of loss = -LL; with ``want_draws`` also the draws, transposed: FT [L,C].
        ``delta`` [M,L] instead of ``meanT`` (None): the kernel forms mean = delta^T alpha itself, in the padding row of its
        product (gpsa_quadform_elbo_delta_f32; only where gpsa_quadform_elbo_takes_delta(M))."""
        alpha, Omega = self._c(alpha), self._c(Omega)
        M, Cn = alpha.shape
        L = Omega.shape[0]
        q, eps, Y = self._c(q), self._c(eps), self._c(Y)
        meanT = self._c(meanT) if meanT is not None else None
        delta = self._c(delta) if delta is not None else None
        N = Y.shape[0]
        assert Cn % N == 0 and eps.numel() == Cn * L and Y.shape == (N, L) and q.dtype == torch.float64
        S = Cn // N
        dev = alpha.device
        g = torch.empty(L, Cn, dtype=torch.float32, device=dev)
        dm = torch.empty(L, Cn, dtype=torch.float32, device=dev)
        abar = torch.empty(M, Cn, dtype=torch.float32, device=dev)
        part = torch.empty(self.lib.gpsa_quadform_elbo_parts(), dtype=torch.float64, device=dev)
        wsb = self.lib.gpsa_quadform_elbo_f32_workspace(M, Cn, L)
        if wsb <= 0:
            raise _lib.GpsaHipError("gpsa_quadform_elbo_f32: more than 16 row tiles (M > 256)")
        ws = self._ws(wsb, alpha)
        FT = torch.empty(L, Cn, dtype=torch.float32, device=dev) if want_draws else None
        if delta is not None:
            assert meanT is None and tuple(delta.shape) == (M, L) and delta.dtype == torch.float32
            rc = self.lib.gpsa_quadform_elbo_delta_f32(_dt(Omega), _p(alpha), _p(Omega), M, Cn, L, _p(delta), _p(q),
                                                       _p(var_u), _p(eps), _p(Y), N, S, _p(noise_u), _p(g), _p(dm),
                                                       _p(abar), _p(part), _p(FT), _p(ws), ws.numel(), self._stream(alpha))
            _lib.check(rc, "gpsa_quadform_elbo_delta_f32")
            return (g, dm, abar, part.sum(), FT) if want_draws else (g, dm, abar, part.sum())
        rc = self.lib.gpsa_quadform_elbo_f32(_dt(Omega), _p(alpha), _p(Omega), M, Cn, L, _p(meanT), _p(q), _p(var_u),
                                             _p(eps), _p(Y), N, S, _p(noise_u), _p(g), _p(dm), _p(abar), _p(part),
                                             _p(FT), _p(ws), ws.numel(), self._stream(alpha))
        _lib.check(rc, "gpsa_quadform_elbo_f32")
        return (g, dm, abar, part.sum(), FT) if want_draws else (g, dm, abar, part.sum())

    def lmc_loglik_fused(self, F, W, Y, noise_u):
        """gpsa_lmc_loglik_fused_f32: (sum z^2 as an fp64 0-dim tensor, dLoss/dF [S,N,L], dLoss/dW [L,P]) of the LMC
        likelihood loss = -sum log N(Y; F W, s) / S at upstream gradient 1, F_obs never formed"""
        F, W, Y = self._c(F), self._c(W), self._c(Y)
        S, N, L = F.shape
        P = W.shape[1]
        assert W.shape[0] == L and Y.shape == (N, P)
        dev = F.device
        nparts = int(self.lib.gpsa_quadform_elbo_parts())
        zpart = torch.empty(nparts, dtype=torch.float64, device=dev)
        dF, dW = torch.empty_like(F), torch.empty_like(W)
        wsb = self.lib.gpsa_lmc_loglik_workspace(S * N, L, P, nparts)
        ws = self._ws(wsb, F)
        rc = self.lib.gpsa_lmc_loglik_fused_f32(_p(F), _p(W), _p(Y), _p(noise_u), S, N, L, P, _p(zpart), nparts, _p(dF),
                                                _p(dW), _p(ws), ws.numel(), self._stream(F))
        _lib.check(rc, "gpsa_lmc_loglik_fused_f32")
        return zpart.sum(), dF, dW

    def quadform_bwd_alpha_kept(self, W, g, dcT=None, dmeanT=None):
        """2 sum_l g_l o W_l  (+ dcT dmeanT, the mean term's share of the alpha-gradient, in the same pass)"""
        W, g = self._c(W), self._c(g)
        L, M, Cn = W.shape
        if dcT is not None:
            dcT, dmeanT = self._c(dcT.to(W.dtype)), self._c(dmeanT.to(W.dtype))
            assert dcT.shape == (M, L) and dmeanT.shape == (L, Cn)
        out = torch.empty(M, Cn, dtype=W.dtype, device=W.device)
        rc = self.lib.gpsa_quadform_bwd_alpha_kept(_dt(W), _p(W), _p(g), M, Cn, L, _p(dcT), _p(dmeanT), _p(out),
                                                   self._stream(W))
        _lib.check(rc, "gpsa_quadform_bwd_alpha_kept")
        return out

    def quadform_bwd_omega(self, alpha, g, out_dtype=None):
        """``out_dtype``: storage type of the result (fp64 from fp32 inputs: the MFMA path widens its
        partial sums while adding them; elsewhere a converted copy)"""
        alpha, g = self._c(alpha), self._c(g)
        M, Cn = alpha.shape
        L = g.shape[0]
        odt = alpha.dtype if out_dtype is None else out_dtype
        out = torch.empty(L, M, M, dtype=odt, device=alpha.device)
        ws = self._qf_ws(alpha, L)
        rc = self.lib.gpsa_quadform_bwd_omega(_dt(alpha), _dt(out), _p(alpha), _p(g), M, Cn, L, _p(out),
                                              _p(ws), ws.numel(), self._stream(alpha))
        if rc == _lib.GPSA_EUNSUPPORTED and odt != alpha.dtype:
            return self.quadform_bwd_omega(alpha, g).to(odt)
        _lib.check(rc, "gpsa_quadform_bwd_omega")
        return out

    def quadform_bwd_omega_delta(self, alpha, g, dmeanT, ddelta=None, beta=0.0, out_dtype=torch.float64):
        """gpsa_quadform_bwd_omega_delta_f32: (dOmega [L,M,M], ddelta [M,L] = beta ddelta + alpha dmeanT^T) from one Gram
        launch (the mean term's gradient in the padding row of the kernel's last row tile); None where the shape does not
        allow it (gpsa_quadform_bwd_omega_takes_delta)"""
        alpha, g, dmeanT = self._c(alpha), self._c(g), self._c(dmeanT)
        M, Cn = alpha.shape
        L = g.shape[0]
        if not self.lib.gpsa_quadform_bwd_omega_takes_delta(M, Cn):
            return None
        out = torch.empty(L, M, M, dtype=out_dtype, device=alpha.device)
        if ddelta is None:
            ddelta = torch.zeros(M, L, dtype=torch.float32, device=alpha.device)
        ws = self._qf_ws(alpha, L)
        rc = self.lib.gpsa_quadform_bwd_omega_delta_f32(_dt(out), _p(alpha), _p(g), _p(dmeanT), M, Cn, L, _p(out),
                                                        _p(ddelta), float(beta), _p(ws), ws.numel(), self._stream(alpha))
        _lib.check(rc, "gpsa_quadform_bwd_omega_delta_f32")
        return out, ddelta

    def panel_mm(self, P, X, want_colsq=False, transP=False):
        """Y = op(P) @ X in X's dtype; P [M,M] may be stored in either precision"""
        P, X = self._c(P), self._c(X)
        M, Cn = X.shape
        Y = torch.empty_like(X)
        q = torch.empty(Cn, dtype=X.dtype, device=X.device) if want_colsq else None
        ws = self._ws(max(4 * 256 * 256, 8 * M * M) + 512, X)
        rc = self.lib.gpsa_panel_mm(_dt(X), _dt(P), int(bool(transP)), _p(P), _p(X), M, Cn, _p(Y), _p(q),
                                    _p(ws), ws.numel(), self._stream(X))
        _lib.check(rc, "gpsa_panel_mm")
        return Y, q

    def whiten(self, Kinv, Kuf, out_dtype, want_q=True):
        """alpha = Kinv @ Kuf (fp64 MFMA; Kuf fp32 or fp64, widened on the fly) stored as ``out_dtype``,
        q[c] = Kuf[:,c] . alpha[:,c] (fp64).  Returns None when M is beyond the register-resident kernel
        (callers chain panel_mm)."""
        Kinv, Kuf = self._c(Kinv), self._c(Kuf)
        assert Kinv.dtype == torch.float64
        M, Cn = Kuf.shape
        nbytes = self._wsq("gpsa_whiten_workspace", M)
        if nbytes == 0:
            return None
        alpha = torch.empty(M, Cn, dtype=out_dtype, device=Kuf.device)
        q = torch.empty(Cn, dtype=torch.float64, device=Kuf.device) if want_q else None
        ws = self._ws(nbytes, Kuf)
        rc = self.lib.gpsa_whiten_f64(_p(Kinv), _dt(Kuf), _p(Kuf), M, Cn, _dt(alpha), _p(alpha), _p(q),
                                      _p(ws), nbytes, self._stream(Kuf))
        _lib.check(rc, "gpsa_whiten_f64")
        return alpha, q

    def col_axpy(self, Y, X, d, s=1.0, out=None):
        Y, X, d = self._c(Y), self._c(X), self._c(d)
        M, Cn = X.shape
        out = torch.empty_like(Y) if out is None else out
        rc = self.lib.gpsa_col_axpy(_dt(X), _p(Y), _p(X), _p(d), float(s), M, Cn, _p(out),
                                    self._stream(X))
        _lib.check(rc, "gpsa_col_axpy")
        return out

    # ------------------------------------------------------------------ sampling
    def data_sample_fwd(self, meanT, v, q, var_u, eps):
        meanT, v, q, eps = self._c(meanT), self._c(v), self._c(q.double()), self._c(eps)
        L, Cn = meanT.shape
        F = torch.empty(Cn, L, dtype=torch.float32, device=meanT.device)
        Sigma = torch.empty_like(meanT)
        rc = self.lib.gpsa_data_sample_fwd(_p(meanT), _p(v), _p(q), _p(var_u), _p(eps), Cn, L, _p(F),
                                           _p(Sigma), self._stream(meanT))
        _lib.check(rc, "gpsa_data_sample_fwd")
        return F, Sigma

    def data_sample_bwd(self, dF, eps, Sigma, var_u):
        """-> g_ext [L+1, C] (rows 0..L-1: g[l,c]; row L: qbar[c] = -sum_l g), dmeanT [L,C], dvar [1].
        (one buffer for both; qbar is exactly minus the column sums of g, which the layer backward uses)"""
        dF, eps = self._c(dF), self._c(eps)
        L, Cn = Sigma.shape
        g_ext = torch.empty(L + 1, Cn, dtype=torch.float32, device=Sigma.device)
        dmeanT = torch.empty_like(Sigma)
        dvar = torch.empty(1, dtype=torch.float32, device=Sigma.device)
        ws = self._ws(8 * (Cn // 32 + 2), Sigma)
        rc = self.lib.gpsa_data_sample_bwd(_p(dF), _p(eps), _p(Sigma), _p(var_u), Cn, L, _p(g_ext),
                                           _p(dmeanT), g_ext.data_ptr() + 4 * L * Cn, 0, _p(dvar), _p(ws),  # (0 = GPSA_F32)
                                           ws.numel(), self._stream(Sigma))
        _lib.check(rc, "gpsa_data_sample_bwd")
        return g_ext, dmeanT, dvar

    @staticmethod
    def _f32(t):
        t = t if t.dtype == torch.float32 else t.float()
        return t if t.is_contiguous() else t.contiguous()

    def warp_sample_fwd(self, meanT, v, q, var_u, X, slopes, intercept, eps):
        """-> Gmean [n,D], Gs [S,n,D] (fp32), bad [blocks] (int32 flags), Gs64 [S,n,D] (the draws unrounded)"""
        meanT, v, q, eps = self._c(meanT), self._c(v), self._c(q), self._c(eps)
        var_u, X, slopes, intercept = (self._f32(t) for t in (var_u, X, slopes, intercept))
        D, n = meanT.shape
        S = eps.shape[0]
        Gmean = torch.empty(n, D, dtype=torch.float32, device=meanT.device)
        Gs = torch.empty(S, n, D, dtype=torch.float32, device=meanT.device)
        Gs64 = torch.empty(S, n, D, dtype=torch.float64, device=meanT.device)
        bad = torch.empty((n + 255) // 256, dtype=torch.int32, device=meanT.device)
        rc = self.lib.gpsa_warp_sample_fwd(_p(meanT), _p(v), _p(q), _p(var_u), _p(X), _p(slopes),
                                           _p(intercept), _p(eps), n, D, S, _p(Gmean), _p(Gs), _p(Gs64),
                                           _p(bad), self._stream(meanT))
        _lib.check(rc, "gpsa_warp_sample_fwd")
        return Gmean, Gs, bad, Gs64

    def warp_sample_bwd(self, dGmean, dGs, eps, var_u, X, dGs64=None):
        """-> dmeanT, g [D,n], qbar [n] (fp64); dvar_u [1], dslopes [D,D], dintercept [D] (fp32).
        The draws' gradient is dGs (fp32, may be None) + dGs64 (fp64, may be None)."""
        eps = self._c(eps)
        dGs = None if dGs is None else self._f32(dGs)
        dGs64 = None if dGs64 is None else self._c(dGs64)
        dGmean = None if dGmean is None else self._c(dGmean)
        var_u, X = self._f32(var_u), self._f32(X)
        S, n, D = eps.shape
        dev = eps.device
        f64, f32 = torch.float64, torch.float32
        dmeanT = torch.empty(D, n, dtype=f64, device=dev)
        g = torch.empty(D, n, dtype=f64, device=dev)
        qbar = torch.empty(n, dtype=f64, device=dev)
        dvar = torch.empty(1, dtype=f32, device=dev)
        dslopes = torch.empty(D, D, dtype=f32, device=dev)
        dint = torch.empty(D, dtype=f32, device=dev)
        ws = self._ws(8 * 21 * ((n + 255) // 256) + 64, eps)
        rc = self.lib.gpsa_warp_sample_bwd(_p(dGmean), _p(dGs), _p(dGs64), _p(eps), _p(var_u), _p(X), n, D, S,
                                           _p(dmeanT), _p(g), _p(qbar), _p(dvar), _p(dslopes), _p(dint),
                                           _p(ws), ws.numel(), self._stream(eps))
        _lib.check(rc, "gpsa_warp_sample_bwd")
        return dmeanT, g, qbar, dvar, dslopes, dint

    # ------------------------------------------------------------------ mean function at Z
    def mean_resid_fwd(self, Z, slopes, intercept, delta, scale=1.0):
        """-> mu_z = scale (Z slopes + intercept) [M,D] fp32, resid = delta - mu_z [M,D] fp64"""
        Z, slopes, intercept, delta = (self._f32(t) for t in (Z, slopes, intercept, delta))
        M, D = Z.shape
        mu = torch.empty(M, D, dtype=torch.float32, device=Z.device)
        resid = torch.empty(M, D, dtype=torch.float64, device=Z.device)
        rc = self.lib.gpsa_mean_resid_fwd(_p(Z), _p(slopes), _p(intercept), _p(delta), M, D, float(scale),
                                          _p(mu), _p(resid), self._stream(Z))
        _lib.check(rc, "gpsa_mean_resid_fwd")
        return mu, resid

    def mean_resid_bwd(self, dresid, Z, slopes, scale=1.0):
        """-> ddelta, dZ [M,D], dslopes [D,D], dintercept [D] (fp32)"""
        dresid = self._c(dresid if dresid.dtype == torch.float64 else dresid.double())
        Z, slopes = self._f32(Z), self._f32(slopes)
        M, D = Z.shape
        f32, dev = torch.float32, Z.device
        ddelta = torch.empty(M, D, dtype=f32, device=dev)
        dZ = torch.empty(M, D, dtype=f32, device=dev)
        dslopes = torch.empty(D, D, dtype=f32, device=dev)
        dint = torch.empty(D, dtype=f32, device=dev)
        rc = self.lib.gpsa_mean_resid_bwd(_p(dresid), _p(Z), _p(slopes), M, D, float(scale), _p(ddelta),
                                          _p(dZ), _p(dslopes), _p(dint), self._stream(Z))
        _lib.check(rc, "gpsa_mean_resid_bwd")
        return ddelta, dZ, dslopes, dint

    # ------------------------------------------------------------------ likelihood
    def loglik_fwd(self, F, Y, noise_u):
        F, Y = self._c(F), self._c(Y)
        S, N, P = F.shape
        out = torch.empty(1, dtype=torch.float64, device=F.device)
        ws = self._ws(8 * 4100, F)
        rc = self.lib.gpsa_loglik_fwd(_p(F), _p(Y), _p(noise_u), S, N, P, _p(out), _p(ws), ws.numel(),
                                      self._stream(F))
        _lib.check(rc, "gpsa_loglik_fwd")
        return out

    def loglik_bwd(self, F, Y, noise_u, gout):
        F, Y = self._c(F), self._c(Y)
        S, N, P = F.shape
        dF = torch.empty_like(F)
        dn = torch.empty(1, dtype=torch.float32, device=F.device)
        ws = self._ws(8 * 4100, F)
        rc = self.lib.gpsa_loglik_bwd(_p(F), _p(Y), _p(noise_u), _p(gout), S, N, P, _p(dF), _p(dn),
                                      _p(ws), ws.numel(), self._stream(F))
        _lib.check(rc, "gpsa_loglik_bwd")
        return dF, dn

    # ------------------------------------------------------------------ small helpers
    def bdot(self, A, B):
        """A, B: [batch, ...] or un-batched (broadcast): out[b] = <A[b], B[b]>"""
        nb = A.shape[0] if A.dim() == 3 else B.shape[0]
        A2 = self._c(A)
        B2 = self._c(B)
        n = A2.shape[-1] * A2.shape[-2]
        sA = n if A2.dim() == 3 else 0
        sB = n if B2.dim() == 3 else 0
        out = torch.empty(nb, dtype=A2.dtype, device=A2.device)
        ws = self._ws(8 * 32 * nb, A2)
        rc = self.lib.gpsa_bdot(_dt(A2), _p(A2), sA, _p(B2), sB, n, nb, _p(out), _p(ws), ws.numel(),
                                self._stream(A2))
        _lib.check(rc, "gpsa_bdot")
        return out

    # ------------------------------------------------------------------ KL terms (fp64)
    @staticmethod
    def _lstride(t):
        """(tensor usable as [L, M, M] with row-major M x M blocks, stride between blocks)"""
        M = t.shape[-1]
        if t.stride(-1) == 1 and t.stride(-2) == M:
            return t, t.stride(0)
        t = t.contiguous()
        return t, M * M

    def mvn_kl_fwd(self, Kinv, logdetK, Omega, logdetO, Dm):
        Omega, so = self._lstride(Omega)
        L, M = Omega.shape[0], Omega.shape[-1]
        Kinv, Dm = self._c(Kinv), self._c(Dm)
        kl = torch.empty(L, dtype=torch.float64, device=Kinv.device)
        KD = self.gemm(Kinv, Dm)
        rc = self.lib.gpsa_mvn_kl_fwd(_p(Kinv), _p(logdetK), _p(Omega), so, _p(logdetO), logdetO.stride(0),
                                      _p(Dm), _p(KD), M, L, _p(kl), self._stream(Kinv))
        _lib.check(rc, "gpsa_mvn_kl_fwd")
        return kl, KD

    def mvn_kl_bwd(self, Kuu, Kinv, Omega, Oinv, Dm, KD, g):
        Omega, so = self._lstride(Omega)
        Oinv, si = self._lstride(Oinv)
        L, M = Omega.shape[0], Omega.shape[-1]
        Kuu, Kinv, Dm, KD, g = self._c(Kuu), self._c(Kinv), self._c(Dm), self._c(KD), self._c(g)
        dev = Kinv.device
        dOm = torch.empty(L, M, M, dtype=torch.float64, device=dev)
        dDm = torch.empty(M, L, dtype=torch.float64, device=dev)
        Sp = torch.empty(M, M, dtype=torch.float64, device=dev)
        rc = self.lib.gpsa_mvn_kl_bwd(_p(Kuu), _p(Kinv), _p(Omega), so, _p(Oinv), si, _p(Dm), _p(KD), _p(g),
                                      M, L, _p(dOm), _p(dDm), _p(Sp), self._stream(Kinv))
        _lib.check(rc, "gpsa_mvn_kl_bwd")
        return dOm, dDm, Sp

    def mvn_kl_grouped_fwd(self, mats, inv, logdet, plan, D):
        """all KL terms of a step: -> kl [T], KD [T,M] (see include/gpsa_hip.h; ``plan``: KLPlan)"""
        T, M = D.shape
        kl = torch.empty(T, dtype=torch.float64, device=D.device)
        KD = torch.empty(T, M, dtype=torch.float64, device=D.device)
        rc = self.lib.gpsa_mvn_kl_grouped_fwd(_p(mats), _p(inv), _p(logdet), _p(plan.om_idx), _p(plan.pr_idx),
                                              _p(D), M, T, _p(kl), _p(KD), self._stream(D))
        _lib.check(rc, "gpsa_mvn_kl_grouped_fwd")
        return kl, KD

    def mvn_kl_grouped_bwd(self, mats, inv, plan, D, KD, g):
        """-> dOmega [T,M,M], dD [T,M], S [P,M,M]  (dK_p = 0.5 K_p^-1 S_p K_p^-1)"""
        T, M = D.shape
        dev = D.device
        dOm = torch.empty(T, M, M, dtype=torch.float64, device=dev)
        dD = torch.empty(T, M, dtype=torch.float64, device=dev)
        S = torch.empty(plan.P, M, M, dtype=torch.float64, device=dev)
        rc = self.lib.gpsa_mvn_kl_grouped_bwd(_p(mats), _p(inv), _p(plan.om_idx), _p(plan.pr_list),
                                              _p(plan.grp_off), _p(plan.order), _p(D), _p(KD), _p(g), M, T,
                                              plan.P, _p(dOm), _p(dD), _p(S), self._stream(D))
        _lib.check(rc, "gpsa_mvn_kl_grouped_bwd")
        return dOm, dD, S

    # ------------------------------------------------------------------ ELBO glue
    def elbo_fwd(self, ll, kl, kl_scale):
        """loss [1] fp32 = -sum(ll) + kl_scale * sum(kl);  ll, kl: fp64 vectors"""
        ll, kl = self._c(ll), self._c(kl)
        loss = torch.empty(1, dtype=torch.float32, device=ll.device)
        rc = self.lib.gpsa_elbo_fwd(_p(ll), ll.numel(), _p(kl), kl.numel(), float(kl_scale), _p(loss),
                                    self._stream(ll))
        _lib.check(rc, "gpsa_elbo_fwd")
        return loss

    def elbo_bwd(self, gloss, n_ll, n_kl, kl_scale):
        gloss = self._f32(gloss.reshape(1))
        dll = torch.empty(n_ll, dtype=torch.float64, device=gloss.device)
        dkl = torch.empty(max(n_kl, 1), dtype=torch.float64, device=gloss.device)[:n_kl]
        rc = self.lib.gpsa_elbo_bwd(_p(gloss), n_ll, n_kl, float(kl_scale), _p(dll), _p(dkl),
                                    self._stream(gloss))
        _lib.check(rc, "gpsa_elbo_bwd")
        return dll, dkl

    # ------------------------------------------------------------------ k-means (initialisation)
    def kmeans_assign(self, X, centres, want_d2=False):
        X, centres = self._c(X), self._c(centres)
        N, D = X.shape
        K = centres.shape[0]
        assign = torch.empty(N, dtype=torch.int32, device=X.device)
        d2 = torch.empty(N, dtype=torch.float32, device=X.device) if want_d2 else None
        rc = self.lib.gpsa_kmeans_assign(_p(X), N, D, _p(centres), K, _p(assign), _p(d2), self._stream(X))
        _lib.check(rc, "gpsa_kmeans_assign")
        return assign, d2

    def kmeans_update(self, X, assign, centres):
        """in place: centres[k] <- mean of the points assigned to k; returns counts [K] (int32)"""
        X = self._c(X)
        assert centres.is_contiguous()
        N, D = X.shape
        K = centres.shape[0]
        counts = torch.empty(K, dtype=torch.int32, device=X.device)
        ws = self._ws(self.lib.gpsa_kmeans_workspace(N, D, K), X)
        rc = self.lib.gpsa_kmeans_update(_p(X), _p(assign), N, D, K, _p(centres), _p(counts), _p(ws),
                                         ws.numel(), self._stream(X))
        _lib.check(rc, "gpsa_kmeans_update")
        return counts

    def add_diag(self, A, s):
        assert A.is_contiguous()
        M = A.shape[-1]
        nb = A.numel() // (M * M)
        rc = self.lib.gpsa_add_diag(_dt(A), _p(A), M, nb, float(s), self._stream(A))
        _lib.check(rc, "gpsa_add_diag")
        return A


_ops = None


def get_ops():
    """The process-wide ops backend (HIP).  Fails loudly without the extension or a device."""
    global _ops
    if _ops is None:
        _ops = HipOps()
    return _ops


def set_ops(obj):
    """Test hook: tests/ may install a fake backend to exercise the host logic without a GPU."""
    global _ops
    _ops = obj
