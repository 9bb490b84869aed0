"""TEST-ONLY fake backend: the ``ops`` interface of spatial_alignment_amd/ops.py restated with plain
torch tensor ops, so that the host logic (model classes, autograd node wiring, the hand-derived
backward formulas of engine.py) can be exercised on a machine without a GPU.

Never imported by the package.  Each method documents the contract the matching HIP kernel meets;
the ``-m gpu`` tests check the real kernels against the same contracts.
"""
import math

import torch

JIT2 = 2e-5


def _cov(kind, Z, X, ls_u, var_u):
    ell, var = torch.exp(ls_u[0]), torch.exp(var_u[0])
    d = Z.unsqueeze(1) - X.unsqueeze(0)
    if kind == "rbf":
        u = d / ell
        return var * torch.exp(-0.5 * (u * u).sum(-1))
    dist = torch.sqrt((d * d).sum(-1) + 1e-10)
    if kind == "matern12":
        return var * torch.exp(-0.5 * dist / ell)
    z = math.sqrt(3.0) * dist / ell
    return var * (1 + z) * torch.exp(-z)


class FakeOps:
    name = "fake-cpu"

    def kmat(self, kind, Z, X, ls_u, var_u, jitter=0.0, dtype=None, out=None):
        dtype = dtype or Z.dtype
        K = _cov(kind, Z.to(dtype), X.to(dtype), ls_u.to(dtype), var_u.to(dtype))
        if jitter:
            n = min(K.shape)
            K = K.clone()
            K[range(n), range(n)] += jitter
        if out is not None:
            out.copy_(K)
            return out
        return K

    def kmat_bwd(self, kind, Z, X, ls_u, var_u, Kbar, need_dX=True, same=False, out_dtype=None):
        dt = out_dtype or Kbar.dtype
        Kbar = Kbar.to(dt)
        with torch.enable_grad():
            Zr, Xr = Z.to(dt).clone().requires_grad_(True), X.to(dt).clone().requires_grad_(True)
            lr, vr = ls_u.to(dt).clone().requires_grad_(True), var_u.to(dt).clone().requires_grad_(True)
            K = _cov(kind, Zr, Xr, lr, vr)
            gz, gx, gl, gv = torch.autograd.grad(K, [Zr, Xr, lr, vr], Kbar)
        o = out_dtype or Z.dtype
        if same:
            return (gz + gx).to(o), None, torch.cat([gl.reshape(1), gv.reshape(1)]).to(o)
        return gz.to(o), (gx.to(o) if need_dX else None), torch.cat([gl.reshape(1), gv.reshape(1)]).to(o)

    def gemm(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None, splitk=1):
        a = A.transpose(-1, -2) if transA else A
        b = B.transpose(-1, -2) if transB else B
        r = alpha * (a @ b)
        if out is None:
            return r
        out.copy_(r + beta * out if beta != 0.0 else r)
        return out

    @staticmethod
    def pick_splitk(k, m, n):
        return 1

    def omega_fwd(self, A, jitter, out=None):
        A64 = A.double()
        Om = A64 @ A64.transpose(-1, -2) + jitter * torch.eye(A.shape[-1], dtype=torch.float64)
        if out is not None:
            out.copy_(Om)
            return out
        return Om

    def omega_bwd(self, G, A, symmetric=False):
        G = G.double()
        return ((2.0 * G if symmetric else G + G.transpose(-1, -2)) @ A.double()).float()

    def chol(self, A):
        L, info = torch.linalg.cholesky_ex(A)
        logdet = 2.0 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
        return L, logdet, info.to(torch.int32)

    def tri_inv(self, L):
        eye = torch.eye(L.shape[-1], dtype=L.dtype).expand_as(L)
        return torch.linalg.solve_triangular(L, eye, upper=False)

    def chol_inv(self, A):
        L, logdet, info = self.chol(A)
        return self.tri_inv(L), logdet, info

    def quadform_fwd(self, alpha, Omega):
        return torch.einsum("mc,lmk,kc->lc", alpha, Omega.to(alpha.dtype), alpha)

    def quadform_bwd_alpha(self, alpha, Omega, g):
        return 2.0 * torch.einsum("lc,lmk,kc->mc", g, Omega.to(alpha.dtype), alpha)

    def quadform_fwd_keep(self, alpha, Omega, dcT=None):
        W = torch.einsum("lmk,kc->lmc", Omega.to(alpha.dtype), alpha)
        v = torch.einsum("mc,lmc->lc", alpha, W)
        return (v, W) if dcT is None else (v, W, dcT.to(alpha.dtype).t() @ alpha)

    def quadform_bwd_alpha_kept(self, W, g, dcT=None, dmeanT=None):
        r = 2.0 * torch.einsum("lc,lmc->mc", g, W)
        return r if dcT is None else r + dcT.to(W.dtype) @ dmeanT.to(W.dtype)

    def quadform_bwd_omega(self, alpha, g, out_dtype=None):
        r = torch.einsum("lc,mc,kc->lmk", g, alpha, alpha)
        return r if out_dtype is None else r.to(out_dtype)

    def panel_mm(self, P, X, want_colsq=False, transP=False):
        Y = (P.t() if transP else P).to(X.dtype) @ X
        return Y, ((Y * Y).sum(0) if want_colsq else None)

    def whiten(self, Kinv, Kuf, out_dtype, want_q=True):
        K64 = Kuf.double()
        a = Kinv @ K64
        return a.to(out_dtype), ((K64 * a).sum(0) if want_q else None)

    def col_axpy(self, Y, X, d, s=1.0, out=None):
        r = Y + s * d.unsqueeze(0) * X
        if out is None:
            return r
        out.copy_(r)
        return out

    def data_sample_fwd(self, meanT, v, q, var_u, eps):
        resid = (torch.exp(var_u[0].double()) - q.double()).to(v.dtype)
        Sigma = resid.unsqueeze(0) + v + JIT2
        F = meanT.t() + torch.sqrt(Sigma).t() * eps
        return F.contiguous(), Sigma

    def data_sample_bwd(self, dF, eps, Sigma, var_u):
        g = (dF * eps).t() * 0.5 / torch.sqrt(Sigma)
        g_ext = torch.cat([g, -g.sum(0, keepdim=True)], 0)
        return g_ext, dF.t().contiguous(), (torch.exp(var_u[0]) * g.sum()).reshape(1)

    def warp_sample_fwd(self, meanT, v, q, var_u, X, slopes, intercept, eps):
        Sigma = torch.exp(var_u.reshape(-1)[0].double()) - q.unsqueeze(0) + v + JIT2  # [D,n]
        mu = X.double() @ slopes.double() + intercept.double() + meanT.t()
        Gs = mu.unsqueeze(0) + Sigma.t().unsqueeze(0) * eps.double()
        bad = (~(Sigma > 0)).any().to(torch.int32).reshape(1)
        return mu.float(), Gs.float(), bad, Gs

    def warp_sample_bwd(self, dGmean, dGs, eps, var_u, X, dGs64=None):
        d = torch.zeros(eps.shape, dtype=torch.float64) if dGs is None else dGs.double()
        if dGs64 is not None:
            d = d + dGs64
        dm = d.sum(0)
        if dGmean is not None:
            dm = dm + dGmean.double()
        g = (d * eps.double()).sum(0)  # [n,D]
        dvar = (torch.exp(var_u.reshape(-1)[0].double()) * g.sum()).reshape(1).float()
        return (dm.t().contiguous(), g.t().contiguous(), -g.sum(1), dvar,
                (X.double().t() @ dm).float(), dm.sum(0).float())

    def mean_resid_fwd(self, Z, slopes, intercept, delta, scale=1.0):
        mu = scale * (Z.double() @ slopes.double() + intercept.double())
        return mu.float(), delta.double() - mu

    def mean_resid_bwd(self, dresid, Z, slopes, scale=1.0):
        r = dresid.double()
        return (r.float(), (-scale * r @ slopes.double().t()).float(),
                (-scale * Z.double().t() @ r).float(), (-scale * r.sum(0)).float())

    def loglik_fwd(self, F, Y, noise_u):
        s = torch.exp(noise_u[0].double()) + 1e-5
        z = (Y.double().unsqueeze(0) - F.double()) / s
        ll = (-0.5 * z * z - torch.log(s) - 0.5 * math.log(2 * math.pi)).sum() / F.shape[0]
        return ll.reshape(1)

    def loglik_bwd(self, F, Y, noise_u, gout):
        S = F.shape[0]
        e = torch.exp(noise_u[0].double())
        s = e + 1e-5
        r = Y.double().unsqueeze(0) - F.double()
        dF = gout[0] * r / (s * s) / S
        ds = ((r * r) / s**3 - 1.0 / s).sum() / S
        return dF.float(), (gout[0] * ds * e).float().reshape(1)

    def elbo_fwd(self, ll, kl, kl_scale):
        return (kl_scale * kl.double().sum() - ll.double().sum()).float().reshape(1)

    def elbo_bwd(self, gloss, n_ll, n_kl, kl_scale):
        g = gloss.double().reshape(())
        return (-g).expand(n_ll).clone(), (kl_scale * g).expand(n_kl).clone()

    def mvn_kl_grouped_fwd(self, mats, inv, logdet, plan, D):
        T, M = D.shape
        kl = torch.zeros(T, dtype=torch.float64)
        KD = torch.zeros(T, M, dtype=torch.float64)
        for t in range(T):
            p, o = int(plan.pr_idx[t]), int(plan.om_idx[t])
            if p < 0:
                continue
            KD[t] = inv[p] @ D[t]
            kl[t] = 0.5 * (logdet[p] - logdet[o] + (inv[p] * mats[o]).sum() + D[t] @ KD[t] - M)
        return kl, KD

    def mvn_kl_grouped_bwd(self, mats, inv, plan, D, KD, g):
        T, M = D.shape
        dOm = torch.zeros(T, M, M, dtype=torch.float64)
        dD = torch.zeros(T, M, dtype=torch.float64)
        S = torch.zeros(plan.P, M, M, dtype=torch.float64)
        for pg in range(plan.P):
            p = int(plan.pr_list[pg])
            for q in range(int(plan.grp_off[pg]), int(plan.grp_off[pg + 1])):
                t = int(plan.order[q])
                o = int(plan.om_idx[t])
                dOm[t] = 0.5 * g[t] * (inv[p] - inv[o])
                dD[t] = g[t] * KD[t]
                S[pg] += g[t] * (mats[p] - mats[o] - torch.outer(D[t], D[t]))
        return dOm, dD, S

    def mvn_kl_fwd(self, Kinv, logdetK, Omega, logdetO, Dm):
        M = Omega.shape[-1]
        tr = (Kinv.unsqueeze(0) * Omega).sum((-1, -2))
        KD = Kinv @ Dm
        kl = 0.5 * (logdetK[0] - logdetO + tr + (Dm * KD).sum(0) - M)
        return kl, KD

    def mvn_kl_bwd(self, Kuu, Kinv, Omega, Oinv, Dm, KD, g):
        dOm = (0.5 * g)[:, None, None] * (Kinv.unsqueeze(0) - Oinv)
        dDm = KD * g.unsqueeze(0)
        S = (g[:, None, None] * Omega).sum(0) + (Dm * g.unsqueeze(0)) @ Dm.t()
        return dOm, dDm, g.sum() * Kuu - S

    def bdot(self, A, B):
        return (A * B).sum((-1, -2)).reshape(-1)

    def add_diag(self, A, s):
        A.diagonal(dim1=-2, dim2=-1).add_(s)
        return A
