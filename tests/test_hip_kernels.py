"""GPU: every C-ABI kernel against its contract (tests/fake_ops.py evaluated on the CPU in fp64)."""
import os

import numpy as np
import pytest
import torch

from fake_ops import FakeOps

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
FK = FakeOps()


@pytest.fixture(scope="module")
def hip():
    from spatial_alignment_amd.ops import HipOps

    return HipOps()


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def close(got, want, tol):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = (got - want).norm() / max(want.norm().item(), 1e-30)
    assert err <= tol, f"rel err {err:.3e} > {tol}"


TOL = {torch.float32: 2e-5, torch.float64: 1e-11}


def _lib_einval():
    from spatial_alignment_amd import _lib
    return _lib.GPSA_EINVAL


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind", ["rbf", "matern12", "matern32"])
@pytest.mark.parametrize("M,C,D", [(7, 300, 2), (25, 1000, 1), (200, 777, 3), (33, 33, 2)])
def test_kmat_fwd_bwd(hip, dtype, kind, M, C, D):
    Z, X = rnd(M, D, dtype=dtype, scale=3), rnd(C, D, dtype=dtype, seed=1, scale=3)
    ls, var = torch.tensor([0.4], dtype=dtype), torch.tensor([-0.3], dtype=dtype)
    Kb = rnd(M, C, dtype=dtype, seed=2)
    K = hip.kmat(kind, Z.to(DEV), X.to(DEV), ls.to(DEV), var.to(DEV), 1e-5)
    close(K, FK.kmat(kind, Z.double(), X.double(), ls.double(), var.double(), 1e-5), TOL[dtype])
    dZ, dX, dp = hip.kmat_bwd(kind, Z.to(DEV), X.to(DEV), ls.to(DEV), var.to(DEV), Kb.to(DEV))
    rZ, rX, rp = FK.kmat_bwd(kind, Z.double(), X.double(), ls.double(), var.double(), Kb.double())
    t = TOL[dtype] * 20
    close(dZ, rZ, t)
    close(dX, rX, t)
    close(dp, rp, t)


@pytest.mark.parametrize("kind", ["rbf", "matern12", "matern32"])
@pytest.mark.parametrize("M,C,D", [(25, 1000, 1), (200, 777, 2), (200, 200, 2)])
def test_kmat_fp32_storage_fp64_compute(hip, kind, M, C, D):
    """fp32 coordinates / hyper-parameters read as stored, covariance and its backward in fp64,
    gradients handed back in fp32 (= the fp64 result rounded once)."""
    f32, f64 = torch.float32, torch.float64
    Z, X = rnd(M, D, dtype=f32, scale=3), rnd(C, D, dtype=f32, seed=1, scale=3)
    ls, var = torch.tensor([0.4], dtype=f32), torch.tensor([-0.3], dtype=f32)
    Kb = rnd(M, C, dtype=f64, seed=2)
    K = hip.kmat(kind, Z.to(DEV), X.to(DEV), ls.to(DEV), var.to(DEV), 1e-5, dtype=f64)
    assert K.dtype == f64
    close(K, FK.kmat(kind, Z.double(), X.double(), ls.double(), var.double(), 1e-5), 1e-11)
    dZ, dX, dp = hip.kmat_bwd(kind, Z.to(DEV), X.to(DEV), ls.to(DEV), var.to(DEV), Kb.to(DEV))
    assert dZ.dtype == f32 and dX.dtype == f32 and dp.dtype == f32
    rZ, rX, rp = FK.kmat_bwd(kind, Z.double(), X.double(), ls.double(), var.double(), Kb)
    for a, b in ((dZ, rZ), (dX, rX), (dp, rp)):
        close(a, b, 2e-7)
    if M == C:  # K_uu: both arguments are the same points, the two gradients arrive summed
        sZ, sX, sp = hip.kmat_bwd(kind, Z.to(DEV), X.to(DEV), ls.to(DEV), var.to(DEV), Kb.to(DEV), same=True)
        assert sX is None
        close(sZ, rZ + rX, 2e-7)
        close(sp, rp, 2e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm(hip, dtype, ta, tb):
    m, n, k = 70, 45, 133
    A = rnd(*((k, m) if ta else (m, k)), dtype=dtype)
    B = rnd(*((n, k) if tb else (k, n)), dtype=dtype, seed=3)
    want = FK.gemm(A.double(), B.double(), bool(ta), bool(tb), alpha=0.7)
    close(hip.gemm(A.to(DEV), B.to(DEV), bool(ta), bool(tb), alpha=0.7), want, TOL[dtype])
    C0 = rnd(m, n, dtype=dtype, seed=4)
    out = C0.clone().to(DEV)
    hip.gemm(A.to(DEV), B.to(DEV), bool(ta), bool(tb), alpha=0.7, beta=0.5, out=out)
    close(out, want + 0.5 * C0.double(), TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(10, 300, 200), (200, 2, 1000), (3, 5, 7), (1, 70, 33)])
def test_gemm_with_fewer_than_16_rows_or_columns(hip, dtype, ta, tb, m, n, k):
    """products below one MFMA tile in a direction (the ten-latent-GP mean term, the [M, 2] gradients of the warp GPs):
    they run on the matrix-core kernel too since round 4"""
    A = rnd(*((k, m) if ta else (m, k)), dtype=dtype)
    B = rnd(*((n, k) if tb else (k, n)), dtype=dtype, seed=3)
    want = FK.gemm(A.double(), B.double(), bool(ta), bool(tb), alpha=1.3)
    close(hip.gemm(A.to(DEV), B.to(DEV), bool(ta), bool(tb), alpha=1.3), want, TOL[dtype])
    C0 = rnd(m, n, dtype=dtype, seed=4)
    out = C0.clone().to(DEV)
    hip.gemm(A.to(DEV), B.to(DEV), bool(ta), bool(tb), alpha=1.3, beta=-0.25, out=out)
    close(out, want - 0.25 * C0.double(), TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gemm_batched_broadcast_splitk(hip, dtype):
    A, B = rnd(5, 40, 40, dtype=dtype), rnd(40, 3000, dtype=dtype, seed=1)
    close(hip.gemm(A.to(DEV), B.to(DEV)), A.double() @ B.double(), TOL[dtype])
    X, Yt = rnd(30, 5000, dtype=dtype), rnd(17, 5000, dtype=dtype, seed=2)
    got = hip.gemm(X.to(DEV), Yt.to(DEV), transB=True, splitk=7)
    close(got, X.double() @ Yt.double().t(), TOL[dtype] * 3)
    A3 = rnd(4, 33, 33, dtype=dtype)
    close(hip.gemm(A3.to(DEV), A3.to(DEV), transB=True), A3.double() @ A3.double().transpose(1, 2), TOL[dtype])


@pytest.mark.parametrize("M", [7, 50, 200])
def test_mvn_kl_grouped(hip, M):
    """all KL terms in one launch each way vs the per-term formulas (incl. an absent term)"""
    from spatial_alignment_amd.engine import KLPlan

    f64 = torch.float64
    P, prior = 3, [0, 1, -1, 0, 2, 2, 2, 1, 2]
    T = len(prior)
    A = rnd(P + T, M, M, dtype=f64, seed=M)
    mats = A @ A.transpose(1, 2) / M + 0.1 * torch.eye(M, dtype=f64)
    inv, logdet = torch.linalg.inv(mats), torch.logdet(mats)
    D, g = rnd(T, M, dtype=f64, seed=1), rnd(T, dtype=f64, seed=2)
    plan_d, plan_h = KLPlan(prior, P, DEV), KLPlan(prior, P, "cpu")
    kl, KD = hip.mvn_kl_grouped_fwd(mats.to(DEV), inv.to(DEV), logdet.to(DEV), plan_d, D.to(DEV))
    rkl, rKD = FK.mvn_kl_grouped_fwd(mats, inv, logdet, plan_h, D)
    close(kl, rkl, 1e-12)
    close(KD, rKD, 1e-12)
    assert float(kl[2]) == 0.0
    got = hip.mvn_kl_grouped_bwd(mats.to(DEV), inv.to(DEV), plan_d, D.to(DEV), KD, g.to(DEV))
    want = FK.mvn_kl_grouped_bwd(mats, inv, plan_h, D, rKD, g)
    for a, b in zip(got, want):
        close(a, b, 1e-12)
    assert float(got[0][2].abs().max()) == 0.0 and float(got[1][2].abs().max()) == 0.0


@pytest.mark.parametrize("M,B", [(5, 3), (50, 4), (200, 6), (233, 2), (64, 1)])
def test_omega_fwd_bwd(hip, M, B):
    """Omega = A A^T + 1e-5 I from the fp32 parameter in fp64, and its adjoint dA = (G + G^T) A"""
    A = rnd(B, M, M, seed=M).tril()
    Om = hip.omega_fwd(A.to(DEV), 1e-5)
    assert Om.dtype == torch.float64
    close(Om, FK.omega_fwd(A, 1e-5), 1e-13)
    buf = torch.full((B + 2, M, M), -1.0, dtype=torch.float64, device=DEV)
    hip.omega_fwd(A.to(DEV), 1e-5, out=buf[1 : B + 1])
    assert torch.equal(buf[1 : B + 1], Om) and float(buf[0].max()) == -1.0 and float(buf[-1].max()) == -1.0
    G = rnd(B, M, M, dtype=torch.float64, seed=3)
    dA = hip.omega_bwd(G.to(DEV), A.to(DEV))
    assert dA.dtype == torch.float32
    close(dA, FK.omega_bwd(G, A), 2e-6)
    Gs = G + G.transpose(1, 2)
    close(hip.omega_bwd(Gs.to(DEV), A.to(DEV), symmetric=True), FK.omega_bwd(Gs, A), 2e-6)


@pytest.mark.parametrize("M,C,L", [(200, 1000, 50), (64, 333, 7), (16, 70, 5), (256, 513, 6), (130, 4099, 10),
                                   (300, 1000, 3), (500, 1302, 4), (1000, 640, 2)])
def test_quadform_keep_f32(hip, M, C, L):
    """the data GP's form with its products kept (gpsa_quadform_fwd_keep_f32, opaque buffer) and the streaming
    backward over them (gpsa_quadform_bwd_alpha_kept_f32) against the recomputing pair"""
    al = rnd(M, C, seed=1)
    A = rnd(L, M, M, seed=2, dtype=torch.float64) / M ** 0.5
    Om = A @ A.transpose(1, 2) + 1e-5 * torch.eye(M, dtype=torch.float64)
    g = rnd(L, C, seed=3)
    ald, Omd, gd = al.to(DEV), Om.to(DEV), g.to(DEV)
    st = hip._stream(ald)
    wsb = hip.lib.gpsa_quadform_keep_f32_workspace(M, L)
    nb = hip.lib.gpsa_quadform_keep_f32_bytes(M, C, L)
    if M > 256 and C % 4 and C >= 128:
        # beyond the register-resident kernel the kept rows must be 16-byte aligned: 0 = "do not keep" (the step
        # engine then recomputes through the padded-copy path); the entry point itself still works on such a shape
        assert nb == 0
        nb = L * M * C * 4
    assert wsb > 0 and L * M * C * 4 <= nb <= 1.6 * L * (M + 16) * (C + 256) * 4
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    v = torch.empty(L, C, device=DEV)
    W = torch.full((nb // 4,), float("nan"), device=DEV)
    rc = hip.lib.gpsa_quadform_fwd_keep_f32(1, ald.data_ptr(), Omd.data_ptr(), M, C, L, v.data_ptr(), W.data_ptr(),
                                            ws.data_ptr(), wsb, st)
    assert rc == 0
    Wr = torch.einsum("lmk,kc->lmc", Om, al.double())
    close(v, (Wr * al.double()[None]).sum(1), 2e-6)
    close(hip.quadform_fwd(ald, Omd), v, 3e-6)        # the cheap forward agrees with the kept one
    out = torch.full((M, C), float("nan"), device=DEV)
    rc = hip.lib.gpsa_quadform_bwd_alpha_kept_f32(W.data_ptr(), gd.data_ptr(), M, C, L, None, None, out.data_ptr(), st)
    assert rc == 0
    want = 2.0 * torch.einsum("lc,lmc->mc", g.double(), Wr)
    close(out, want, 2e-6)
    close(out, hip.quadform_bwd_alpha(ald, Omd, gd), 3e-6)
    # with the mean term's share dcT dmeanT in the same pass
    dc, dm = rnd(M, L, seed=5), rnd(L, C, seed=6)
    dcd, dmd = dc.to(DEV), dm.to(DEV)
    out2 = torch.full((M, C), float("nan"), device=DEV)
    rc = hip.lib.gpsa_quadform_bwd_alpha_kept_f32(W.data_ptr(), gd.data_ptr(), M, C, L, dcd.data_ptr(), dmd.data_ptr(),
                                                  out2.data_ptr(), st)
    assert rc == 0
    close(out2, want + dc.double() @ dm.double(), 2e-6)
    assert hip.lib.gpsa_quadform_bwd_alpha_kept_f32(W.data_ptr(), gd.data_ptr(), M, C, L, dcd.data_ptr(), None,
                                                    out2.data_ptr(), st) == _lib_einval()
    if M > 256:  # beyond the register-resident kernel the products are kept row-major
        assert nb == L * M * C * 4


@pytest.mark.parametrize("M,N,S,L", [(200, 700, 3, 50), (64, 333, 1, 7), (16, 70, 2, 5), (100, 5000, 2, 9),
                                     (208, 129, 5, 3), (30, 64, 1, 1), (200, 20000, 2, 4), (5, 1, 1, 2), (200, 3, 2, 1),
                                     (1, 17, 1, 3), (240, 600, 2, 5), (256, 333, 1, 3), (209, 100, 3, 2),
                                     (200, 20000, 5, 50)])  # the headline step's exact launch (column tiles split over
                                                            # workgroups, partial tiles leaving through the slabs)
def test_quadform_elbo(hip, M, N, S, L):
    """variance + draw + Gaussian likelihood + abar in one pass over the products (gpsa_quadform_elbo_f32) against
    the formulas of the separate kernels (elementwise.hip) evaluated in fp64"""
    C = S * N
    al = rnd(M, C, seed=1) / M ** 0.5
    A = rnd(L, M, M, seed=2, dtype=torch.float64) / M ** 0.5
    Om = A @ A.transpose(1, 2) + 1e-5 * torch.eye(M, dtype=torch.float64)
    meanT = rnd(L, C, seed=3)
    var_u, noise_u = torch.tensor([0.3]), torch.tensor([-0.7])
    q = (rnd(C, seed=4, dtype=torch.float64).abs() * 0.2).clamp(max=1.0)
    eps, Y = rnd(S, N, L, seed=5), rnd(N, L, seed=6)
    g, dm, abar, z2, FT = hip.quadform_elbo(al.to(DEV), Om.to(DEV), meanT.to(DEV), q.to(DEV), var_u.to(DEV),
                                            eps.to(DEV), Y.to(DEV), noise_u.to(DEV), want_draws=True)
    big = L * M * C > 2e8  # the fp64 restatement of the headline size: on the device, output by output
    rd = DEV if big else "cpu"
    ad, Om = al.double().to(rd), Om.to(rd)
    meanT, eps, Y, q, var_u, noise_u = (t.to(rd) for t in (meanT, eps, Y, q, var_u, noise_u))
    if big:
        W = None
        v = torch.stack([((Om[l] @ ad) * ad).sum(0) for l in range(L)])
    else:
        W = torch.einsum("lmk,kc->lmc", Om, ad)
        v = (W * ad[None]).sum(1)                                       # [L, C]
    var = (var_u.double().exp() - q)[None] + v + 2e-5
    e = eps.double().reshape(C, L).t()                                  # [L, C]
    F = meanT.double() + var.sqrt() * e
    sN = noise_u.double().exp() + 1e-5
    r = Y.double().t().repeat(1, S) - F                                 # column c = s*N + n  ->  Y[n]
    dF = -r / (sN * sN * S)
    gw = dF * e * 0.5 / var.sqrt()
    close(dm, dF, 3e-6)
    close(g, gw, 3e-6)
    close(FT, F, 1e-6)
    want_abar = 2.0 * torch.einsum("lc,lmc->mc", gw, W) if W is not None else \
        2.0 * sum(Om[l] @ (ad * gw[l][None]) for l in range(L))
    close(abar, want_abar, 5e-6)
    close(z2.reshape(1), ((r / sN) ** 2).sum().reshape(1), 1e-6)
    g2 = hip.quadform_elbo(al.to(DEV), Om.to(DEV), meanT.to(DEV), q.to(DEV), var_u.to(DEV), eps.to(DEV), Y.to(DEV),
                           noise_u.to(DEV))  # without the draws: the same numbers, bit for bit
    assert torch.equal(g2[0], g) and torch.equal(g2[2], abar)
    # the mean formed inside the kernel (delta^T alpha in the product's first padding row): where M allows it, the same
    # results as with the mean handed in - up to the rounding of the mean itself (fp32 matrix cores there, fp64 here)
    delta = rnd(M, L, seed=7)
    takes = bool(hip.lib.gpsa_quadform_elbo_takes_delta(M))
    assert takes == (M % 16 != 0 and M > 16 * ({2: 2, 4: 4, 7: 7, 13: 13, 16: 16}[min(k for k in (2, 4, 7, 13, 16) if 16 * k >= M)] - 1))
    if not takes:
        with pytest.raises(Exception):
            hip.quadform_elbo(al.to(DEV), Om.to(DEV), None, q.to(DEV), var_u.to(DEV), eps.to(DEV), Y.to(DEV), noise_u.to(DEV),
                              delta=delta.to(DEV))
    else:
        mean_d = (delta.double().to(rd).t() @ ad)                       # [L, C]
        g3, dm3, abar3, z3, FT3 = hip.quadform_elbo(al.to(DEV), Om.to(DEV), None, q.to(DEV), var_u.to(DEV), eps.to(DEV),
                                                    Y.to(DEV), noise_u.to(DEV), want_draws=True, delta=delta.to(DEV))
        F3 = mean_d + var.sqrt() * e
        r3 = Y.double().t().repeat(1, S) - F3
        dF3 = -r3 / (sN * sN * S)
        gw3 = dF3 * e * 0.5 / var.sqrt()
        close(FT3, F3, 2e-6)
        close(dm3, dF3, 5e-6)
        close(g3, gw3, 5e-6)
        want3 = 2.0 * torch.einsum("lc,lmc->mc", gw3, W) if W is not None else \
            2.0 * sum(Om[l] @ (ad * gw3[l][None]) for l in range(L))
        close(abar3, want3, 8e-6)
        close(z3.reshape(1), ((r3 / sN) ** 2).sum().reshape(1), 2e-6)
    with pytest.raises(Exception):  # beyond 16 row tiles: refused, not wrong
        hip.quadform_elbo(rnd(272, C).to(DEV), rnd(L, 272, 272).to(DEV), meanT.to(DEV), q.to(DEV), var_u.to(DEV),
                          eps.to(DEV), Y.to(DEV), noise_u.to(DEV))


@pytest.mark.parametrize("S,N,L,P", [(2, 100, 3, 7), (5, 1000, 10, 500), (1, 333, 20, 1100), (3, 17, 8, 512),
                                     (2, 50, 32, 40), (1, 1, 1, 1), (2, 4000, 10, 513), (2, 300, 33, 130),
                                     (1, 200, 64, 70), (5, 40000, 10, 500)])
def test_lmc_loglik_fused(hip, S, N, L, P):
    """the LMC likelihood without F_obs (gpsa_lmc_loglik_fused_f32): sum z^2, dLoss/dF_latent and dLoss/dW against
    autograd of the reference's own expressions (vgpsa.py:428-432, 532-538) in fp64"""
    F = rnd(S, N, L, seed=1)
    W = rnd(L, P, seed=2)
    Y = rnd(N, P, seed=3)
    noise_u = torch.tensor([-0.4])
    z2, dF, dW = hip.lmc_loglik_fused(F.to(DEV), W.to(DEV), Y.to(DEV), noise_u.to(DEV))
    Fd, Wd = F.double().requires_grad_(), W.double().requires_grad_()
    s = noise_u.double().exp() + 1e-5
    Fobs = Fd @ Wd
    ll = torch.distributions.Normal(Fobs, s).log_prob(Y.double()).sum() / S
    (-ll).backward()
    close(z2.reshape(1), (((Y.double() - Fobs) / s) ** 2).sum().detach().reshape(1), 2e-6)
    close(dF, Fd.grad, 5e-6)
    close(dW, Wd.grad, 5e-6)
    z2b, dFb, dWb = hip.lmc_loglik_fused(F.to(DEV), W.to(DEV), Y.to(DEV), noise_u.to(DEV))
    assert torch.equal(dF, dFb) and torch.equal(dW, dWb) and torch.equal(z2, z2b)  # fixed-order reductions
    if L <= 32:
        return


def test_lmc_loglik_fused_refuses_more_than_64_latent_outputs(hip):
    with pytest.raises(Exception):
        hip.lmc_loglik_fused(rnd(1, 8, 65).to(DEV), rnd(65, 4).to(DEV), rnd(8, 4).to(DEV), torch.tensor([0.0]).to(DEV))


def test_lmc_loglik_mfma_equals_vector_kernel(hip, monkeypatch):
    """round 5's matrix-core kernel against round 4's vector-pipe kernel on BASELINE config 3's shape (to rounding:
    different summation orders), with its time next to it"""
    import subprocess
    import sys

    code = (
        "import torch, time, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from spatial_alignment_amd.ops import get_ops\n"
        "hip = get_ops(); g = torch.Generator().manual_seed(1)\n"
        "F = torch.randn(5, 40000, 10, generator=g).cuda(); W = torch.randn(10, 500, generator=g).cuda()\n"
        "Y = torch.randn(40000, 500, generator=g).cuda(); nu = torch.tensor([-0.4]).cuda()\n"
        "for _ in range(3): out = hip.lmc_loglik_fused(F, W, Y, nu)\n"
        "torch.cuda.synchronize(); t0 = time.perf_counter()\n"
        "for _ in range(20): out = hip.lmc_loglik_fused(F, W, Y, nu)\n"
        "torch.cuda.synchronize(); print('MS', 1e3 * (time.perf_counter() - t0) / 20)\n"
        "torch.save([o.cpu() for o in out], sys.argv[1])\n")
    res = {}
    for mode in ("1", "0"):
        path = f"/tmp/lmc_{mode}.pt"
        env = dict(__import__("os").environ, GPSA_LMC_MFMA=mode)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = (torch.load(path), [ln for ln in r.stdout.splitlines() if ln.startswith("MS")][0])
    print("lmc likelihood, 5 x 40000 spots, 10 -> 500 outputs: matrix cores", res["1"][1], "vector pipe", res["0"][1])
    for a, b in zip(res["1"][0], res["0"][0]):
        assert float((a.double() - b.double()).norm() / b.double().norm()) <= 3e-6


@pytest.mark.parametrize("M,n0,n1", [(5, 3, 2), (200, 4, 50), (72, 1, 1)])
def test_omega_two_segments(hip, M, n0, n1):
    """gpsa_omega_fwd2 / _bwd2: two parameter tensors in one launch == the two single-segment calls, bit for bit"""
    A0, A1 = rnd(n0, M, M, seed=M).to(DEV), rnd(n1, M, M, seed=M + 1).to(DEV)
    st = hip._stream(A0)
    O0 = torch.empty(n0, M, M, dtype=torch.float64, device=DEV)
    O1 = torch.empty(n1, M, M, dtype=torch.float64, device=DEV)
    rc = hip.lib.gpsa_omega_fwd2(A0.data_ptr(), n0, O0.data_ptr(), A1.data_ptr(), n1, O1.data_ptr(), M, 1e-5, st)
    assert rc == 0
    assert torch.equal(O0, hip.omega_fwd(A0, 1e-5)) and torch.equal(O1, hip.omega_fwd(A1, 1e-5))
    G0 = rnd(n0, M, M, dtype=torch.float64, seed=5).to(DEV)
    G1 = rnd(n1, M, M, dtype=torch.float64, seed=6).to(DEV)
    for sym in (0, 1):
        d0, d1 = torch.empty_like(A0), torch.empty_like(A1)
        rc = hip.lib.gpsa_omega_bwd2(G0.data_ptr(), A0.data_ptr(), d0.data_ptr(), n0, G1.data_ptr(), A1.data_ptr(),
                                     d1.data_ptr(), n1, M, sym, st)
        assert rc == 0
        assert torch.equal(d0, hip.omega_bwd(G0, A0, symmetric=bool(sym)))
        assert torch.equal(d1, hip.omega_bwd(G1, A1, symmetric=bool(sym)))


@pytest.mark.parametrize("M,B", [(1, 2), (5, 3), (50, 4), (200, 6), (233, 2)])
def test_chol_and_tri_inv(hip, M, B):
    A = rnd(B, M, M, dtype=torch.float64)
    K = A @ A.transpose(1, 2) + 0.1 * torch.eye(M, dtype=torch.float64)
    L, logdet, info = hip.chol(K.to(DEV))
    rL, rld, _ = FK.chol(K)
    assert int(info.abs().max()) == 0
    close(L, rL, 1e-10)
    close(logdet, rld, 1e-12)
    close(hip.tri_inv(L), FK.tri_inv(rL), 1e-9)


@pytest.mark.parametrize("M,B", [(1, 2), (5, 3), (31, 2), (32, 2), (33, 2), (50, 4), (64, 1), (100, 3), (128, 2),
                                 (200, 57), (224, 2), (225, 2), (256, 3), (257, 2), (300, 2), (500, 3),
                                 (513, 1), (1000, 2)])
def test_chol_inv_fused(hip, M, B):
    """register-resident Cholesky + inverse of the factor vs LAPACK (and the unfused pair's results)."""
    A = rnd(B, M, M, dtype=torch.float64, seed=M)
    K = A @ A.transpose(1, 2) / M + 0.05 * torch.eye(M, dtype=torch.float64)
    Kd = K.to(DEV)
    Linv, logdet, info = hip.chol_inv(Kd)
    assert torch.equal(Kd.cpu(), K)  # input untouched
    rL, rld, _ = FK.chol(K)
    assert int(info.abs().max()) == 0
    close(logdet, rld, 1e-12)
    close(Linv, FK.tri_inv(rL), 1e-9)
    assert float(Linv.cpu().triu(1).abs().max()) == 0.0 if M > 1 else True
    eye = torch.eye(M, dtype=torch.float64)
    resid = Linv.cpu() @ K @ Linv.cpu().transpose(1, 2) - eye
    assert float(resid.abs().max()) < 1e-9


@pytest.mark.parametrize("M,B,n_always,lo,hi", [(200, 57, 3, 10, 17), (200, 57, 3, 3, 10), (200, 57, 3, 50, 57),
                                                (64, 9, 1, 4, 4), (240, 6, 2, 3, 5), (200, 12, 0, 5, 9)])
def test_chol_inv_selection(hip, M, B, n_always, lo, hi):
    """gpsa_chol_inv_sel_f64: the priors and ONE rank's own range of the batch in one launch (owner computes): the
    selected entries equal the full call's bit for bit, the others are not touched"""
    A = rnd(B, M, M, dtype=torch.float64, seed=M + B)
    K = (A @ A.transpose(1, 2) / M + 0.05 * torch.eye(M, dtype=torch.float64)).to(DEV)
    full = hip.chol_inv(K)
    Linv = torch.full_like(K, -7.0)
    logdet = torch.full((B,), -7.0, dtype=torch.float64, device=DEV)
    info = torch.full((B,), -7, dtype=torch.int32, device=DEV)
    hip.chol_inv_sel(K, n_always, lo, hi, Linv, logdet, info)
    sel = [b for b in range(B) if b < n_always or lo <= b < hi]
    rest = [b for b in range(B) if b not in sel]
    for got, want in zip((Linv, logdet, info), full):
        assert torch.equal(got[sel], want[sel])
        if rest:
            assert bool((got[rest] == -7).all())


def test_chol_inv_on_covariance_conditioning(hip):
    """K_uu of the warp GP at init (RBF on a grid + 1e-5 jitter, cond ~ 1e7): fp64 residual stays small."""
    g = torch.linspace(0, 10, 15, dtype=torch.float64)
    Z = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)[:200]
    K = torch.exp(-0.5 * torch.cdist(Z, Z) ** 2) + 1e-5 * torch.eye(200, dtype=torch.float64)
    Linv, logdet, info = hip.chol_inv(K.unsqueeze(0).to(DEV))
    assert int(info[0]) == 0
    X = Linv[0].cpu()
    assert float((X @ K @ X.t() - torch.eye(200, dtype=torch.float64)).abs().max()) < 1e-7
    close(logdet, torch.logdet(K).reshape(1), 1e-9)


def test_chol_inv_flags_indefinite(hip):
    K = torch.eye(40, dtype=torch.float64).repeat(3, 1, 1)
    K[1, 7, 7] = -1.0
    K[2, 39, 39] = 0.0
    _, logdet, info = hip.chol_inv(K.to(DEV))
    assert info.cpu().tolist() == [0, 8, 40]
    assert bool(torch.isnan(logdet[1])) and bool(torch.isnan(logdet[2])) and float(logdet[0]) == 0.0


def test_chol_inv_blocked_flags_indefinite(hip):
    """M > 256 (blocked): the first non-positive pivot is reported with its global column, whichever block"""
    K = torch.eye(600, dtype=torch.float64).repeat(4, 1, 1) * 2.0
    K[1, 7, 7] = -1.0       # first block
    K[2, 450, 450] = 0.0    # second block (blocks of 304 columns)
    K[3, 599, 599] = -3.0   # last column of the last block
    Linv, logdet, info = hip.chol_inv(K.to(DEV))
    assert info.cpu().tolist() == [0, 8, 451, 600]
    assert all(bool(torch.isnan(logdet[i])) for i in (1, 2, 3))
    close(logdet[:1], torch.tensor([600 * np.log(2.0)]), 1e-12)
    close(Linv[0], torch.eye(600, dtype=torch.float64) / np.sqrt(2.0), 1e-12)


def test_chol_flags_indefinite(hip):
    K = torch.eye(20, dtype=torch.float64).repeat(2, 1, 1)
    K[1, 7, 7] = -1.0
    _, _, info = hip.chol(K.to(DEV))
    assert info.cpu().tolist() == [0, 8]


QF = [(10, 100, 3), (25, 1000, 5), (50, 333, 2), (100, 500, 4), (200, 2100, 7), (256, 300, 2), (16, 64, 1),
      (300, 130, 2), (200, 50001, 3), (13, 7, 1), (380, 1000, 3), (392, 200, 2), (500, 700, 3), (513, 100, 2),
      (1000, 1040, 2), (641, 4100, 1),
      # M > 256 with more outputs: the LDS-DMA kernels that never write Omega_l alpha (big_quad / big_accum), the
      # alpha-gradient's split over the outputs, partial row blocks and column tiles
      (500, 1300, 20), (1000, 3968, 9), (260, 128, 17), (300, 2052, 33),
      # ... and column counts that are not multiples of 4 (S * N is whatever the data has): padded copies
      (1000, 1302, 3), (600, 131, 5), (700, 2050, 2), (300, 1001, 4)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("M,C,L", QF)
def test_quadform(hip, dtype, M, C, L):
    al = rnd(M, C, dtype=dtype)
    A = rnd(L, M, M, dtype=torch.float64, seed=1, scale=0.3)
    Om = (A @ A.transpose(1, 2)).to(dtype)
    g = rnd(L, C, dtype=dtype, seed=2)
    t = 3e-5 if dtype == torch.float32 else 1e-11
    close(hip.quadform_fwd(al.to(DEV), Om.to(DEV)), FK.quadform_fwd(al.double(), Om.double()), t)
    close(hip.quadform_bwd_alpha(al.to(DEV), Om.to(DEV), g.to(DEV)),
          FK.quadform_bwd_alpha(al.double(), Om.double(), g.double()), t)
    close(hip.quadform_bwd_omega(al.to(DEV), g.to(DEV)), FK.quadform_bwd_omega(al.double(), g.double()), t)
    if dtype == torch.float32:  # fp64 result from fp32 inputs (partials widened while added, or a converted copy)
        wide = hip.quadform_bwd_omega(al.to(DEV), g.to(DEV), out_dtype=torch.float64)
        assert wide.dtype == torch.float64
        close(wide, FK.quadform_bwd_omega(al.double(), g.double()), t)


@pytest.mark.parametrize("M,C,L", [(200, 2100, 7), (200, 20000, 50), (100, 5000, 3), (25, 1000, 5), (10, 100, 1), (250, 260, 2),
                                   (200, 100000, 3), (208, 400, 2), (200, 2102, 2), (300, 400, 2), (120, 4000, 2)])
def test_quadform_bwd_omega_with_ddelta(hip, M, C, L):
    """gpsa_quadform_bwd_omega_delta_f32: the Gram sums and, out of the first padding row of the kernel's last row tile,
    d delta_F = alpha dmean^T (accumulated onto a given ddelta with beta); refused (None) where M fills its last row tile,
    lies before it, is beyond the kernel, or C is not a multiple of 4"""
    al, g, dm = rnd(M, C), rnd(L, C, seed=2), rnd(L, C, seed=3)
    mb = next((k for k in (2, 4, 7, 13, 16) if 16 * k >= M), 0)
    takes = mb != 0 and M % 16 != 0 and M > 16 * (mb - 1) and C % 4 == 0
    assert bool(hip.lib.gpsa_quadform_bwd_omega_takes_delta(M, C)) == takes
    got = hip.quadform_bwd_omega_delta(al.to(DEV), g.to(DEV), dm.to(DEV))
    if not takes:
        assert got is None
        return
    dOm, dd = got
    close(dOm, FK.quadform_bwd_omega(al.double(), g.double()), 3e-5)
    want = al.double() @ dm.double().t()
    close(dd, want, 3e-5)
    assert torch.equal(dOm, hip.quadform_bwd_omega(al.to(DEV), g.to(DEV), out_dtype=torch.float64))  # the same Gram sums
    base = rnd(M, L, seed=4).to(DEV)
    _, dd2 = hip.quadform_bwd_omega_delta(al.to(DEV), g.to(DEV), dm.to(DEV), ddelta=base.clone(), beta=1.0)
    close(dd2, base.double().cpu() + want, 3e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("M,C,L", [(12, 50, 1), (200, 1250, 2), (65, 777, 3)])
def test_quadform_kept_products(hip, dtype, M, C, L):
    """few-output form that keeps W_l = Omega_l alpha, and the backward made of the kept products"""
    al = rnd(M, C, dtype=dtype)
    A = rnd(L, M, M, dtype=torch.float64, seed=1, scale=0.3)
    Om = (A @ A.transpose(1, 2)).to(dtype)
    g = rnd(L, C, dtype=dtype, seed=2)
    t = 3e-5 if dtype == torch.float32 else 1e-11
    v, W = hip.quadform_fwd_keep(al.to(DEV), Om.to(DEV))
    rv, rW = FK.quadform_fwd_keep(al.double(), Om.double())
    assert W.shape == (L, M, C) and W.dtype == dtype
    close(v, rv, t)
    close(W, rW, t)
    close(hip.quadform_bwd_alpha_kept(W, g.to(DEV)), FK.quadform_bwd_alpha(al.double(), Om.double(), g.double()), t)
    dcT, dm = rnd(M, L, dtype=dtype, seed=5), rnd(L, C, dtype=dtype, seed=6)
    close(hip.quadform_bwd_alpha_kept(W, g.to(DEV), dcT.to(DEV), dm.to(DEV)),
          FK.quadform_bwd_alpha(al.double(), Om.double(), g.double()) + dcT.double() @ dm.double(), t)
    # the forward with the layer's mean term from the same pass
    v2, W2, mean = hip.quadform_fwd_keep(al.to(DEV), Om.to(DEV), dcT.to(DEV))
    assert torch.equal(W2, W)
    close(v2, rv, t)
    close(mean, dcT.double().t() @ al.double(), t)


def test_quadform_mfma_matches_generic_large(hip):
    """MFMA path vs the generic tiled path (same inputs, both on the GPU) at a headline-like shape."""
    M, C, L = 200, 20000, 6
    al, g = rnd(M, C).to(DEV), rnd(L, C, seed=2).to(DEV)
    A = rnd(L, M, M, seed=1, scale=0.1).to(DEV)
    Om = A @ A.transpose(1, 2)
    Om64, al64 = Om.double(), al.double()
    want = torch.einsum("mc,lmk,kc->lc", al64, Om64, al64)
    close(hip.quadform_fwd(al, Om), want, 3e-5)
    want_a = 2 * torch.einsum("lc,lmk,kc->mc", g.double(), Om64, al64)
    close(hip.quadform_bwd_alpha(al, Om, g), want_a, 3e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("M,C", [(12, 50), (200, 1500), (64, 257)])
def test_panel_mm_col_axpy(hip, dtype, M, C):
    P, X = rnd(M, M, dtype=dtype), rnd(M, C, dtype=dtype, seed=1)
    Y, q = hip.panel_mm(P.to(DEV), X.to(DEV), want_colsq=True)
    rY, rq = FK.panel_mm(P.double(), X.double(), True)
    close(Y, rY, TOL[dtype] * 2)
    close(q, rq, TOL[dtype] * 2)
    d = rnd(C, dtype=dtype, seed=5)
    close(hip.col_axpy(Y, X.to(DEV), d.to(DEV), 0.5), FK.col_axpy(rY, X.double(), d.double(), 0.5), TOL[dtype] * 2)


@pytest.mark.parametrize("M,C", [(12, 50), (200, 1500), (300, 70)])
@pytest.mark.parametrize("xdt", [torch.float32, torch.float64])
def test_panel_mm_fp64_operand_and_transpose(hip, M, C, xdt):
    """the fp64 factor is read as stored (rounded while packed / converted), optionally transposed"""
    P = rnd(M, M, dtype=torch.float64).tril()
    X = rnd(M, C, dtype=xdt, seed=1)
    tol = TOL[xdt] * 2
    for tp in (False, True):
        Y, q = hip.panel_mm(P.to(DEV), X.to(DEV), want_colsq=True, transP=tp)
        assert Y.dtype == xdt
        rY = (P.t() if tp else P) @ X.double()
        close(Y, rY, tol)
        close(q, (rY * rY).sum(0), tol)


@pytest.mark.parametrize("M,C,L", [(25, 1000, 5), (200, 2100, 3), (300, 130, 2)])
def test_quadform_fp64_omega_fp32_alpha(hip, M, C, L):
    a = rnd(M, C, dtype=torch.float32)
    A = rnd(L, M, M, dtype=torch.float64, seed=1)
    Om = A @ A.transpose(1, 2) / M
    g = rnd(L, C, dtype=torch.float32, seed=2)
    v = hip.quadform_fwd(a.to(DEV), Om.to(DEV))
    assert v.dtype == torch.float32
    close(v, FK.quadform_fwd(a.double(), Om), 3e-5)
    close(hip.quadform_bwd_alpha(a.to(DEV), Om.to(DEV), g.to(DEV)), FK.quadform_bwd_alpha(a.double(), Om, g.double()), 3e-5)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("M,C", [(12, 50), (30, 64), (50, 1000), (100, 333), (200, 4100), (256, 129), (200, 20001), (100, 17000),
                                 (300, 500), (384, 129), (380, 17000),
                                 # long panels: the persistent output-stationary kernel (csrc/proj64.hip; fp64 results)
                                 (200, 98403), (197, 100000), (100, 98500), (120, 98400)])
def test_whiten_f64_mfma(hip, out_dtype, M, C):
    """alpha = Kinv Kuf and q = diag(Kuf^T alpha) on the fp64 matrix cores vs a CPU fp64 product."""
    f64 = torch.float64
    Kinv, Kuf = rnd(M, M, dtype=f64), rnd(M, C, dtype=f64, seed=1)
    Kinv = Kinv + Kinv.t()
    a, q = hip.whiten(Kinv.to(DEV), Kuf.to(DEV), out_dtype)
    ra, rq = FK.whiten(Kinv, Kuf, f64)
    assert a.dtype == out_dtype and q.dtype == f64
    close(a, ra, 1e-13 if out_dtype == f64 else 1e-6)
    close(q, rq, 1e-13)
    a2, q2 = hip.whiten(Kinv.to(DEV), Kuf.to(DEV), out_dtype, want_q=False)
    assert q2 is None and torch.equal(a2, a)
    # fp32 right-hand side, widened on the fly (the K^-1 solve of the data layer's backward)
    a3, q3 = hip.whiten(Kinv.to(DEV), Kuf.float().to(DEV), out_dtype)
    ra3, rq3 = FK.whiten(Kinv, Kuf.float(), f64)
    close(a3, ra3, 1e-13 if out_dtype == f64 else 1e-6)
    close(q3, rq3, 1e-13)


@pytest.mark.parametrize("kind", ["rbf", "matern12", "matern32"])
@pytest.mark.parametrize("M,C,D", [(200, 100000, 2), (197, 98403, 1), (100, 98500, 3), (120, 100001, 2)])
def test_whiten_gen_forms_the_same_projection(hip, kind, M, C, D):
    """K_uf formed INSIDE the projection kernel (gpsa_whiten_gen_f64_dual, the headline step's data-GP forward) against
    the two launches it replaces (gpsa_kmat fp64 on fp64 points + gpsa_whiten_f64_dual): the same device function in
    the same precision feeds the same matrix instructions - bit-equal alpha (both copies) and q."""
    from spatial_alignment_amd import _lib
    f64, lib = torch.float64, hip.lib
    Z, X = (rnd(M, D) * 3.0).to(DEV), (rnd(C, D, dtype=f64, seed=1) * 3.0).to(DEV)
    ls, var = torch.tensor([0.3], device=DEV), torch.tensor([-0.2], device=DEV)
    Kinv = rnd(M, M, dtype=f64, seed=2)
    Kinv = (Kinv + Kinv.t()).to(DEV)
    st = hip._stream(X)
    wsb = lib.gpsa_whiten_workspace(M)
    kid = {"rbf": 0, "matern12": 1, "matern32": 2}[kind]
    Kuf = torch.empty(M, C, dtype=f64, device=DEV)
    assert lib.gpsa_kmat(1, 2, kid,  # (GPSA_F64, GPSA_F32_X64)
                         Z.data_ptr(), M, X.data_ptr(), C, D, ls.data_ptr(),
                         var.data_ptr(), 0.0, Kuf.data_ptr(), st) == 0
    out = []
    for gen in (False, True):
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        a64, a32 = torch.full((M, C), float("nan"), dtype=f64, device=DEV), torch.full((M, C), float("nan"), device=DEV)
        q = torch.full((C,), float("nan"), dtype=f64, device=DEV)
        if gen:
            rc = lib.gpsa_whiten_gen_f64_dual(Kinv.data_ptr(), kid, Z.data_ptr(), X.data_ptr(), D, ls.data_ptr(),
                                              var.data_ptr(), M, C, a64.data_ptr(), a32.data_ptr(), q.data_ptr(),
                                              ws.data_ptr(), wsb, st)
        else:
            rc = lib.gpsa_whiten_f64_dual(Kinv.data_ptr(), Kuf.data_ptr(), M, C, a64.data_ptr(), a32.data_ptr(),
                                          q.data_ptr(), ws.data_ptr(), wsb, st)
        assert rc == 0
        out.append((a64, a32, q))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    ra, rq = FK.whiten(Kinv.cpu(), FK.kmat(kind, Z.cpu().double(), X.cpu(), ls.cpu().double(), var.cpu().double()), f64)
    close(out[1][0], ra, 1e-9)
    close(out[1][2], rq, 1e-9)
    # short panels are declined (the caller keeps the two launches)
    assert lib.gpsa_whiten_gen_f64_dual(Kinv.data_ptr(), kid, Z.data_ptr(), X.data_ptr(), D, ls.data_ptr(),
                                        var.data_ptr(), M, 20000, out[1][0].data_ptr(), out[1][1].data_ptr(),
                                        out[1][2].data_ptr(), ws.data_ptr(), wsb, st) == _lib.GPSA_EUNSUPPORTED


@pytest.mark.parametrize("M,L,C", [(200, 50, 100000), (197, 7, 4100), (256, 64, 5000), (16, 1, 4096), (200, 50, 12500),
                                   (33, 10, 20004), (208, 49, 4108), (65, 50, 8000)])
def test_thin_update(hip, M, L, C):
    """out += A B with a thin inner dimension in one pass over the long panel (gpsa_thin_update_f32: the mean term's
    share of the data GP's projection gradient) against the fp64 product."""
    A, B, out = rnd(M, L), rnd(L, C, seed=1), rnd(M, C, seed=2)
    want = out.double() + A.double() @ B.double()
    Ad, Bd, od = A.to(DEV), B.to(DEV), out.to(DEV)
    assert hip.lib.gpsa_thin_update_f32(Ad.data_ptr(), M, L, Bd.data_ptr(), C, od.data_ptr(), hip._stream(od)) == 0
    close(od, want, 2e-6)
    # shapes it declines: the caller runs gpsa_gemm with beta = 1
    for m, l, c in ((300, 50, C), (M, 65, C), (M, L, 1000), (M, L, C - 1)):
        assert hip.lib.gpsa_thin_update_f32(Ad.data_ptr(), m, l, Bd.data_ptr(), c, od.data_ptr(), hip._stream(od)) == -3


def test_whiten_unsupported_size_is_reported(hip):
    f64 = torch.float64
    assert hip.whiten(torch.eye(400, dtype=f64, device=DEV), torch.ones(400, 8, dtype=f64, device=DEV), f64) is None


@pytest.mark.parametrize("C,L", [(100, 5), (4097, 50), (33, 33)])
def test_data_sample(hip, C, L):
    meanT, v = rnd(L, C), rnd(L, C, seed=1).abs()
    q, eps = rnd(C, seed=2, dtype=torch.float64).abs() * 0.1, rnd(C, L, seed=3)
    var_u = torch.tensor([0.5])
    F, Sig = hip.data_sample_fwd(meanT.to(DEV), v.to(DEV), q.to(DEV), var_u.to(DEV), eps.to(DEV))
    rF, rS = FK.data_sample_fwd(meanT.double(), v.double(), q.double(), var_u.double(), eps.double())
    close(F, rF, 1e-6)
    close(Sig, rS, 1e-6)
    dF = rnd(C, L, seed=4)
    got = hip.data_sample_bwd(dF.to(DEV), eps.to(DEV), Sig, var_u.to(DEV))
    want = FK.data_sample_bwd(dF.double(), eps.double(), rS, var_u.double())
    assert got[0].shape == (L + 1, C)  # g rows, then the qbar row
    for a, b in zip(got, want):
        close(a, b, 2e-6)


@pytest.mark.parametrize("n,D,S", [(100, 2, 3), (5000, 1, 1), (777, 3, 5)])
def test_warp_sample(hip, n, D, S):
    f64 = torch.float64
    meanT, v = rnd(D, n, dtype=f64), rnd(D, n, dtype=f64, seed=1).abs()
    q, X = rnd(n, dtype=f64, seed=2).abs() * 0.1, rnd(n, D, seed=3, scale=4)
    A, b = torch.eye(D) + 0.1 * rnd(D, D, seed=7), rnd(D, seed=8)
    eps, var_u = rnd(S, n, D, seed=4), torch.tensor([0.2])
    dev = lambda *ts: [t.to(DEV) for t in ts]
    Gm, Gs, bad, Gs64 = hip.warp_sample_fwd(*dev(meanT, v, q, var_u, X, A, b, eps))
    rGm, rGs, rbad, rGs64 = FK.warp_sample_fwd(meanT, v, q, var_u, X, A, b, eps)
    close(Gm, rGm, 1e-6)
    close(Gs, rGs, 1e-6)
    close(Gs64, rGs64, 1e-13)  # the draws before their rounding to the fp32 API tensor
    assert torch.equal(Gs64.float(), Gs)
    assert bad.numel() == (n + 255) // 256 and int(bad.abs().max()) == 0
    dGm, dGs = rnd(n, D, seed=5), rnd(S, n, D, seed=6)
    dGs64 = rnd(S, n, D, seed=9, dtype=f64)
    # the draws' gradient arrives as an fp32 part, an fp64 part, or both
    for a32, a64 in ((dGs, None), (None, dGs64), (dGs, dGs64)):
        got = hip.warp_sample_bwd(dGm.to(DEV), None if a32 is None else a32.to(DEV), *dev(eps, var_u, X),
                                  None if a64 is None else a64.to(DEV))
        want = FK.warp_sample_bwd(dGm, a32, eps, var_u, X, a64)
        for a, w in zip(got, want):
            close(a, w, 1e-11 if a.dtype == f64 else 2e-6)
    _, _, bad, _ = hip.warp_sample_fwd(*dev(meanT, v - 100, q, var_u, X, A, b, eps))
    assert int(bad.max()) == 1


@pytest.mark.parametrize("M,D,scale", [(50, 2, 1.0), (200, 2, 1.0), (333, 3, 100.0), (7, 1, 1.0)])
def test_mean_resid(hip, M, D, scale):
    Z, delta = rnd(M, D, seed=1, scale=5), rnd(M, D, seed=2, scale=5)
    A, b = torch.eye(D) + 0.1 * rnd(D, D, seed=3), rnd(D, seed=4)
    mu, r = hip.mean_resid_fwd(*[t.to(DEV) for t in (Z, A, b, delta)], scale)
    rmu, rr = FK.mean_resid_fwd(Z, A, b, delta, scale)
    assert mu.dtype == torch.float32 and r.dtype == torch.float64
    close(mu, rmu, 1e-6)
    close(r, rr, 1e-13)
    dres = rnd(M, D, dtype=torch.float64, seed=5)
    got = hip.mean_resid_bwd(dres.to(DEV), Z.to(DEV), A.to(DEV), scale)
    want = FK.mean_resid_bwd(dres, Z, A, scale)
    for a, w in zip(got, want):
        close(a, w, 2e-6)
    # identity mean function and delta = Z (the initial state): the residual is exactly zero
    _, r0 = hip.mean_resid_fwd(Z.to(DEV), torch.eye(D, device=DEV), torch.zeros(D, device=DEV), Z.to(DEV))
    assert float(r0.abs().max()) == 0.0


@pytest.mark.parametrize("S,N,P", [(1, 100, 3), (5, 2000, 50), (2, 33, 7)])
def test_loglik(hip, S, N, P):
    F, Y, nu = rnd(S, N, P), rnd(N, P, seed=1), torch.tensor([-0.8])
    close(hip.loglik_fwd(F.to(DEV), Y.to(DEV), nu.to(DEV)), FK.loglik_fwd(F, Y, nu), 1e-6)
    go = torch.tensor([-1.3], dtype=torch.float64)
    dF, dn = hip.loglik_bwd(F.to(DEV), Y.to(DEV), nu.to(DEV), go.to(DEV))
    rdF, rdn = FK.loglik_bwd(F, Y, nu, go)
    close(dF, rdF, 1e-6)
    close(dn, rdn, 1e-5)


def test_bdot_add_diag(hip):
    A, B = rnd(40, 40, dtype=torch.float64), rnd(6, 40, 40, dtype=torch.float64, seed=1)
    close(hip.bdot(A.to(DEV), B.to(DEV)), FK.bdot(A, B), 1e-12)
    X = B.clone().to(DEV)
    hip.add_diag(X, 0.25)
    close(X, FK.add_diag(B.clone(), 0.25), 1e-15)


@pytest.mark.parametrize("M,L,strided", [(12, 3, False), (200, 50, False), (40, 2, True)])
def test_mvn_kl_fwd_bwd(hip, M, L, strided):
    f64 = torch.float64
    A = rnd(M, M, dtype=f64)
    K = A @ A.t() + 0.5 * torch.eye(M, dtype=f64)
    Kinv = torch.linalg.inv(K)
    V = 3 if strided else 1
    B = rnd(L * V, M, M, dtype=f64, seed=1)
    Om_all = B @ B.transpose(1, 2) + 0.1 * torch.eye(M, dtype=f64)
    Oinv_all = torch.linalg.inv(Om_all)
    ld_all = torch.logdet(Om_all)
    Dm, g = rnd(M, L, dtype=f64, seed=2), rnd(L, dtype=f64, seed=3)
    ldK = torch.logdet(K).reshape(1)
    sel = slice(1 if strided else 0, None, V)
    d = lambda t: t.to(DEV)
    Om_d, Oi_d, ld_d = d(Om_all)[sel], d(Oinv_all)[sel], d(ld_all)[sel]
    kl, KD = hip.mvn_kl_fwd(d(Kinv), d(ldK), Om_d, ld_d, d(Dm))
    rkl, rKD = FK.mvn_kl_fwd(Kinv, ldK, Om_all[sel], ld_all[sel], Dm)
    close(kl, rkl, 1e-11)
    close(KD, rKD, 1e-11)
    got = hip.mvn_kl_bwd(d(K), d(Kinv), Om_d, Oi_d, d(Dm), KD, d(g))
    want = FK.mvn_kl_bwd(K, Kinv, Om_all[sel], Oinv_all[sel], Dm, rKD, g)
    for a, b in zip(got, want):
        close(a, b, 1e-10)


def test_kept_products_stay_inside_their_workspace():
    """round 6 (found by tools/fuzz_kernels.py): the panel kernels' staging ring kept requesting chunks behind the
    packed operand's last one; with the workspace at the very end of its device allocation that was a memory fault.
    tools/keep_probe.py puts it there (a child process per shape: a fault would take the process down)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "keep_probe.py"), "--quick"], capture_output=True,
                       text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("M=")]
    assert len(lines) == 3 and all(ln.endswith("rc=0 ok") for ln in lines), (r.stdout[-600:], r.stderr[-300:])
