"""The batched entry points of the C ABI (the warp GPs of several views in one launch) against loops over
their single-problem forms, which tests/test_hip_kernels.py pins to the fp64 contracts."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
f64, f32 = torch.float64, torch.float32
_raw_stream = torch._C._cuda_getCurrentRawStream


@pytest.fixture(scope="module")
def hip():
    from spatial_alignment_amd.ops import get_ops

    return get_ops()


def rnd(*shape, dtype=f32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=f64) * scale).to(dtype)


def stream():
    return _raw_stream(0)


def p(t):
    return 0 if t is None else t.data_ptr()


KIND = {"rbf": 0, "matern12": 1, "matern32": 2}


@pytest.mark.parametrize("kind,M,D,B,Cs,nl", [("rbf", 20, 2, 3, 128, [100, 128, 0]), ("matern32", 200, 3, 2, 320, [300, 257]),
                                             ("matern12", 7, 1, 5, 64, [64, 1, 33, 64, 10])])
def test_kmat_batched_fwd_bwd(hip, kind, M, D, B, Cs, nl):
    Z = rnd(B, M, D, seed=1, scale=3).to(DEV)
    X = rnd(B, Cs, D, seed=2, scale=3).to(DEV)
    ls, var = rnd(B, seed=3, scale=0.3).to(DEV), rnd(B, seed=4, scale=0.3).to(DEV)
    nlive = (C.c_longlong * B)(*nl)
    K = torch.full((B, M, Cs), 7.0, dtype=f64, device=DEV)
    rc = hip.lib.gpsa_kmat_batched(KIND[kind], p(Z), M * D, M, p(X), Cs * D, Cs, D, p(ls), p(var), 1, nlive, B, 0.0,
                                   p(K), M * Cs, stream())
    assert rc == 0
    for b in range(B):
        want = hip.kmat(kind, Z[b], X[b], ls[b:b + 1], var[b:b + 1], 0.0, dtype=f64)
        assert torch.equal(K[b][:, : nl[b]], want[:, : nl[b]])
        assert float(K[b][:, nl[b]:].abs().max()) == 0.0 if nl[b] < Cs else True
    # backward: panels with garbage in the padding columns (must be ignored)
    Kbar = rnd(B, M, Cs, dtype=f64, seed=5).to(DEV)
    dZ = torch.empty(B, M, D, dtype=f64, device=DEV)
    dpar = torch.empty(B, 2, dtype=f64, device=DEV)
    wsb = int(hip.lib.gpsa_kmat_bwd_batched_workspace(M, Cs, D, B))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.lib.gpsa_kmat_bwd_batched(KIND[kind], p(Z), M * D, M, p(X), Cs * D, Cs, D, p(ls), p(var), 1, nlive, B,
                                       p(Kbar), M * Cs, 0, p(dZ), M * D, p(dpar), p(ws), wsb, stream())
    assert rc == 0
    for b in range(B):
        if nl[b] == 0:
            assert float(dZ[b].abs().max()) == 0.0 and float(dpar[b].abs().max()) == 0.0
            continue
        wZ, _, wp = hip.kmat_bwd(kind, Z[b], X[b][: nl[b]].contiguous(), ls[b:b + 1], var[b:b + 1],
                                 Kbar[b][:, : nl[b]].contiguous(), need_dX=False, out_dtype=f64)
        assert (dZ[b] - wZ).norm() <= 1e-12 * wZ.norm() and (dpar[b] - wp).norm() <= 1e-12 * wp.norm()


@pytest.mark.parametrize("kind,M,D,B", [("rbf", 50, 2, 3), ("matern32", 200, 2, 2)])
def test_kmat_batched_same_points(hip, kind, M, D, B):
    """K_uu of several views: X = Z, jitter on the diagonal, backward folds the X-side sums into dZ"""
    Z = rnd(B, M, D, seed=1, scale=3).to(DEV)
    ls, var = rnd(B, seed=3, scale=0.3).to(DEV), rnd(B, seed=4, scale=0.3).to(DEV)
    K = torch.empty(B, M, M, dtype=f64, device=DEV)
    rc = hip.lib.gpsa_kmat_batched(KIND[kind], p(Z), M * D, M, p(Z), M * D, M, D, p(ls), p(var), 1, None, B, 1e-5, p(K),
                                   M * M, stream())
    assert rc == 0
    Kbar = rnd(B, M, M, dtype=f64, seed=5).to(DEV)
    dZ = torch.empty(B, M, D, dtype=f64, device=DEV)
    dpar = torch.empty(B, 2, dtype=f64, device=DEV)
    wsb = int(hip.lib.gpsa_kmat_bwd_batched_workspace(M, M, D, B))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.lib.gpsa_kmat_bwd_batched(KIND[kind], p(Z), M * D, M, p(Z), M * D, M, D, p(ls), p(var), 1, None, B, p(Kbar),
                                       M * M, 1, p(dZ), M * D, p(dpar), p(ws), wsb, stream())
    assert rc == 0
    for b in range(B):
        assert torch.equal(K[b], hip.kmat(kind, Z[b], Z[b], ls[b:b + 1], var[b:b + 1], 1e-5, dtype=f64))
        wZ, _, wp = hip.kmat_bwd(kind, Z[b], Z[b], ls[b:b + 1], var[b:b + 1], Kbar[b], same=True, out_dtype=f64)
        assert (dZ[b] - wZ).norm() <= 1e-12 * wZ.norm() and (dpar[b] - wp).norm() <= 1e-12 * wp.norm()


# (200, 2, 20000): several column blocks per workgroup of the register-accumulating kernel; (50, 2, 3000): a last row
# chunk that is not full
@pytest.mark.parametrize("M,D,C", [(200, 2, 5000), (30, 3, 777), (200, 2, 20000), (50, 2, 3000)])
def test_kmat_bwd_x64(hip, M, D, C):
    """the data GP's covariance backward: fp32 inducing points / panel, fp64 coordinates, fp64 results"""
    Z, X64 = rnd(M, D, seed=1, scale=3).to(DEV), rnd(C, D, dtype=f64, seed=2, scale=3).to(DEV)
    ls, var = rnd(1, seed=3, scale=0.3).to(DEV), rnd(1, seed=4, scale=0.3).to(DEV)
    Kbar = rnd(M, C, seed=5).to(DEV)
    dZ, dX = torch.empty(M, D, dtype=f64, device=DEV), torch.empty(C, D, dtype=f64, device=DEV)
    dpar = torch.empty(2, dtype=f64, device=DEV)
    wsb = int(hip.lib.gpsa_kmat_bwd_workspace(1, M, C, D))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.lib.gpsa_kmat_bwd_x64(0, p(Z), M, p(X64), C, D, p(ls), p(var), p(Kbar), p(dZ), p(dX), p(dpar), p(ws), wsb,
                                   stream())
    assert rc == 0
    wZ, wX, wp = hip.kmat_bwd("rbf", Z.double(), X64, ls.double(), var.double(), Kbar.double())
    for a, w in ((dZ, wZ), (dX, wX), (dpar, wp)):
        assert (a - w).norm() <= 1e-12 * w.norm()
    # the gradient panel in two pieces, Kbar + s * d o X2, formed as it is read (in fp64)
    X2, d = rnd(M, C, seed=6).to(DEV), rnd(C, seed=7).to(DEV)
    rc = hip.lib.gpsa_kmat_bwd_x64_axpy(0, p(Z), M, p(X64), C, D, p(ls), p(var), p(Kbar), p(X2), p(d), 2.0, p(dZ), p(dX),
                                        p(dpar), p(ws), wsb, stream())
    assert rc == 0
    full = Kbar.double() + 2.0 * d.double()[None, :] * X2.double()
    wZ, wX, wp = hip.kmat_bwd("rbf", Z.double(), X64, ls.double(), var.double(), full)
    for a, w in ((dZ, wZ), (dX, wX), (dpar, wp)):
        assert (a - w).norm() <= 1e-12 * w.norm()


@pytest.mark.parametrize("M,D,C", [(200, 2, 5000), (30, 3, 777), (200, 2, 20000)])
def test_kmat_bwd_x64_f64_panel_in_two_pieces(hip, M, D, C):
    """the exact mode's covariance backward: an fp64 panel Kbar + s * d o X2 (X2 fp64, d fp32) formed as it is read"""
    Z, X64 = rnd(M, D, seed=1, scale=3).to(DEV), rnd(C, D, dtype=f64, seed=2, scale=3).to(DEV)
    ls, var = rnd(1, seed=3, scale=0.3).to(DEV), rnd(1, seed=4, scale=0.3).to(DEV)
    Kbar, X2, d = rnd(M, C, dtype=f64, seed=5).to(DEV), rnd(M, C, dtype=f64, seed=6).to(DEV), rnd(C, seed=7).to(DEV)
    dZ, dX = torch.empty(M, D, dtype=f64, device=DEV), torch.empty(C, D, dtype=f64, device=DEV)
    dpar = torch.empty(2, dtype=f64, device=DEV)
    wsb = int(hip.lib.gpsa_kmat_bwd_workspace(1, M, C, D))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.lib.gpsa_kmat_bwd_x64_f64_axpy(0, p(Z), M, p(X64), C, D, p(ls), p(var), p(Kbar), p(X2), p(d), 2.0, p(dZ),
                                            p(dX), p(dpar), p(ws), wsb, stream())
    assert rc == 0
    full = Kbar + 2.0 * d.double()[None, :] * X2
    wZ, wX, wp = hip.kmat_bwd("rbf", Z.double(), X64, ls.double(), var.double(), full)
    for a, w in ((dZ, wZ), (dX, wX), (dpar, wp)):
        assert (a - w).norm() <= 1e-12 * w.norm()
    rc = hip.lib.gpsa_kmat_bwd_x64_f64(0, p(Z), M, p(X64), C, D, p(ls), p(var), p(full), p(dZ), p(dX), p(dpar), p(ws),
                                       wsb, stream())
    assert rc == 0
    for a, w in ((dZ, wZ), (dX, wX), (dpar, wp)):
        assert (a - w).norm() <= 1e-12 * w.norm()


@pytest.mark.parametrize("M,C", [(200, 100000), (200, 20000), (25, 1000), (300, 777), (500, 40328)])
def test_exact_dkuu(hip, M, C):
    """dK += -(G + d o A) A^T: the exact inducing-point gradient's C-long fp64 product, left operand formed as staged"""
    G, A, d = rnd(M, C, dtype=f64, seed=1).to(DEV), rnd(M, C, dtype=f64, seed=2).to(DEV), rnd(C, seed=3).to(DEV)
    dK0 = rnd(M, M, dtype=f64, seed=4).to(DEV)
    dK = dK0.clone()
    wsb = int(hip.lib.gpsa_exact_dkuu_workspace(M, C))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=DEV)
    assert hip.lib.gpsa_exact_dkuu_f64(p(G), p(A), p(d), M, C, p(dK), p(ws), wsb, stream()) == 0
    want = dK0 - (G + d.double()[None, :] * A) @ A.t()
    assert (dK - want).norm() <= 1e-12 * want.norm()
    dK2 = dK0.clone()  # bitwise repeatable (fixed-order split-K)
    assert hip.lib.gpsa_exact_dkuu_f64(p(G), p(A), p(d), M, C, p(dK2), p(ws), wsb, stream()) == 0
    assert torch.equal(dK, dK2)


def _longk(hip, G, Bm, d, sym, alpha, beta, out):
    """gpsa_longk_f64 on lists of per-product tensors (G / d may be None)"""
    n = len(Bm)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    M, K = Bm[0].shape
    wsb = int(hip.lib.gpsa_longk_f64_workspace(M, K, n))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=DEV)
    dt = 0 if (d is None or d[0].dtype == f32) else 1
    al, be = (C.c_double * n)(*alpha), (C.c_double * n)(*beta)
    rc = hip.lib.gpsa_longk_f64(n, arr(G) if G is not None else None, arr(Bm), arr(d) if d is not None else None, dt,
                                M, K, K, int(sym), al, be, arr(out), p(ws), wsb, stream())
    return rc, wsb


@pytest.mark.parametrize("M,K,n", [(200, 100000, 1), (200, 10048, 2), (25, 1000, 3), (100, 4098, 2), (256, 2048, 1),
                                   (200, 1282, 6), (50, 20000, 4)])
@pytest.mark.parametrize("mode", ["G+dB/f32", "G+dB/f64", "dB/sym", "G"])
def test_longk_f64(hip, M, K, n, mode):
    """out = beta out + alpha (G + d o B) B^T, nprob long-K fp64 products in one launch, left operand formed in registers;
    K tails that are not a multiple of the 8-column group, every operand combination the step uses"""
    Bm = [rnd(M, K, dtype=f64, seed=10 + i).to(DEV) for i in range(n)]
    G = None if mode == "dB/sym" else [rnd(M, K, dtype=f64, seed=20 + i).to(DEV) for i in range(n)]
    dty = f32 if mode == "G+dB/f32" else f64
    d = None if mode == "G" else [rnd(K, dtype=dty, seed=30 + i).to(DEV) for i in range(n)]
    out0 = [rnd(M, M, dtype=f64, seed=40 + i).to(DEV) for i in range(n)]
    out = [o.clone() for o in out0]
    alpha, beta = [(-1.0) ** i * (1.0 + 0.5 * i) for i in range(n)], [float(i % 2) for i in range(n)]
    rc, wsb = _longk(hip, G, Bm, d, mode == "dB/sym", alpha, beta, out)
    if wsb == 0:
        assert rc != 0  # not covered: the caller keeps its generic path
        pytest.skip("shape not covered by the long-K kernel")
    assert rc == 0
    for i in range(n):
        A = (G[i] if G is not None else 0.0) + (d[i].double()[None, :] * Bm[i] if d is not None else 0.0)
        want = beta[i] * out0[i] + alpha[i] * (A @ Bm[i].t())
        assert (out[i] - want).norm() <= 1e-12 * want.norm(), (i, float((out[i] - want).norm() / want.norm()))
    out2 = [o.clone() for o in out0]  # bitwise repeatable (fixed-order reduction)
    assert _longk(hip, G, Bm, d, mode == "dB/sym", alpha, beta, out2)[0] == 0
    assert all(torch.equal(a, b) for a, b in zip(out, out2))


@pytest.mark.parametrize("M,Cs,B", [(200, 640, 3), (50, 64, 5), (300, 128, 2), (200, 49201, 2), (100, 33000, 3)])
def test_whiten_batched(hip, M, Cs, B):
    A = rnd(B, M, M, dtype=f64, seed=1).to(DEV)
    Kinv = (A @ A.transpose(1, 2) / M + torch.eye(M, dtype=f64, device=DEV)).contiguous()
    Kuf = rnd(B, M, Cs, dtype=f64, seed=2).to(DEV)
    alpha = torch.empty_like(Kuf)
    q = torch.empty(B, Cs, dtype=f64, device=DEV)
    wsb = int(hip.lib.gpsa_whiten_workspace(M)) * B
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.lib.gpsa_whiten_batched_f64(p(Kinv), M * M, p(Kuf), M, Cs, M * Cs, p(alpha), p(q), B, p(ws), wsb, stream())
    assert rc == 0
    for b in range(B):
        wa, wq = hip.whiten(Kinv[b], Kuf[b], f64)
        assert (alpha[b] - wa).norm() <= 1e-13 * wa.norm() and (q[b] - wq).norm() <= 1e-13 * wq.norm()
        ref = Kinv[b] @ Kuf[b]
        assert (alpha[b] - ref).norm() <= 1e-12 * ref.norm()


@pytest.mark.parametrize("M,C", [(200, 3000), (200, 98401), (100, 98400)])
def test_whiten_dual_store(hip, M, C):
    """gpsa_whiten_f64_dual: the projection kept twice from the same accumulators (fp64 and its fp32 rounding); the long
    panels run the persistent kernel (csrc/proj64.hip), whose column tiles may be cut between two workgroups (q then
    closes by two atomic adds onto a zeroed word: bitwise repeatable)."""
    A = rnd(M, M, dtype=f64, seed=1).to(DEV)
    Kinv = (A @ A.t() / M + torch.eye(M, dtype=f64, device=DEV)).contiguous()
    Kuf = rnd(M, C, dtype=f64, seed=2).to(DEV)
    a64, a32 = torch.empty_like(Kuf), torch.empty(M, C, dtype=f32, device=DEV)
    q = torch.full((C,), 7.0, dtype=f64, device=DEV)
    wsb = int(hip.lib.gpsa_whiten_workspace(M))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    assert hip.lib.gpsa_whiten_f64_dual(p(Kinv), p(Kuf), M, C, p(a64), p(a32), p(q), p(ws), wsb, stream()) == 0
    ref = Kinv @ Kuf
    assert (a64 - ref).norm() <= 1e-13 * ref.norm()
    assert torch.equal(a32, a64.float())
    rq = (Kuf * ref).sum(0)
    assert (q - rq).norm() <= 1e-13 * rq.norm()
    b64, b32, q2 = torch.empty_like(a64), torch.empty_like(a32), torch.empty_like(q)
    assert hip.lib.gpsa_whiten_f64_dual(None, p(Kuf), M, C, p(b64), p(b32), p(q2), p(ws), wsb, stream()) == 0  # packed already
    assert torch.equal(b64, a64) and torch.equal(b32, a32) and torch.equal(q2, q)


@pytest.mark.parametrize("M,C", [(200, 5000), (50, 333), (300, 1000)])
def test_whiten_with_fused_column_update(hip, M, C):
    """gpsa_whiten_axpy_f32 = gpsa_whiten_f64 on an fp32 panel followed by gpsa_col_axpy, in one pass"""
    A = rnd(M, M, dtype=f64, seed=1).to(DEV)
    Kinv = (A @ A.t() / M + torch.eye(M, dtype=f64, device=DEV)).contiguous()
    X, X2, d = rnd(M, C, seed=2).to(DEV), rnd(M, C, seed=3).to(DEV), rnd(C, seed=4).to(DEV)
    out = torch.empty(M, C, dtype=f32, device=DEV)
    wsb = int(hip.lib.gpsa_whiten_workspace(M))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    assert hip.lib.gpsa_whiten_axpy_f32(p(Kinv), p(X), M, C, p(X2), p(d), 2.0, p(out), p(ws), wsb, stream()) == 0
    gamma, _ = hip.whiten(Kinv, X, f32, want_q=False)
    want = (Kinv @ X.double() + 2.0 * d.double().unsqueeze(0) * X2.double())
    assert (out.double() - want).norm() <= 2e-7 * want.norm()
    two_pass = hip.col_axpy(gamma, X2, d, 2.0)
    assert (out - two_pass).abs().max() <= 4e-7 * float(want.abs().max())  # one rounding instead of two


@pytest.mark.parametrize("M,Cs,L,B", [(200, 320, 2, 3), (40, 64, 3, 2)])
def test_keep_forms_batched(hip, M, Cs, L, B):
    alpha = rnd(B, M, Cs, dtype=f64, seed=1).to(DEV)
    Om = rnd(B, L, M, M, dtype=f64, seed=2).to(DEV)
    Om = (Om + Om.transpose(-1, -2)).contiguous()
    dcT = rnd(B, M, L, dtype=f64, seed=3).to(DEV)
    v = torch.empty(B, L, Cs, dtype=f64, device=DEV)
    W = torch.empty(B, L, M, Cs, dtype=f64, device=DEV)
    meanT = torch.empty(B, L, Cs, dtype=f64, device=DEV)
    assert hip.lib.gpsa_quadform_fwd_keep_batched_f64(p(alpha), p(Om), M, Cs, L, p(v), p(W), p(dcT), p(meanT), B,
                                                      stream()) == 0
    g, dm = rnd(B, L, Cs, dtype=f64, seed=4).to(DEV), rnd(B, L, Cs, dtype=f64, seed=5).to(DEV)
    dal = torch.empty(B, M, Cs, dtype=f64, device=DEV)
    assert hip.lib.gpsa_quadform_bwd_alpha_kept_batched_f64(p(W), p(g), M, Cs, L, p(dcT), p(dm), p(dal), B, stream()) == 0
    d = rnd(B, Cs, dtype=f64, seed=6).to(DEV)
    ax = torch.empty(B, M, Cs, dtype=f64, device=DEV)
    assert hip.lib.gpsa_col_axpy_batched_f64(p(dal), p(alpha), p(d), 0.7, M, Cs, p(ax), B, stream()) == 0
    dOm = torch.empty(B, L, M, M, dtype=f64, device=DEV)
    wsb = int(hip.lib.gpsa_gram_batched_workspace(M, Cs, L, B))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    assert hip.lib.gpsa_gram_batched_f64(p(alpha), p(g), M, Cs, L, p(dOm), B, p(ws), wsb, stream()) == 0
    close = lambda a, w, tol=1e-12: (a - w).norm() <= tol * w.norm()
    for b in range(B):
        wv, wW, wm = hip.quadform_fwd_keep(alpha[b], Om[b], dcT[b])
        assert close(v[b], wv) and close(W[b], wW) and close(meanT[b], wm)
        assert close(dal[b], hip.quadform_bwd_alpha_kept(W[b], g[b], dcT[b], dm[b]))
        assert close(ax[b], dal[b] + 0.7 * d[b].unsqueeze(0) * alpha[b])
        want = torch.einsum("lc,mc,kc->lmk", g[b], alpha[b], alpha[b])
        assert close(dOm[b], want, 1e-11)


def test_fused_adam_matches_torch():
    """gpsa_adam_step against torch.optim.Adam (the optimiser of the reference loop) over several steps"""
    from spatial_alignment_amd.optim import FusedAdam

    gen = torch.Generator().manual_seed(0)
    shapes = [(3, 200, 200), (17,), (1,), (200, 2), (5, 40, 40)]
    p0 = [torch.randn(*s, generator=gen) for s in shapes]
    ref = [torch.nn.Parameter(t.clone()) for t in p0]
    got = [torch.nn.Parameter(t.clone().to(DEV)) for t in p0]
    o_ref = torch.optim.Adam(ref, lr=1e-2)
    o_got = FusedAdam(got, lr=1e-2)
    for it in range(7):
        gs = [torch.randn(*s, generator=gen) * (10.0 ** (it % 3 - 1)) for s in shapes]
        for a, b, g in zip(ref, got, gs):
            a.grad, b.grad = g.clone(), g.clone().to(DEV)
        o_ref.step()
        o_got.step()
    for a, b in zip(ref, got):
        assert (b.detach().cpu() - a.detach()).abs().max() <= 2e-6 * max(1.0, float(a.detach().abs().max()))


def test_fused_adam_state_dict_and_idle_parameters():
    """The step count lives in optimizer.state (torch.optim.Adam(capturable=True)'s layout): a resumed optimiser
    continues the bias correction where it stopped, the state interchanges with torch's Adam, and a parameter
    without a gradient in a step does not advance (torch's behaviour)."""
    from spatial_alignment_amd.optim import FusedAdam

    gen = torch.Generator().manual_seed(3)
    shapes = [(40, 40), (7,), (3, 5)]
    p0 = [torch.randn(*s, generator=gen) for s in shapes]
    grads = [[torch.randn(*s, generator=gen) for s in shapes] for _ in range(6)]
    idle = {(2, 1), (3, 1), (4, 2)}  # (step, parameter) pairs without a gradient

    def run(make_opt, resume_at=None, swap=None):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in p0]
        opt = make_opt(ps)
        for it, gs in enumerate(grads):
            if resume_at == it:  # save, rebuild (possibly as the other implementation), load
                sd = opt.state_dict()
                opt = (swap or make_opt)(ps)
                opt.load_state_dict(sd)
            for k, (p, g) in enumerate(zip(ps, gs)):
                p.grad = None if (it, k) in idle else g.clone().to(DEV)
            opt.step()
        return [p.detach().cpu() for p in ps], opt

    fused = lambda ps: FusedAdam(ps, lr=1e-2)
    torch_adam = lambda ps: torch.optim.Adam(ps, lr=1e-2, capturable=True)
    want, _ = run(torch_adam)
    for name, got in (("plain", run(fused)[0]), ("resumed", run(fused, resume_at=3)[0]),
                      ("torch -> fused", run(torch_adam, resume_at=3, swap=fused)[0]),
                      ("fused -> torch", run(fused, resume_at=3, swap=torch_adam)[0])):
        for a, b in zip(want, got):
            assert (a - b).abs().max() <= 2e-6 * max(1.0, float(a.abs().max())), name
    _, opt = run(fused)
    steps = [float(opt.state[p]["step"]) for p in opt.param_groups[0]["params"]]
    assert steps == [6.0, 4.0, 5.0]
