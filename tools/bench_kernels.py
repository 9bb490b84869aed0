"""Micro-benchmarks of single C-ABI entry points on one GPU (HIP-event timed).  Diagnostic tool.
usage: python tools/bench_kernels.py chol_inv|gram|kmat_bwd|gemm64 ...
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

if os.environ.get("GPSA_AB_LIB"):  # A/B a second build of the library inside one GPU call
    from spatial_alignment_amd import _lib as _lib_mod  # noqa: E402

    _lib_mod.LIB_PATH = os.path.abspath(os.environ["GPSA_AB_LIB"])
    os.environ["GPSA_ALLOW_STALE_LIB"] = "1"  # (an older build on purpose)
ge.build()
from spatial_alignment_amd import ops as ops_mod  # noqa: E402

o = ops_mod.get_ops()
dev = "cuda"


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def spd(B, M, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(B, M, M, generator=g, dtype=torch.float64)
    return (A @ A.transpose(1, 2) / M + 0.05 * torch.eye(M, dtype=torch.float64)).to(dev)


what = sys.argv[1:] or ["chol_inv"]
if "chol_inv" in what:
    for B, M in [(57, 200), (7, 200), (57, 100), (57, 256), (51, 50)]:
        K = spd(B, M)
        print(f"chol_inv B={B} M={M}: {timeit(lambda: o.chol_inv(K)):.1f} us", flush=True)
if "gram" in what:
    for C in (100000, 12500):
        a = torch.randn(200, C, device=dev)
        g = torch.randn(50, C, device=dev)
        print(f"quadform_bwd_omega C={C}: {timeit(lambda: o.quadform_bwd_omega(a, g), n=10, warm=2):.1f} us", flush=True)
if "kmat_bwd" in what:
    for M, C, dt in [(200, 200, torch.float64), (200, 1250, torch.float64), (200, 10000, torch.float64),
                     (200, 12500, torch.float32), (200, 100000, torch.float32)]:
        Z = torch.rand(M, 2, device=dev, dtype=dt) * 10
        X = torch.rand(C, 2, device=dev, dtype=dt) * 10
        ls, var = torch.zeros(1, device=dev, dtype=dt), torch.zeros(1, device=dev, dtype=dt)
        Kb = torch.randn(M, C, device=dev, dtype=dt)
        print(f"kmat_bwd {dt} M={M} C={C}: {timeit(lambda: o.kmat_bwd('rbf', Z, X, ls, var, Kb, need_dX=True)):.1f} us", flush=True)
if "panel" in what:
    for C in (12500, 100000):
        P = torch.randn(200, 200, device=dev, dtype=torch.float64).tril()
        X = torch.randn(200, C, device=dev)
        Om = torch.randn(50, 200, 200, device=dev, dtype=torch.float64)
        g = torch.randn(50, C, device=dev)
        print(f"panel_mm f32 (fp64 P) C={C}: {timeit(lambda: o.panel_mm(P, X), n=20):.1f} us", flush=True)
        print(f"quadform_bwd_alpha C={C}: {timeit(lambda: o.quadform_bwd_alpha(X, Om, g), n=10, warm=2):.1f} us", flush=True)
        Os = Om + Om.transpose(1, 2)
        print(f"quadform_fwd C={C}: {timeit(lambda: o.quadform_fwd(X, Os), n=10, warm=2):.1f} us", flush=True)
if "keep" in what:  # the headline forward: quadratic form with the products kept for the backward
    for C in (100000,):
        X = torch.randn(200, C, device=dev)
        Om = torch.randn(50, 200, 200, device=dev, dtype=torch.float64)
        Os = Om + Om.transpose(1, 2)
        print(f"quadform_fwd_keep C={C}: {timeit(lambda: o.quadform_fwd_keep(X, Os), n=10, warm=2):.1f} us", flush=True)
if "elbo" in what:  # variance + draw + likelihood + abar in one pass over the products (vs keep-forward + kept_wsum)
    for N, S in ((20000, 5), (20000, 1)):
        C = N * S
        X = torch.randn(200, C, device=dev) / 14
        A = torch.randn(50, 200, 200, device=dev, dtype=torch.float64) / 14
        Om = A @ A.transpose(1, 2)
        meanT = torch.randn(50, C, device=dev)
        q = torch.rand(C, device=dev, dtype=torch.float64) * 0.2
        vu, nu = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        eps, Y = torch.randn(S, N, 50, device=dev), torch.randn(N, 50, device=dev)
        print(f"quadform_elbo C={C}: {timeit(lambda: o.quadform_elbo(X, Om, meanT, q, vu, eps, Y, nu), n=10, warm=2):.1f} us", flush=True)
if "solve" in what:  # gamma = K^-1 abar of the data-layer backward: one fp64-MFMA pass vs two fp32 triangular passes
    for C in (12500, 100000):
        Kinv = spd(1, 200)[0]
        Linv = torch.linalg.inv(torch.linalg.cholesky(Kinv.cpu())).to(dev)
        X = torch.randn(200, C, device=dev)
        print(f"whiten fp32 in/out C={C}: {timeit(lambda: o.whiten(Kinv, X, torch.float32, want_q=False), n=20):.1f} us", flush=True)
        print(f"2 x panel_mm C={C}: {timeit(lambda: o.panel_mm(Linv, o.panel_mm(Linv, X)[0], transP=True), n=20):.1f} us", flush=True)
        X64 = X.double()
        print(f"whiten fp64 in, fp32 out C={C}: {timeit(lambda: o.whiten(Kinv, X64, torch.float32), n=20):.1f} us", flush=True)
        Xw = X64[:, : C // 10].contiguous()  # one view's warp-layer panel
        print(f"whiten fp64 in/out C={C // 10}: {timeit(lambda: o.whiten(Kinv, Xw, torch.float64), n=20):.1f} us", flush=True)
if "gemm32" in what:  # the data layer's fp32 products (M = 200, L = 50, C columns)
    for C in (12500, 100000):
        al = torch.randn(200, C, device=dev)
        W = torch.randn(200, C, device=dev)
        dm = torch.randn(50, C, device=dev)
        dc = torch.randn(200, 50, device=dev)
        out = torch.randn(200, C, device=dev)
        sk = o.pick_splitk(C, 200, 200)
        print(f"f32 dKuu NT 200x{C}x200 splitk={sk}: {timeit(lambda: o.gemm(W, al, transB=True, alpha=-1.0, splitk=sk), n=20):.1f} us", flush=True)
        sk2 = o.pick_splitk(C, 200, 50)
        print(f"f32 ddc NT 200x{C}x50 splitk={sk2}: {timeit(lambda: o.gemm(al, dm, transB=True, splitk=sk2), n=20):.1f} us", flush=True)
        print(f"f32 meanT TN 50x200x{C}: {timeit(lambda: o.gemm(dc, al, transA=True), n=20):.1f} us", flush=True)
        print(f"f32 abar += dc dmean NN 200x50x{C}: {timeit(lambda: o.gemm(dc, dm, beta=1.0, out=out), n=20):.1f} us", flush=True)
if "gemm64w" in what:  # the warp-layer shapes (M = 200, C = columns of one view)
    for C in (1250, 10000):
        A = torch.randn(200, 200, device=dev, dtype=torch.float64)
        X = torch.randn(200, C, device=dev, dtype=torch.float64)
        print(f"gemm f64 NN 200x200x{C}: {timeit(lambda: o.gemm(A, X)):.1f} us", flush=True)
        sk = o.pick_splitk(C, 200, 200)
        print(f"gemm f64 NT 200x{C}x200 splitk={sk}: {timeit(lambda: o.gemm(X, X, transB=True, splitk=sk)):.1f} us", flush=True)
if "splitk" in what:
    for dt in (torch.float64, torch.float32):
        for K in (10000, 100000, 1250):
            X = torch.randn(200, K, device=dev, dtype=dt)
            for sk in (4, 8, 16, 32, 64, 128, 256):
                if K // sk < 64:
                    continue
                print(f"{dt} NT 200x{K}x200 splitk={sk}: {timeit(lambda: o.gemm(X, X, transB=True, splitk=sk), n=20):.1f} us", flush=True)
if "gemm64" in what:
    for B in (4, 50, 57):
        A = torch.randn(B, 200, 200, device=dev, dtype=torch.float64)
        print(f"gemm f64 NT batch={B} 200^3: {timeit(lambda: o.gemm(A, A, transB=True)):.1f} us", flush=True)
        print(f"gemm f64 TN batch={B} 200^3: {timeit(lambda: o.gemm(A, A, transA=True)):.1f} us", flush=True)
if "bigm" in what:  # M beyond 256: MFMA panel variants (24 / 32 row tiles) vs the generic tiled path (GPSA_FORCE_GENERIC=1)
    for M in (380, 500):
        C, L = 50000, 8
        X = torch.randn(M, C, device=dev)
        A = torch.randn(L, M, M, device=dev, dtype=torch.float64) / M ** 0.5
        Om = A @ A.transpose(1, 2)
        g = torch.randn(L, C, device=dev)
        print(f"M={M} quadform_fwd: {timeit(lambda: o.quadform_fwd(X, Om), n=5, warm=2):.0f} us", flush=True)
        print(f"M={M} quadform_bwd_alpha: {timeit(lambda: o.quadform_bwd_alpha(X, Om, g), n=5, warm=2):.0f} us", flush=True)
        P = torch.randn(M, M, device=dev, dtype=torch.float64).tril()
        print(f"M={M} panel_mm: {timeit(lambda: o.panel_mm(P, X), n=5, warm=2):.0f} us", flush=True)
        Kinv = spd(1, M)[0]
        X64 = X[:, :20000].double().contiguous()
        r = o.whiten(Kinv, X64, torch.float64)
        print(f"M={M} whiten f64 C=20000: " + ("unsupported" if r is None else f"{timeit(lambda: o.whiten(Kinv, X64, torch.float64), n=5, warm=2):.0f} us"), flush=True)
if "lib64" in what:  # the library's batched fp64 GEMM (rocBLAS / hipBLASLt through torch) on the M x M x M shapes
    for B in (4, 50, 57):
        A = torch.randn(B, 200, 200, device=dev, dtype=torch.float64)
        print(f"ours  f64 NT batch={B} 200^3: {timeit(lambda: o.gemm(A, A, transB=True)):.1f} us", flush=True)
        print(f"torch f64 NT batch={B} 200^3: {timeit(lambda: torch.bmm(A, A.transpose(1, 2))):.1f} us", flush=True)
        print(f"ours  f64 TN batch={B} 200^3: {timeit(lambda: o.gemm(A, A, transA=True)):.1f} us", flush=True)
        print(f"torch f64 TN batch={B} 200^3: {timeit(lambda: torch.bmm(A.transpose(1, 2), A)):.1f} us", flush=True)
    for C in (1250, 10000):
        X = torch.randn(200, C, device=dev, dtype=torch.float64)
        K = torch.randn(200, 200, device=dev, dtype=torch.float64)
        sk = o.pick_splitk(C, 200, 200)
        print(f"ours  f64 NT 200x{C}x200 splitk={sk}: {timeit(lambda: o.gemm(X, X, transB=True, splitk=sk)):.1f} us", flush=True)
        print(f"torch f64 NT 200x{C}x200: {timeit(lambda: X @ X.t()):.1f} us", flush=True)
        print(f"ours  f64 NN 200x200x{C}: {timeit(lambda: o.gemm(K, X)):.1f} us", flush=True)
        print(f"torch f64 NN 200x200x{C}: {timeit(lambda: K @ X):.1f} us", flush=True)
if "gramw" in what:  # the warp layer's Gram shape: L = D = 2 outputs, one view's columns
    for C in (10000, 20000, 4096):
        a = torch.randn(200, C, device=dev)
        g = torch.randn(2, C, device=dev)
        print(f"quadform_bwd_omega L=2 C={C}: {timeit(lambda: o.quadform_bwd_omega(a, g, out_dtype=torch.float64), n=20, warm=3):.1f} us", flush=True)

if "big" in what:  # the large-M contraction kernels (BASELINE configs 4 / 5 shapes, fewer outputs)
    shapes = [(1000, 25600, 64), (500, 40320, 96)]
    if "big5" in what:  # alpha (400 MB) beyond the Infinity Cache, as at BASELINE config 5
        shapes = [(1000, 102400, 48)]
    for M, C, L in shapes:
        a = torch.randn(M, C, device=dev)
        A = torch.randn(L, M, M, device=dev, dtype=torch.float64) / M ** 0.5
        Om = A @ A.transpose(1, 2)
        g = torch.randn(L, C, device=dev)
        fl = 2.0 * C * L * M * M
        for name, fn in (("quadform_fwd", lambda: o.quadform_fwd(a, Om)), ("quadform_bwd_alpha", lambda: o.quadform_bwd_alpha(a, Om, g)),
                         ("quadform_bwd_omega", lambda: o.quadform_bwd_omega(a, g))):
            us = timeit(fn, n=5, warm=2)
            print(f"{name} M={M} C={C} L={L}: {us / 1e3:.2f} ms  nominal {fl / us / 1e6:.1f} TF", flush=True)
