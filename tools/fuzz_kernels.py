"""Randomised shapes through the C-ABI kernels against the CPU contract (tests/fake_ops.py, fp64): odd column
counts, row counts around the 16-row tile edges, every dtype pairing.  Diagnostic (the fixed-size cases live in
tests/test_hip_kernels.py).   usage: python tools/fuzz_kernels.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from fake_ops import FakeOps  # noqa: E402
from spatial_alignment_amd import ops as ops_mod  # noqa: E402

hip, FK, DEV = ops_mod.get_ops(), FakeOps(), "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


TRACE = os.environ.get("FUZZ_TRACE") == "1"  # print every check as it passes (a device fault then names its neighbourhood)


def note(name, err, tol, ctx):
    if TRACE:
        torch.cuda.synchronize()
        print("ok", name, f"{err:.1e}", ctx, flush=True)
    w = worst.get(name, (0.0, None))
    if err > w[0]:
        worst[name] = (err, ctx)
    assert err <= tol, (name, err, tol, ctx)


def rnd(*shape, dtype=torch.float32, scale=1.0):
    return torch.randn(*shape, dtype=torch.float64, generator=G).mul_(scale).to(dtype)


G = torch.Generator().manual_seed(rng.randrange(1 << 30))
for it in range(cases):
    M = rng.choice([1, 2, 7, 15, 16, 17, 31, 33, 50, 100, 127, 199, 200, 208, 209, 255, 256, 257, 300, 385, 500, 641])
    C = rng.choice([1, 3, 15, 64, 65, 100, 255, 257, 1000, 1249, 1250, 4095, 4096, 4100, 8191, 8200, 16385])
    L = rng.choice([1, 2, 3, 5, 8])
    dt = rng.choice([torch.float32, torch.float64])
    t = 5e-5 if dt == torch.float32 else 1e-10
    ctx = dict(M=M, C=C, L=L, dt=str(dt))
    al = rnd(M, C, dtype=dt)
    A = rnd(L, M, M, dtype=torch.float64, scale=1.0 / max(M, 1) ** 0.5)
    Om = (A @ A.transpose(1, 2)).to(dt)
    g = rnd(L, C, dtype=dt)
    ald, Omd, gd = al.to(DEV), Om.to(DEV), g.to(DEV)
    note("quadform_fwd", rel(hip.quadform_fwd(ald, Omd), FK.quadform_fwd(al.double(), Om.double())), t, ctx)
    note("quadform_bwd_alpha", rel(hip.quadform_bwd_alpha(ald, Omd, gd),
                                   FK.quadform_bwd_alpha(al.double(), Om.double(), g.double())), t, ctx)
    note("quadform_bwd_omega", rel(hip.quadform_bwd_omega(ald, gd), FK.quadform_bwd_omega(al.double(), g.double())), t, ctx)
    if dt == torch.float32:  # the data GP's kept-products pair (both layouts: accumulator order, row-major beyond 256)
        lib, st = hip.lib, hip._stream(ald)
        wsb, nb = lib.gpsa_quadform_keep_f32_workspace(M, L), lib.gpsa_quadform_keep_f32_bytes(M, C, L)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
        vk, Wk = torch.empty(L, C, device=DEV), torch.full((nb // 4,), float("nan"), device=DEV)
        rc_keep = lib.gpsa_quadform_fwd_keep_f32(0, ald.data_ptr(), Omd.data_ptr(), M, C, L, vk.data_ptr(), Wk.data_ptr(),
                                                 ws.data_ptr(), wsb, st)
        if TRACE:
            torch.cuda.synchronize()
            print("   fwd_keep_f32 rc", rc_keep, "wsb", wsb, "keep bytes", nb, ctx, flush=True)
        # (rc != 0: a shape the kept-products pair declines - unaligned panel rows; the step pads those)
        if rc_keep == 0:
            note("keep_f32.v", rel(vk, FK.quadform_fwd(al.double(), Om.double())), t, ctx)
            dc, dm = rnd(M, L), rnd(L, C)
            dcd, dmd = dc.to(DEV), dm.to(DEV)  # (named: a temporary's pointer dangles)
            ok = torch.empty(M, C, device=DEV)
            assert lib.gpsa_quadform_bwd_alpha_kept_f32(Wk.data_ptr(), gd.data_ptr(), M, C, L, dcd.data_ptr(),
                                                        dmd.data_ptr(), ok.data_ptr(), st) == 0
            note("keep_f32.dalpha", rel(ok, FK.quadform_bwd_alpha(al.double(), Om.double(), g.double()) + dc.double() @ dm.double()),
                 t, ctx)
    if L <= 3:
        dcT = rnd(M, L, dtype=dt)
        v, W, mean = hip.quadform_fwd_keep(ald, Omd, dcT.to(DEV))
        note("fwd_keep.v", rel(v, FK.quadform_fwd(al.double(), Om.double())), t, ctx)
        note("fwd_keep.mean", rel(mean, dcT.double().t() @ al.double()), t, ctx)
        note("bwd_alpha_kept", rel(hip.quadform_bwd_alpha_kept(W, gd), FK.quadform_bwd_alpha(al.double(), Om.double(), g.double())), t, ctx)
    if M <= 256 and dt == torch.float32:  # the thin-inner-dimension update (abar += delta dmean^T)
        Lt, Ct = rng.choice([1, 3, 4, 7, 31, 50, 64]), rng.choice([4096, 4100, 5000, 12500, 20004])
        At, Bt, ot = rnd(M, Lt), rnd(Lt, Ct), rnd(M, Ct)
        Atd, Btd, otd = At.to(DEV), Bt.to(DEV), ot.to(DEV)
        assert hip.lib.gpsa_thin_update_f32(Atd.data_ptr(), M, Lt, Btd.data_ptr(), Ct, otd.data_ptr(), hip._stream(otd)) == 0
        note("thin_update", rel(otd, ot.double() + At.double() @ Bt.double()), 2e-6, dict(ctx, Lt=Lt, Ct=Ct))
    if it % 8 == 0 and M <= 208 and M > 96:  # K_uf formed inside the projection kernel (long panels only) vs kmat + whiten
        Cg, Dg = rng.choice([98304, 98403, 100000, 131075]), rng.choice([1, 2, 3])
        kg = rng.choice(["rbf", "matern12", "matern32"])
        Zg, Xg = (rnd(M, Dg, scale=3.0)).to(DEV), rnd(Cg, Dg, dtype=torch.float64, scale=3.0).to(DEV)
        lsg, varg = torch.tensor([0.3], device=DEV), torch.tensor([-0.2], device=DEV)
        Kg = rnd(M, M, dtype=torch.float64)
        Kg = (Kg + Kg.t()).to(DEV)
        wsb = hip.lib.gpsa_whiten_workspace(M)
        wsg = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        a64 = torch.full((M, Cg), float("nan"), dtype=torch.float64, device=DEV)
        a32 = torch.full((M, Cg), float("nan"), device=DEV)
        qg = torch.full((Cg,), float("nan"), dtype=torch.float64, device=DEV)
        rc = hip.lib.gpsa_whiten_gen_f64_dual(Kg.data_ptr(), {"rbf": 0, "matern12": 1, "matern32": 2}[kg], Zg.data_ptr(),
                                              Xg.data_ptr(), Dg, lsg.data_ptr(), varg.data_ptr(), M, Cg, a64.data_ptr(),
                                              a32.data_ptr(), qg.data_ptr(), wsg.data_ptr(), wsb, hip._stream(Xg))
        if rc == 0:  # (-3: the persistent kernel does not take the shape - M outside its two row-tile counts)
            Kuf = hip.kmat(kg, Zg, Xg, lsg, varg, 0.0, dtype=torch.float64)
            ra, rq = hip.whiten(Kg, Kuf, torch.float64)
            gctx = dict(ctx, Cg=Cg, Dg=Dg, kind=kg)
            note("whiten_gen.alpha64", rel(a64, ra), 1e-13, gctx)
            note("whiten_gen.alpha32", rel(a32, ra), 1e-6, gctx)
            note("whiten_gen.q", rel(qg, rq), 1e-13, gctx)
            del Kuf, ra, rq
        else:
            assert rc == -3, rc
        del a64, a32, qg, Xg
    P = rnd(M, M, dtype=torch.float64).tril()
    Y, cs = hip.panel_mm(P.to(DEV), ald, want_colsq=True, transP=rng.random() < 0.5 and False)
    rY, rcs = FK.panel_mm(P, al.double(), want_colsq=True)
    note("panel_mm", rel(Y, rY), t, ctx)
    note("panel_mm.colsq", rel(cs, rcs), t, ctx)
    if M <= 384:
        Kinv = rnd(M, M, dtype=torch.float64)
        Kinv = Kinv + Kinv.t()
        odt = rng.choice([torch.float32, torch.float64])
        r = hip.whiten(Kinv.to(DEV), ald, odt)
        ra, rq = FK.whiten(Kinv, al, torch.float64)
        note("whiten.alpha", rel(r[0], ra), 1e-6 if odt == torch.float32 else 1e-12, ctx)
        note("whiten.q", rel(r[1], rq), 1e-12, ctx)
    B = rng.choice([1, 2, 5])
    S = rnd(B, M, M, dtype=torch.float64)
    K = S @ S.transpose(1, 2) / max(M, 1) + 0.05 * torch.eye(M, dtype=torch.float64)
    Linv, logdet, info = hip.chol_inv(K.to(DEV))
    rL, rld, _ = FK.chol(K)
    assert int(info.abs().max()) == 0, ctx
    note("chol_inv.Linv", rel(Linv, FK.tri_inv(rL)), 1e-8, ctx)
    note("chol_inv.logdet", rel(logdet, rld), 1e-11, ctx)
    D = rng.choice([1, 2, 3])
    kind = rng.choice(["rbf", "matern12", "matern32"])
    Mz = min(M, 64)
    Z, X = rnd(Mz, D, scale=3.0), rnd(min(C, 3000), D, scale=3.0)
    ls, var = torch.tensor([0.3]), torch.tensor([-0.2])
    kdt = rng.choice([torch.float32, torch.float64])
    Kd = hip.kmat(kind, Z.to(DEV), X.to(DEV), ls.to(DEV), var.to(DEV), 0.0, dtype=kdt)
    note("kmat", rel(Kd, FK.kmat(kind, Z.double(), X.double(), ls.double(), var.double())), 2e-6 if kdt == torch.float32 else 1e-6, ctx)
    Kb = rnd(Mz, X.shape[0], dtype=kdt)
    got = hip.kmat_bwd(kind, Z.to(DEV), X.to(DEV), ls.to(DEV), var.to(DEV), Kb.to(DEV))
    want = FK.kmat_bwd(kind, Z.double(), X.double(), ls.double(), var.double(), Kb.double())
    for nm, a, b in zip(("dZ", "dX", "dpar"), got, want):
        note("kmat_bwd." + nm, rel(a, b), 2e-4 if kdt == torch.float32 else 5e-6, dict(ctx, kind=kind, D=D))
torch.cuda.synchronize()
print(f"{cases} random cases passed; worst relative errors:")
for k, (e, c) in sorted(worst.items()):
    print(f"  {k:22s} {e:.2e}  at {c}")
