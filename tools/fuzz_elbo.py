"""Random shapes through gpsa_quadform_elbo_f32 against the formulas in fp64 (GPU box):  python tools/fuzz_elbo.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatial_alignment_amd.ops import HipOps  # noqa: E402

o = HipOps()
dev = "cuda:0"
gen = torch.Generator().manual_seed(int(os.environ.get("SEED", "0")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for it in range(n):
    M = int(torch.randint(1, 209, (1,), generator=gen))
    N = int(torch.randint(1, 3000, (1,), generator=gen))
    S = int(torch.randint(1, 4, (1,), generator=gen))
    L = int(torch.randint(1, 70, (1,), generator=gen))
    C = S * N
    rnd = lambda *sh: torch.randn(*sh, generator=gen, dtype=torch.float64)
    al = (rnd(M, C) / M ** 0.5).float()
    A = rnd(L, M, M) / M ** 0.5
    Om = A @ A.transpose(1, 2) + 1e-5 * torch.eye(M, dtype=torch.float64)
    meanT = rnd(L, C).float()
    var_u, noise_u = torch.tensor([0.3]), torch.tensor([-0.7])
    q = (rnd(C).abs() * 0.2).clamp(max=1.0)
    eps, Y = rnd(S, N, L).float(), rnd(N, L).float()
    g, dm, abar, z2 = o.quadform_elbo(al.to(dev), Om.to(dev), meanT.to(dev), q.to(dev), var_u.to(dev), eps.to(dev),
                                      Y.to(dev), noise_u.to(dev))
    ad = al.double()
    W = torch.einsum("lmk,kc->lmc", Om, ad)
    v = (W * ad[None]).sum(1)
    var = (var_u.double().exp() - q)[None] + v + 2e-5
    e = eps.double().reshape(C, L).t()
    F = meanT.double() + var.sqrt() * e
    sN = noise_u.double().exp() + 1e-5
    r = Y.double().t().repeat(1, S) - F
    dF = -r / (sN * sN * S)
    gw = dF * e * 0.5 / var.sqrt()
    rel = lambda a, b: float((a.cpu().double() - b).norm() / max(float(b.norm()), 1e-30))
    errs = (rel(dm, dF), rel(g, gw), rel(abar, 2.0 * torch.einsum("lc,lmc->mc", gw, W)),
            abs(float(z2) - float(((r / sN) ** 2).sum())) / float(((r / sN) ** 2).sum()))
    worst = max(worst, *errs)
    flag = "" if max(errs) < 2e-5 else "   <-- FAIL"
    print(f"M={M:3d} N={N:4d} S={S} L={L:2d}: dmean {errs[0]:.1e} g {errs[1]:.1e} abar {errs[2]:.1e} z2 {errs[3]:.1e}{flag}",
          flush=True)
print("worst", worst)
sys.exit(0 if worst < 2e-5 else 1)
