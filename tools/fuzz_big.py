"""Random shapes through the large-M contraction entry points (M > 256: big_quad / big_accum / gram_big, their padded
paths for column counts that are not multiples of 4, the kept-products forward) against fp64 torch.
usage: python tools/fuzz_big.py [n_cases] [seed]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from spatial_alignment_amd import ops as ops_mod  # noqa: E402

o = ops_mod.get_ops()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
worst = 0.0
for case in range(n_cases):
    M, L = ri(257, 1100), ri(1, 24)
    C = ri(128, 6000) if case % 3 else 4 * ri(32, 1500)
    al = torch.randn(M, C, generator=gen)
    A = torch.randn(L, M, M, generator=gen, dtype=torch.float64) / M ** 0.5
    Om = A @ A.transpose(1, 2) + 1e-5 * torch.eye(M, dtype=torch.float64)
    g = torch.randn(L, C, generator=gen)
    ald, Omd, gd = al.cuda(), Om.cuda(), g.cuda()
    a64 = ald.double()
    W = torch.einsum("lmk,kc->lmc", Omd, a64)
    want = {"fwd": (W * a64[None]).sum(1), "bwd_alpha": 2.0 * torch.einsum("lc,lmc->mc", gd.double(), W),
            "bwd_omega": torch.einsum("lc,mc,kc->lmk", gd.double(), a64, a64)}
    got = {"fwd": o.quadform_fwd(ald, Omd), "bwd_alpha": o.quadform_bwd_alpha(ald, Omd, gd),
           "bwd_omega": o.quadform_bwd_omega(ald, gd)}
    errs = {k: float((got[k].double() - want[k]).norm() / want[k].norm()) for k in want}
    worst = max(worst, *errs.values())
    flag = "" if max(errs.values()) < 3e-5 else "   <-- FAIL"
    print(f"M={M:5d} C={C:5d} L={L:3d}  " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()) + flag, flush=True)
    assert not flag, (M, C, L, errs)
print("worst", worst)
