"""BUILD-CONTAINER ONLY (imports /root/reference): the CPU oracle against the unmodified reference on RANDOM configurations,
in fp64 - views (2-4, unequal sizes), spatial dims (1-3), one or two modalities, LMC or not, the three covariance
functions for either GP, fixed views (none / an int / a list), S, G_test or not.  The 11 committed goldens pin the oracle
on fixed cases; this sweeps around them.  Nothing is written: it prints the worst relative error per quantity.
usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_oracle_random.py [n_cases] [seed0]"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402  (the reference import, the noise tap, build / step)
from oracle import gpsa_oracle as orc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    nb = np.linalg.norm(b)
    if nb == 0 or np.isnan(b).any():
        return 0.0 if np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(np.nan_to_num(a), np.nan_to_num(b)) else 1.0
    return float(np.linalg.norm(a - b) / nb)


worst = {}
for case_i in range(n_cases):
    r = random.Random(seed0 + case_i)
    rng = np.random.default_rng(1000 + seed0 + case_i)
    V, D = r.choice([2, 3, 4]), r.choice([1, 2, 2, 3])
    mods = r.choice([["expression"], ["expression"], ["rna", "protein"]])
    X, Y, ns, lat = {}, {}, {}, {}
    for m in mods:
        sizes = [r.randint(25, 60) for _ in range(V)]
        N, P = sum(sizes), r.randint(2, 6)
        X[m] = (rng.uniform(0, 10, size=(N, D))).astype(np.float32)
        Y[m] = rng.standard_normal((N, P)).astype(np.float32)
        ns[m] = sizes
        lat[m] = r.choice([None, None, 2])
    fixed = r.choice([None, None, 0, V - 1, [0], [0, V - 1] if V > 2 else [1]])
    S = r.choice([1, 2, 3])
    mG = r.randint(5, min(20, min(min(s) for s in ns.values())))
    mX = r.randint(5, min(20, min(min(s) for s in ns.values())))
    gt = None
    if r.random() < 0.4:
        st_, nt = r.choice([1, 2]), r.randint(3, 9)
        gt = {m: rng.uniform(0, 10, size=(st_, nt, D)).astype(np.float32) for m in mods}
    case = dict(mods=mods, X=X, Y=Y, n_samples=ns, m_X=mX, m_G=mG, S=S, seed=seed0 + case_i, n_latent_gps=lat,
                kernel_warp=r.choice(list(mg.KERNELS)), kernel_data=r.choice(list(mg.KERNELS)), fixed_view_idx=fixed,
                G_test=gt)
    model, dd = mg.build(case, torch.float64)
    with torch.no_grad():  # away from the trivial initial state
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, dtype=p.dtype))
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for name in ("mean_slopes", "mean_intercepts"):
        state.setdefault(name, getattr(model, name).detach().clone())
    for attr in ("warp_kernel_variances", "warp_kernel_lengthscales", "data_kernel_lengthscale"):
        state.setdefault(attr, getattr(model, attr).detach().clone())
    mg.TAP.mode, mg.TAP.tape, mg.TAP.pos = "rec", [], 0
    ref = mg.step(model, dd, case, torch.float64)
    mg.TAP.mode = "off"
    tape = mg.TAP.tape
    eps_G = [t for (tag, t) in tape if tag[0] == "G"]
    eps_F_all = [t for (tag, t) in tape if tag[0] == "F"]
    n_free = len(eps_G) // S
    eG = [torch.stack(eps_G[v * S:(v + 1) * S]) for v in range(n_free)]
    eF, eFt, i = {}, ({} if gt is not None else None), 0
    for m in mods:
        eF[m] = eps_F_all[i]
        i += 1
        if gt is not None:
            eFt[m] = eps_F_all[i]
            i += 1
    cfg = dict(modality_names=mods, n_views=V, n_spatial_dims=D, kernel_warp=case["kernel_warp"],
               kernel_data=case["kernel_data"], n_latent_gps=lat, fixed_view_idx=fixed)
    got = orc.evaluate(state, cfg, {m: torch.tensor(X[m], dtype=torch.float64) for m in mods},
                       {m: torch.tensor(Y[m], dtype=torch.float64) for m in mods}, ns, S, eG, eF,
                       G_test=None if gt is None else {m: torch.tensor(g, dtype=torch.float64) for m, g in gt.items()},
                       eps_F_test=eFt, dtype=torch.float64)
    errs = {"loss": rel(got["loss"].numpy(), ref["loss"])}
    names = ["G_means", "G_samples", "F_latent", "F_obs"] + (["F_latent_test", "F_obs_test"] if gt is not None else [])
    for nm in names:
        for m in mods:
            errs[nm] = max(errs.get(nm, 0.0), rel(got[nm][m].numpy(), ref[f"{nm}/{m}"]))
    for k, gref in ref.items():
        if k.startswith("grad/"):
            gk = got["grads"].get(k[5:])
            if gk is not None:
                errs["grad"] = max(errs.get("grad", 0.0), rel(gk.numpy(), gref))
    bad = {k: v for k, v in errs.items() if v > 1e-8}
    print(f"case {case_i}: V={V} D={D} mods={len(mods)} lat={list(lat.values())} fixed={fixed} S={S} "
          f"kw={case['kernel_warp']} kd={case['kernel_data']} G_test={'yes' if gt is not None else 'no'}  "
          f"max err {max(errs.values()):.1e}" + (f"  BAD {bad}" if bad else ""), flush=True)
    for k, v in errs.items():
        worst[k] = max(worst.get(k, 0.0), v)
print("worst relative errors over", n_cases, "random configurations:", {k: f"{v:.1e}" for k, v in worst.items()})
