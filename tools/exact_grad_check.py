"""grad/Gtilde (and every other gradient) vs the reference's fp64 run with the exact inducing-point gradient off / on.
usage: python tools/exact_grad_check.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from golden_io import Golden, rel  # noqa: E402
from model_util import build_model, run_step  # noqa: E402

for name in ("c7_m200_conditioning", "c2_three_free_views", "c4_3d_gtest", "c5_two_modalities"):
    g = Golden(name)
    for exact in (False, True):
        model, dd = build_model(g, device="cuda:0")
        model.exact_inducing_grad = exact
        res = run_step(model, dd, g, device="cuda:0")
        ref = g.ref["ref64"]
        errs = {k: rel(v, ref[k]) for k, v in res.items() if k in ref and k.startswith("grad/") and abs(ref[k]).max() > 0}
        worst = max(errs, key=errs.get)
        print(f"{name:24s} exact={exact!s:5s} grad/Gtilde {errs.get('grad/Gtilde', float('nan')):.2e}  worst {worst} {errs[worst]:.2e}  "
              f"loss {rel(res['loss'], ref['loss']):.1e}", flush=True)

import test_hip_configs as thc  # noqa: E402
import spatial_alignment_amd.step_engine as SE  # noqa: E402

for label, kw in (("config-4 shape M=500", dict(side=14, views=8, outputs=6, M=500, S=2, fixed=0, seed=40)),
                  ("config-5 shape M=1000", dict(side=24, views=2, outputs=4, M=1000, S=1, fixed=None, seed=50))):
    for exact in (False, True):
        orig = SE.get_plan

        def patched(model, *a, _e=exact, **k):
            model.exact_inducing_grad = _e
            return orig(model, *a, **k)

        SE.get_plan = patched
        try:
            errs, gerr, _, _ = thc._step_vs_oracle(**kw)
        finally:
            SE.get_plan = orig
        worst = max(gerr, key=gerr.get)
        print(f"{label:24s} exact={exact!s:5s} grad/Gtilde {gerr['Gtilde']:.2e}  worst {worst} {gerr[worst]:.2e}  F {errs['F_samples']:.1e}",
              flush=True)
