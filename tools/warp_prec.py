import sys, os, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import golden_io
from golden_io import Golden
from model_util import build_model, run_step, compare
for name in golden_io.CASES:
    g = Golden(name)
    for wm in (torch.float64, torch.float32):
        model, dd = build_model(g, "cuda")
        model._warp_main_dtype = wm
        res = run_step(model, dd, g, "cuda")
        bad, errs = compare(res, g, 1e-4, 5e-3)
        outs = {k: v for k, v in errs.items() if not k.startswith("grad/")}
        grads = {k: v for k, v in errs.items() if k.startswith("grad/")}
        print(name, wm, "max out %.2e (%s)" % (max(outs.values()), max(outs, key=outs.get)),
              "max grad %.2e (%s)" % (max(grads.values()), max(grads, key=grads.get)), "bad", {k: ("%.1e" % v[0], "%.1e" % v[1]) for k, v in bad.items()})
