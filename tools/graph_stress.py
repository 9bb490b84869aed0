"""Stress of the step engine's hipGraph cache (GPU): BASELINE config 1's size, several models one after another in one
process, each trained twice from the same seed - cache off, cache on - and the two loss trajectories compared bit for
bit.  With GPSA_STEP_GRAPH_MAX small the cache evicts on almost every call.
usage: [GPSA_STEP_GRAPH_MAX=n] [GPSA_STEP_GRAPH_UNSAFE_DESTROY=1] python tools/graph_stress.py [models] [steps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatial_alignment_amd.optim import FusedAdam  # noqa: E402
from spatial_alignment_amd.synthetic import make_grid_problem, make_model  # noqa: E402

dev = torch.device("cuda:0")
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600


def run(seed, enable):
    torch.manual_seed(seed)
    dd = make_grid_problem(side=10, n_views=2, n_outputs=30, device="cpu")
    model = make_model(dd, m=25, device=dev, fixed_view_idx=0)
    dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev),
              "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: d["spatial_coords"] for m, d in dd.items()}
    opt = FusedAdam(model.parameters(), lr=1e-2)
    losses = []
    err = None
    try:
        for i in range(n_steps):
            out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
            loss = model.loss_fn(dd, out[3])
            opt.zero_grad()
            loss.backward()
            opt.step()
            if i == 0:
                for plan in model._step_plans.values():
                    plan.lib.gpsa_step_graph(plan.handle, enable, None)
            if i % 50 == 49:
                losses.append(loss.detach())  # (no host read: the host keeps running ahead of the device)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {str(e)[:80]}"
    cnt = (C.c_longlong * 4)()
    tot = [0, 0, 0, 0]
    for plan in model._step_plans.values():
        plan.lib.gpsa_step_graph(plan.handle, -1, cnt)
        tot = [a + int(b) for a, b in zip(tot, cnt)]
    return [float(x) for x in losses], err, tot


bad = 0
for k in range(n_models):
    ref, e0, _ = run(100 + k, 0)
    got, e1, tot = run(100 + k, 1)
    same = e0 is None and e1 is None and len(ref) == len(got) and all(a == b for a, b in zip(ref, got))
    first = next((i for i, (a, b) in enumerate(zip(ref, got)) if a != b), None)
    print(f"model {k}: cache off err={e0}  cache on err={e1}  [replays, eager, captures, held]={tot}  "
          f"trajectories {'IDENTICAL' if same else 'DIFFER (first at sample %s)' % first}  final {ref[-1] if ref else None} / "
          f"{got[-1] if got else None}", flush=True)
    bad += 0 if same else 1
print("GPSA_STEP_GRAPH_MAX =", os.environ.get("GPSA_STEP_GRAPH_MAX"), " UNSAFE_DESTROY =",
      os.environ.get("GPSA_STEP_GRAPH_UNSAFE_DESTROY"), " models with a difference:", bad, flush=True)
sys.exit(1 if bad else 0)
