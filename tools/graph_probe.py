"""Diagnostic (GPU): the step engine's hipGraph cache on a launch-bound problem (BASELINE config 1's size) - replay
counters and ms/step with the cache on and off, FusedAdam and the reference's verbatim loop (torch.optim.Adam +
loss.item()).   python tools/graph_probe.py [steps]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatial_alignment_amd.optim import FusedAdam
from spatial_alignment_amd.synthetic import make_grid_problem, make_model

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600


def run(enable, verbatim):
    dd = make_grid_problem(side=10, n_views=2, n_outputs=30, device="cpu")
    model = make_model(dd, m=25, device=dev, fixed_view_idx=0)
    dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev),
              "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: d["spatial_coords"] for m, d in dd.items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-2) if verbatim else FusedAdam(model.parameters(), lr=1e-2)

    def step():
        out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
        loss = model.loss_fn(dd, out[3])
        opt.zero_grad()
        loss.backward()
        opt.step()
        if verbatim:
            loss.item()

    step()
    for plan in model._step_plans.values():
        plan.lib.gpsa_step_graph(plan.handle, enable, None)
    for _ in range(1000 if not hasattr(run, 'warm') else 200):
        step()
    run.warm = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out = (C.c_longlong * 4)()
    tot = [0, 0, 0, 0]
    for plan in model._step_plans.values():
        plan.lib.gpsa_step_graph(plan.handle, -1, out)
        tot = [a + int(b) for a, b in zip(tot, out)]
    print(f"cache {'on ' if enable else 'off'} {'verbatim (torch Adam + item)' if verbatim else 'FusedAdam, no sync':30s} "
          f"{1e3 * dt:.3f} ms/step = {1 / dt:7.1f} steps/s   [replays, eager, captures, held] = {tot}", flush=True)


for rep in range(3):  # (alternating: the boxes' host speed drifts by tens of per cent between runs)
    for verbatim in (False, True):
        for enable in ((1, 0) if os.environ.get("PROBE_ALL_OFF") != "1" else (0, 0)):
            run(enable, verbatim)
