"""Which device kernels / copies each phase of the reference loop launches (torch.profiler):  python tools/phase_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatial_alignment_amd.optim import FusedAdam
from spatial_alignment_amd.synthetic import make_grid_problem, make_model

dev = torch.device("cuda:0")
dd = make_grid_problem(side=int(os.environ.get("SIDE", "40")), n_views=2, n_outputs=50, device="cpu")
model = make_model(dd, m=200, device=dev)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev), "n_samples_list": d["n_samples_list"]}
      for m, d in dd.items()}
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
opt = FusedAdam(model.parameters(), lr=1e-2)


def phases():
    yield "forward", lambda: model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
    yield "loss_fn", lambda out: model.loss_fn(dd, out[3])
    yield "zero_grad", lambda: opt.zero_grad()
    yield "backward", lambda loss: loss.backward()
    yield "opt.step", lambda: opt.step()


for it in range(4):
    out = loss = None
    for name, fn in phases():
        prof = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) if it == 3 else None
        if prof:
            prof.__enter__()
        if name == "loss_fn":
            loss = fn(out)
        elif name == "backward":
            fn(loss)
        else:
            r = fn()
            if name == "forward":
                out = r
        torch.cuda.synchronize()
        if prof:
            prof.__exit__(None, None, None)
            names = [e.name[:70] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
            odd = [n for n in names if "copy" in n.lower() or "Memcpy" in n or "fill" in n.lower() or "at::native" in n]
            print(f"{name:10s} {len(names):3d} device ops; copies / fills / torch-native: {odd}")
