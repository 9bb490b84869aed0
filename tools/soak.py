"""Long eager run at the headline configuration: loss trajectory, finiteness, wall time.  Diagnostic.
usage: python tools/soak.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from spatial_alignment_amd.synthetic import make_grid_problem, make_model  # noqa: E402
from spatial_alignment_amd.train import train_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device("cuda:0")
dd_cpu = make_grid_problem(side=100, n_views=2, n_outputs=50, device="cpu")
model = make_model(dd_cpu, m=200, device=dev)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev),
          "n_samples_list": d["n_samples_list"]} for m, d in dd_cpu.items()}
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
from spatial_alignment_amd.optim import FusedAdam  # noqa: E402

opt = FusedAdam(model.parameters(), lr=1e-2)
torch.manual_seed(0)
losses = []
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = train_step(model, opt, dd, view_idx, Ns, S=5)
    if i % max(1, steps // 10) == 0 or i == steps - 1:
        losses.append((i, loss))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} steps in {dt:.2f} s ({steps / dt:.1f} steps/s incl. the numerics sync of every forward)")
print("loss:", ", ".join(f"[{i}] {float(l):.4g}" for i, l in losses))
ok = all(torch.isfinite(p).all() for p in model.parameters())
G = model.forward({"expression": dd["expression"]["spatial_coords"]}, view_idx, Ns, S=1, prediction_mode=True)[0]["expression"]
err0 = float((dd["expression"]["spatial_coords"][:10000] - dd["expression"]["spatial_coords"][10000:]).norm())
err1 = float((G[:10000] - G[10000:]).norm())
print(f"parameters finite: {ok}; |view0 - view1| coordinates before {err0:.3f} -> aligned {err1:.3f}")
