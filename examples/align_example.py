"""Drop-in use of the MI355X build on the reference example's data (the 2 x 100-spot, 30-output problem of
examples/synthetic_data.h5ad, kept here as the inputs of tests/golden/c1_example_fixed0.npz), or on a
simulated lattice:  python examples/align_example.py [--simulate GRID] [--epochs N] [--graphed]

Same model arguments as the reference example (examples/grid_example.py:13-56: 25 inducing points per view
and for the data GP, RBF kernels, view 0 fixed), the loop through spatial_alignment_amd.train.fit.
Needs the HIP library (python -c "import __graft_entry__ as g; g.build()") and an MI355X."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spatial_alignment_amd as gp  # noqa: E402
from spatial_alignment_amd import simulate  # noqa: E402
from spatial_alignment_amd.train import fit  # noqa: E402


def load_problem(args, device):
    if args.simulate:
        X, Y, nsl, _ = simulate.generate_twod_data(2, 30, args.simulate, fixed_view_idx=0, device=device, seed=0)
        return X, Y, nsl
    z = np.load(os.path.join(ROOT, "tests", "golden", "c1_example_fixed0.npz"))
    X = torch.from_numpy(np.array(z["in/X/expression"])).float().to(device)
    Y = torch.from_numpy(np.array(z["in/Y/expression"])).float().to(device)
    return X, Y, [X.shape[0] // 2] * 2


def misalignment(model, X, nsl):
    """|aligned view 0 - aligned view 1| over the spots (the two views observe the same lattice)"""
    dd_x = {"expression": X}
    view_idx, Ns, _, _ = model.create_view_idx_dict(model_data)
    with torch.no_grad():
        G = model.forward(dd_x, view_idx, Ns, S=1, prediction_mode=True)[0]["expression"]
    model.train()
    return float((G[: nsl[0]] - G[nsl[0]:]).norm())


ap = argparse.ArgumentParser()
ap.add_argument("--simulate", type=int, default=0, help="lattice side of a simulated problem instead of the file data")
ap.add_argument("--epochs", type=int, default=1000)
ap.add_argument("--graphed", action="store_true", help="replay the step as one hipGraph (launch-bound sizes)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
X, Y, nsl = load_problem(args, dev)
model_data = {"expression": {"spatial_coords": X, "outputs": Y, "n_samples_list": nsl}}
model = gp.VariationalGPSA(model_data, n_spatial_dims=2, m_X_per_view=25, m_G=25, data_init=True,
                           n_latent_gps={"expression": None}, kernel_func_warp=gp.rbf_kernel,
                           kernel_func_data=gp.rbf_kernel, fixed_view_idx=0).to(dev)
before = float((X[: nsl[0]] - X[nsl[0]:]).norm())
t0 = time.perf_counter()
trace = fit(model, model_data, args.epochs, lr=1e-2, S=5, sync_every=100, graphed=args.graphed,
            callback=lambda it, m, tr: print(f"step {it + 1:5d}  loss {tr[-1]:.4g}", flush=True))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{len(trace)} steps in {dt:.2f} s ({len(trace) / dt:.0f} steps/s); "
      f"|view 0 - view 1| {before:.3f} as observed -> {misalignment(model, X, nsl):.3f} aligned")
