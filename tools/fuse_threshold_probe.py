"""Fused (lazy handles, likelihood inside the data GP's pass) against unfused step time on small problems, after a long
warm-up (the first ~1000 launches of a process run slow: clocks).  Prints 2*C*L*M^2 of the data GP next to both."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatial_alignment_amd.optim import FusedAdam
from spatial_alignment_amd.synthetic import make_grid_problem, make_model
dev = torch.device("cuda:0")
for side, outputs, M in ((10, 30, 25), (20, 30, 50), (30, 30, 50), (40, 50, 100), (50, 50, 100), (70, 50, 200)):
    dd = make_grid_problem(side=side, n_views=2, n_outputs=outputs, device="cpu")
    model = make_model(dd, m=M, device=dev)
    dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev), "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: d["spatial_coords"] for m, d in dd.items()}
    opt = FusedAdam(model.parameters(), lr=1e-3)
    def step():
        out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
        loss = model.loss_fn(dd, out[3]); opt.zero_grad(); loss.backward(); opt.step()
    res = {}
    for rep in range(2):
        for fuse in (True, False):
            model.fuse_elbo = fuse
            for _ in range(150): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): step()
            torch.cuda.synchronize(); res[fuse] = (time.perf_counter() - t0) / 200 * 1e3
    fl = 2.0 * 5 * 2 * side * side * outputs * M * M
    print(f"N=2x{side*side} L={outputs} M={M}: 2CLM^2 = {fl:.2e}  fused {res[True]:.3f} ms  unfused {res[False]:.3f} ms")
