import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spatial_alignment_amd.optim import FusedAdam
from spatial_alignment_amd.synthetic import make_grid_problem, make_model
dev = torch.device("cuda:0")
dd = make_grid_problem(side=10, n_views=2, n_outputs=30, device="cpu")
model = make_model(dd, m=25, device=dev, fixed_view_idx=0)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev), "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
opt = FusedAdam(model.parameters(), lr=1e-2)
def step():
    out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
    loss = model.loss_fn(dd, out[3])
    opt.zero_grad()
    loss.backward()
    opt.step()
def timeit(n=500):
    for _ in range(30): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for fuse, chk in ((False, False), (True, True), (True, False), (False, True), (True, True), (False, False)):
    if True:
        model.fuse_elbo, model.check_numerics = fuse, chk
        print(f"fuse {fuse} check {chk}: {timeit():.3f} ms/step")
