"""Does the training loop allocate device memory in steady state?  (the engine's saved arena is a torch tensor)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spatial_alignment_amd.synthetic import make_grid_problem, make_model
from spatial_alignment_amd.optim import FusedAdam
from spatial_alignment_amd.train import backward
dev = torch.device("cuda:0")
dd = make_grid_problem(side=100, n_views=2, n_outputs=50, device="cpu")
model = make_model(dd, m=200, device=dev)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev), "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
opt = FusedAdam(model.parameters(), lr=1e-2)
def step():
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=5)
    loss = model.loss_fn(dd, out[3])
    opt.zero_grad(set_to_none=True)
    backward(loss)
    opt.step()
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"step {i}: {dt*1e3:8.2f} ms  device allocs {st['num_device_alloc']} frees {st['num_device_free']} reserved {st['reserved_bytes.all.current']/2**30:.2f} GiB active {st['active_bytes.all.current']/2**30:.2f} GiB")
