"""Per-step wall time of the headline problem (synchronised after every step) for 400 steps: where the slow ones are."""
import os, sys, time, torch, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatial_alignment_amd.optim import FusedAdam
from spatial_alignment_amd.synthetic import make_grid_problem, make_model
dev = torch.device("cuda:0")
dd = make_grid_problem(side=100, n_views=2, n_outputs=50, device="cpu")
model = make_model(dd, m=200, device=dev)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev), "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
opt = FusedAdam(model.parameters(), lr=1e-3)
if len(sys.argv) > 1 and sys.argv[1] == "nogc": gc.disable()
ts = []
for i in range(400):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
    loss = model.loss_fn(dd, out[3]); opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
import statistics
med = statistics.median(ts[20:])
slow = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 1.3 * med]
print("median", round(med, 3), "ms; steps slower than 1.3 x median:", slow[:40], "count", len(slow))
print("gc counts", gc.get_count(), "mem reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2))
