"""Is the eager step launch-bound?  Host time to ENQUEUE a step (no sync inside) vs wall time per step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from spatial_alignment_amd.parallel import shard_data_dict
from spatial_alignment_amd.synthetic import make_grid_problem, make_model

shard = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
dd_full = make_grid_problem(side=100, n_views=2, n_outputs=50, device="cpu")
model = make_model(dd_full, m=200, device=dev)
dd = shard_data_dict(dd_full, 0, shard)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev),
          "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
model.check_numerics = False
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
from spatial_alignment_amd.optim import FusedAdam
opt = FusedAdam(model.parameters(), lr=1e-2)
S = int(os.environ.get("GPSA_S", "5"))

def step():
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=S)
    loss = model.loss_fn(dd, out[3])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
enq = 0.0
for _ in range(30):
    a = time.perf_counter()
    step()
    enq += time.perf_counter() - a
    torch.cuda.synchronize()  # drain: the next step's enqueue time is then pure host work
wall_sync = (time.perf_counter() - t0) / 30
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 30
print(f"shard 1/{shard}: host enqueue {enq / 30 * 1e3:.2f} ms/step, wall (pipelined) {wall * 1e3:.2f} ms/step, "
      f"wall with a drain after every step {wall_sync * 1e3:.2f} ms/step")

if len(sys.argv) > 2 and sys.argv[2] == "profile":
    import cProfile, pstats
    torch.autograd.set_multithreading_enabled(False)  # backward on this thread: visible to cProfile
    for _ in range(3):
        step()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumtime").print_stats(60)
