"""Host-side cost of the reference loop on a launch-bound problem (BASELINE config 1's size): cProfile of N steps.
    python tools/host_profile.py [steps] [verbatim]
``verbatim``: examples/grid_example.py:59-78 as written - torch.optim.Adam and a host read of the loss every step."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatial_alignment_amd.optim import FusedAdam
from spatial_alignment_amd.synthetic import make_grid_problem, make_model

dev = torch.device("cuda:0")
dd = make_grid_problem(side=10, n_views=2, n_outputs=30, device="cpu")
model = make_model(dd, m=25, device=dev, fixed_view_idx=0)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev), "n_samples_list": d["n_samples_list"]}
      for m, d in dd.items()}
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
VERBATIM = "verbatim" in sys.argv[1:]
opt = torch.optim.Adam(model.parameters(), lr=1e-2) if VERBATIM else FusedAdam(model.parameters(), lr=1e-2)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300


def step():
    out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
    loss = model.loss_fn(dd, out[3])
    opt.zero_grad()
    loss.backward()
    opt.step()
    if VERBATIM:
        loss.item()


for _ in range(1000):  # (the first ~1000 steps of a process run slow on these boxes)
    step()
for fuse in (True, False):
    model.fuse_elbo = fuse
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print(f"fuse_elbo={fuse}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/step")
model.fuse_elbo = True
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)

# device side of one step in each mode (torch.profiler: kernel durations)
for fuse in (True, False):
    model.fuse_elbo = fuse
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    tot = sum(e.device_time for e in evs)
    print(f"fuse_elbo={fuse}: {len(evs)} device ops, {tot:.0f} us of kernels; longest:")
    for e in sorted(evs, key=lambda e: -e.device_time)[:6]:
        print(f"    {e.device_time:8.1f} us  {e.name[:90]}")

# host time per phase (no synchronisation inside the step: what the Python thread spends enqueueing)
for fuse in (True, False):
    model.fuse_elbo = fuse
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    acc = [0.0] * 6
    for _ in range(n):
        t = [time.perf_counter()]
        out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5); t.append(time.perf_counter())
        loss = model.loss_fn(dd, out[3]); t.append(time.perf_counter())
        opt.zero_grad(); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        if VERBATIM:
            loss.item()
        t.append(time.perf_counter())
        for i in range(6):
            acc[i] += t[i + 1] - t[i]
    torch.cuda.synchronize()
    print(f"fuse_elbo={fuse}: host us/step  forward {1e6*acc[0]/n:.0f}  loss_fn {1e6*acc[1]/n:.0f}  zero_grad {1e6*acc[2]/n:.0f}  "
          f"backward {1e6*acc[3]/n:.0f}  opt.step {1e6*acc[4]/n:.0f}  loss.item {1e6*acc[5]/n:.0f}")
