"""Which torch-native ops (and from where) run inside one training step of the default bench workload.
Diagnostic tool.  usage: python tools/torch_ops_profile.py [shard_factor]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from spatial_alignment_amd.parallel import shard_data_dict  # noqa: E402
from spatial_alignment_amd.synthetic import make_grid_problem, make_model  # noqa: E402

emu = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
dd_full = make_grid_problem(side=100, n_views=2, n_outputs=50, device="cpu")
model = make_model(dd_full, m=200, device=dev)
dd = shard_data_dict(dd_full, 0, emu)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev),
          "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True)


def step():
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=5)
    loss = model.loss_fn(dd, out[3])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:80]:
    where = [s for s in e.stack if "spatial_alignment_amd" in s or "bench" in s or "tools/" in s][:2]
    print(f"{e.count:3d} x {e.key:32s} dev {e.self_device_time_total:8.1f} us  cpu {e.self_cpu_time_total:8.1f} us  {' <- '.join(w.strip()[-70:] for w in where)}")
