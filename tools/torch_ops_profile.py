"""Which torch-native (non-view) ops run inside one training step of the default bench workload, and the
host line that issued each (autograd-engine = a built-in backward node).  Diagnostic tool.
usage: python tools/torch_ops_profile.py [shard_factor]"""
import os
import sys
import traceback
from collections import Counter

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
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
from spatial_alignment_amd.optim import FusedAdam  # noqa: E402
from spatial_alignment_amd.train import backward as train_backward  # noqa: E402

opt = FusedAdam(model.parameters(), lr=1e-2)


def step():
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=5)
    loss = model.loss_fn(dd, out[3])
    opt.zero_grad(set_to_none=True)
    train_backward(loss)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
VIEWS = {"view", "select", "slice", "unsqueeze", "squeeze", "expand", "transpose", "t", "detach", "alias",
         "_unsafe_view", "reshape", "unbind", "split", "split_with_sizes", "as_strided", "permute", "empty",
         "empty_like", "empty_strided", "set_", "lift_fresh", "is_same_size", "sym_size", "sym_stride",
         "_local_scalar_dense", "item", "new_empty_strided", "record_stream", "is_pinned", "resize_"}
cn = Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEWS:
            fr = [f for f in traceback.extract_stack() if "spatial_alignment_amd" in f.filename]
            where = f"{fr[-1].filename.split('/')[-1]}:{fr[-1].lineno} {(fr[-1].line or '')[:80]}" if fr else "autograd-engine / optimizer"
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            cn[(where, name, str(shapes))] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    step()
torch.cuda.synchronize()
for (where, name, shapes), v in sorted(cn.items()):
    print(f"{v:3d} {name:22s} {shapes:40s} | {where}")
print("total", sum(cn.values()))
