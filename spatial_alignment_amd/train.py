"""Training-loop helpers around the hot path: the reference's loop body
(examples/grid_example.py:62-78) as a function, and the same step captured once into a hipGraph and
replayed — ~470 kernel launches per step become one graph launch, which is what matters for the
launch-bound small configurations (config 1, S=1).
"""
import torch


def train_step(model, optimizer, data_dict, view_idx, Ns, S=5, reducer=None, static_grads=False):
    """forward(S) + loss_fn + backward + optimizer step; returns the loss tensor (no host sync).
    ``static_grads``: keep the .grad buffers (zeroed, accumulated into) instead of letting autograd hand
    over fresh ones: needed under graph capture, ~one extra launch per parameter otherwise."""
    Xs = {m: d["spatial_coords"] for m, d in data_dict.items()}
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=S)
    loss = model.loss_fn(data_dict, out[3])
    optimizer.zero_grad(set_to_none=not static_grads)
    loss.backward()
    if reducer is not None:
        reducer()
    optimizer.step()
    return loss


class GraphedTrainStep:
    """One training step captured into a hipGraph (torch.cuda.CUDAGraph) and replayed.

    * every kernel of the step (HIP kernels of this package, Adam, RNG) is inside the graph; replaying
      does all the work of an eager step on the same static buffers;
    * the per-forward numerics check cannot sync inside a capture: the flags are kept on the device
      and ``check()`` raises afterwards (call it every N steps);
    * the optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``);
    * single-GPU only (an all-reduce inside the graph is not attempted here).
    """

    def __init__(self, model, optimizer, data_dict, view_idx, Ns, S=5, warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainStep needs a HIP device")
        self.model, self.optimizer = model, optimizer
        self._saved_check = model.check_numerics
        model.check_numerics = False
        model.overlap_views = True  # independent views become parallel branches of the graph
        # drop every reference to an earlier autograd graph (capture needs fresh AccumulateGrad nodes
        # on the capture stream)
        model._cache = None
        model.F_latent_samples, model.F_observed_samples = {}, {}
        for p in model.parameters():
            p.grad = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up off the default stream, as torch's capture rules ask
            for _ in range(warmup):
                train_step(model, optimizer, data_dict, view_idx, Ns, S, static_grads=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = train_step(model, optimizer, data_dict, view_idx, Ns, S, static_grads=True)
            flags = model._cache.flags
            self.flags = torch.cat([f.reshape(-1).to(torch.int32) for f in flags]).abs().max()

    def step(self):
        self.graph.replay()
        return self.loss

    def check(self):
        """host sync: raise if any step since the last check hit a non-PD covariance / bad variance"""
        if int(self.flags.item()) != 0:
            raise torch.linalg.LinAlgError("GPSA graphed step: non-positive-definite covariance")
