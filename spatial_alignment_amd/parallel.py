"""Data-parallel sharding of the GPSA step over the GPUs of one node.

Everything N-scaled is independent per spot given the (replicated) M x M factors, so each rank owns
a contiguous slice of EVERY view's rows: it runs the warp + data layers and the likelihood for its
rows, adds 1/world of the KL terms, and one all-reduce (RCCL over xGMI; backend "nccl" on ROCm) of
the flattened gradient makes every rank's gradient the full-ELBO gradient.  No other collective.
"""
import torch
import torch.distributed as dist


def shard_rows(n, rank, world):
    """[lo, hi) of rank's contiguous slice of n rows (sizes differ by at most 1)"""
    base, rem = divmod(int(n), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_data_dict(data_dict, rank, world):
    """per-rank data_dict: the same views, each cut to this rank's slice of its rows"""
    out = {}
    for mod, d in data_dict.items():
        ns = [int(x) for x in d["n_samples_list"]]
        rows, new_ns, off = [], [], 0
        for n in ns:
            lo, hi = shard_rows(n, rank, world)
            rows.append(torch.arange(off + lo, off + hi))
            new_ns.append(hi - lo)
            off += n
        idx = torch.cat(rows).to(d["spatial_coords"].device)
        out[mod] = {
            "spatial_coords": d["spatial_coords"][idx].contiguous(),
            "outputs": d["outputs"][idx].contiguous(),
            "n_samples_list": new_ns,
        }
    return out


class GradAllReducer:
    """One fused all-reduce(sum) of all parameter gradients per step (bucket = everything: 8.6 MB at
    the headline config, latency-bound on xGMI, so a single flat buffer is the right shape)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.empty(self.numel, dtype=p0.dtype, device=p0.device)
            self.views = list(self.flat.split([p.numel() for p in self.params]))
        for p in self.params:  # a parameter the step did not touch contributes zeros
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        # pack (one batched launch), reduce, unpack (one batched launch): not one copy per parameter
        torch.cat([p.grad.reshape(-1) for p in self.params], out=self.flat)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        torch._foreach_copy_([p.grad.reshape(-1) for p in self.params], self.views)
