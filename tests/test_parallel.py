"""Data-parallel path with world_size 2 over gloo on the CPU (fake backend): the
all-reduced gradient of the row-sharded step equals the single-process gradient of the full problem."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spatial_alignment_amd.parallel import shard_data_dict, shard_rows

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_rows_partition():
    for n, w in [(10, 3), (7, 7), (100, 8), (5, 2)]:
        cuts = [shard_rows(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def _problem():
    from spatial_alignment_amd.synthetic import make_grid_problem, make_model

    dd = make_grid_problem(side=8, n_views=2, n_outputs=3)
    model = make_model(dd, m=9)
    return dd, model


def _grads(model, dd, eps_G, eps_F, kl_scale=1.0):
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    model.kl_scale = kl_scale
    model.inject_noise(eps_G, eps_F)
    model.zero_grad()
    out = model.forward({"expression": dd["expression"]["spatial_coords"]}, view_idx, Ns, S=2)
    loss = model.loss_fn(dd, out[3])
    loss.backward()
    return loss.detach()


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    from fake_ops import FakeOps
    from spatial_alignment_amd import ops as ops_mod
    from spatial_alignment_amd.parallel import GradAllReducer

    ops_mod.set_ops(FakeOps())
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dd, model = _problem()
    gen = torch.Generator().manual_seed(3)
    eG = [torch.randn(2, 64, 2, generator=gen) for _ in range(2)]
    eF = torch.randn(2, 128, 3, generator=gen)
    sdd = shard_data_dict(dd, rank, world)
    lo, hi = shard_rows(64, rank, world)
    rows = torch.cat([torch.arange(lo, hi), 64 + torch.arange(lo, hi)])
    loss = _grads(model, sdd, [e[:, lo:hi] for e in eG], {"expression": eF[:, rows]}, 1.0 / world)
    GradAllReducer(model.parameters())()
    dist.all_reduce(loss)
    if rank == 0:
        q.put((float(loss), {k: p.grad.clone().numpy() for k, p in model.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_equals_full_step():
    from fake_ops import FakeOps
    from spatial_alignment_amd import ops as ops_mod

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss2, g2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ops_mod.set_ops(FakeOps())
    try:
        dd, model = _problem()
        gen = torch.Generator().manual_seed(3)
        eG = [torch.randn(2, 64, 2, generator=gen) for _ in range(2)]
        eF = torch.randn(2, 128, 3, generator=gen)
        loss1 = _grads(model, dd, eG, {"expression": eF})
        assert abs(float(loss1) - loss2) <= 1e-4 * abs(float(loss1))
        for k, p in model.named_parameters():
            a, b = p.grad.numpy(), g2[k]
            assert np.linalg.norm(a - b) <= 2e-3 * max(np.linalg.norm(a), 1e-6), k
    finally:
        ops_mod.set_ops(None)
