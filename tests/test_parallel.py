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


def _grads(model, dd, eps_G, eps_F, kl_scale=1.0, owner=None):
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    model.kl_scale = kl_scale
    if owner is not None:  # owner computes: this rank's contiguous range of the KL terms at weight 1
        from spatial_alignment_amd.parallel import own_kl_terms

        lo, hi = own_kl_terms(model, *owner)
        assert model.kl_scale == 1.0 and 0 <= lo < hi <= 2 * 2 + 3
    model.inject_noise(eps_G, eps_F)
    model.zero_grad()
    out = model.forward({"expression": dd["expression"]["spatial_coords"]}, view_idx, Ns, S=2)
    loss = model.loss_fn(dd, out[3])
    loss.backward()
    return loss.detach()


def _worker(rank, world, port, q, owner=False):
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
    loss = _grads(model, sdd, [e[:, lo:hi] for e in eG], {"expression": eF[:, rows]}, 1.0 / world,
                  owner=(rank, world) if owner else None)
    GradAllReducer(model.parameters())()
    dist.all_reduce(loss)
    if rank == 0:
        q.put((float(loss), {k: p.grad.clone().numpy() for k, p in model.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("owner", [False, True])
def test_sharded_step_equals_full_step(owner):
    """``owner``: each rank evaluates its own range of the KL terms at weight 1 (parallel.own_kl_terms) instead of all
    of them at 1 / world - the all-reduce sums the shares to the same loss and gradients"""
    from fake_ops import FakeOps
    from spatial_alignment_amd import ops as ops_mod

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if owner else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, owner)) for r in range(2)]
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


# ---- output (L-axis) sharding: SURVEY.md §8e "alternative for large L" ---------------------------------
def _problem_outputs():
    from spatial_alignment_amd.synthetic import make_grid_problem, make_model

    dd = make_grid_problem(side=8, n_views=2, n_outputs=4)
    return dd, make_model(dd, m=9)


def _rank_model(full, dd, rank, world):
    """rank's model on its output slice, carrying the full model's parameters (shared: copies; per-output:
    the rank's rows / columns)"""
    from spatial_alignment_amd.parallel import shard_outputs, shard_rows
    from spatial_alignment_amd.synthetic import make_model

    sdd = shard_outputs(dd, rank, world)
    model = make_model(sdd, m=9, seed=100 + rank)  # deliberately different construction RNG per rank
    lo, hi = shard_rows(4, rank, world)
    with torch.no_grad():
        for (n, p), (_, pf) in zip(model.named_parameters(), full.named_parameters()):
            if n.startswith("Omega_sqt_F_dict."):
                p.copy_(pf[lo:hi])
            elif n.startswith("delta_F_dict."):
                p.copy_(pf[:, lo:hi])
            elif rank == 0:  # the other ranks get the shared parameters by broadcast
                p.copy_(pf)
    return sdd, model, (lo, hi)


def _worker_outputs(rank, world, port, q):
    sys.path.insert(0, HERE)
    from fake_ops import FakeOps
    from spatial_alignment_amd import ops as ops_mod
    from spatial_alignment_amd.parallel import setup_output_sharding

    ops_mod.set_ops(FakeOps())
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dd, full = _problem_outputs()
    sdd, model, (lo, hi) = _rank_model(full, dd, rank, world)
    reducer = setup_output_sharding(model, rank, world, seed=5)
    assert model.kl_weight_G == 0.5 and model.kl_scale == 1.0
    gen = torch.Generator().manual_seed(3)
    eG = [torch.randn(2, 64, 2, generator=gen) for _ in range(2)]
    eF = torch.randn(2, 128, 4, generator=gen)
    loss = _grads(model, sdd, eG, {"expression": eF[:, :, lo:hi]}, 1.0)
    reducer()
    dist.all_reduce(loss)
    grads = {k: p.grad.clone().numpy() for k, p in model.named_parameters()}
    gathered = [None] * world
    dist.all_gather_object(gathered, grads)
    if rank == 0:
        q.put((float(loss), gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_output_sharded_step_equals_full_step():
    from fake_ops import FakeOps
    from spatial_alignment_amd import ops as ops_mod
    from spatial_alignment_amd.parallel import shard_rows

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_outputs, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss2, per_rank = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ops_mod.set_ops(FakeOps())
    try:
        dd, model = _problem_outputs()
        gen = torch.Generator().manual_seed(3)
        eG = [torch.randn(2, 64, 2, generator=gen) for _ in range(2)]
        eF = torch.randn(2, 128, 4, generator=gen)
        loss1 = _grads(model, dd, eG, {"expression": eF})
        assert abs(float(loss1) - loss2) <= 1e-4 * abs(float(loss1))
        for k, p in model.named_parameters():
            a = p.grad.numpy()
            if k.startswith("Omega_sqt_F_dict."):  # per-output parameters: each rank holds its rows
                b = np.concatenate([per_rank[r][k] for r in range(2)], 0)
            elif k.startswith("delta_F_dict."):
                b = np.concatenate([per_rank[r][k] for r in range(2)], 1)
            else:  # shared: all-reduced, identical on both ranks
                b = per_rank[0][k]
                assert np.array_equal(per_rank[0][k], per_rank[1][k]), k
            assert np.linalg.norm(a - b) <= 2e-3 * max(np.linalg.norm(a), 1e-6), k
        assert shard_rows(4, 1, 2) == (2, 4)
    finally:
        ops_mod.set_ops(None)


def test_output_sharding_rejects_latent_mixing_and_separates_generators():
    from spatial_alignment_amd.parallel import setup_output_sharding, shared_parameters
    from spatial_alignment_amd.synthetic import make_grid_problem, make_model

    dd = make_grid_problem(side=6, n_views=2, n_outputs=4)
    with pytest.raises(ValueError):
        setup_output_sharding(make_model(dd, m=4, n_latent_gps={"expression": 2}), 0, 2)
    a, b = make_model(dd, m=4), make_model(dd, m=4)
    setup_output_sharding(a, 0, 2, seed=9)
    setup_output_sharding(b, 1, 2, seed=9)
    assert torch.equal(a._draw([3, 2], "cpu", "G"), b._draw([3, 2], "cpu", "G"))       # common warp draws
    assert not torch.equal(a._draw([3, 2], "cpu", "F"), b._draw([3, 2], "cpu", "F"))   # own output draws
    names = {n for n, _ in a.named_parameters()}
    shared = {id(p) for p in shared_parameters(a)}
    local = {n for n, p in a.named_parameters() if id(p) not in shared}
    assert local == {n for n in names if n.startswith(("Omega_sqt_F_dict.", "delta_F_dict."))} and local


def test_microbatched_step_accumulates_the_full_gradient():
    """train.Microbatches: K forward / ELBO / backward passes over row slices of every view, gradients accumulated,
    equal one pass over all rows (the "minibatched K_NM" of BASELINE config 5) - same draws, sliced"""
    from fake_ops import FakeOps
    from spatial_alignment_amd import ops as ops_mod
    from spatial_alignment_amd.train import Microbatches

    ops_mod.set_ops(FakeOps())
    try:
        dd, model = _problem()
        gen = torch.Generator().manual_seed(3)
        eG = [torch.randn(2, 64, 2, generator=gen) for _ in range(2)]
        eF = torch.randn(2, 128, 3, generator=gen)
        loss1 = _grads(model, dd, eG, {"expression": eF})
        want = {k: p.grad.clone() for k, p in model.named_parameters()}
        before = {k: p.detach().clone() for k, p in model.named_parameters()}

        class Probe(torch.optim.Optimizer):  # records the accumulated gradients at step(), changes nothing
            def __init__(self, params):
                super().__init__(params, {})
                self.seen = None

            def step(self):
                self.seen = {id(p): p.grad.clone() for g in self.param_groups for p in g["params"] if p.grad is not None}

        K = 3
        mb = Microbatches(model, dd, K)
        noise = []
        for k in range(K):
            b = mb.bounds[k]["expression"]  # slice k of view v: multiples of 4 rows, the last one the remainder
            assert all((hi - lo) % 4 == 0 for lo, hi in b) or k == K - 1
            rows = torch.cat([64 * v + torch.arange(lo, hi) for v, (lo, hi) in enumerate(b)])
            noise.append(([e[:, lo:hi] for e, (lo, hi) in zip(eG, b)], {"expression": eF[:, rows]}))
        opt = Probe(model.parameters())
        total = mb.step(opt, S=2, noise=noise)
        assert model.kl_scale == 1.0  # restored
        assert abs(float(total) - float(loss1)) <= 1e-4 * abs(float(loss1))
        for k, p in model.named_parameters():
            assert torch.equal(p.detach(), before[k])
            got = opt.seen[id(p)]
            assert (got - want[k]).norm() <= 2e-3 * max(float(want[k].norm()), 1e-6), k
    finally:
        ops_mod.set_ops(None)


def test_microbatches_refuse_a_slice_without_rows_of_a_modality():
    """every modality must have rows in every slice (the accumulator mirrors the per-pass gradient pieces
    across the slices' plans)"""
    from spatial_alignment_amd.train import Microbatches

    class M:  # (the constructor's check needs no model)
        def create_view_idx_dict(self, dd):
            raise AssertionError("reached slicing")

    x = lambda n: torch.zeros(n, 2)
    dd = {"a": {"spatial_coords": x(200), "outputs": x(200), "n_samples_list": [100, 100]},
          "b": {"spatial_coords": x(16), "outputs": x(16), "n_samples_list": [8, 8]}}
    with pytest.raises(ValueError, match="modality 'b'"):
        Microbatches(M(), dd, 4)  # slices of 4 rows: modality b's views end after two of them
