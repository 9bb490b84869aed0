"""The C++ step engine (forward / backward as one host call each) against the per-layer path it replaces and
against the reference's fp64 run, on every golden case; injected noise, so both paths see the same draws."""
import numpy as np
import pytest
import torch

from golden_io import CASES, Golden, rel
from model_util import build_model, compare, run_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", CASES)
def test_engine_matches_per_layer_path(name):
    g = Golden(name)
    res = {}
    for eng in (True, False):
        model, dd = build_model(g, device=DEV)
        model.use_step_engine = eng
        res[eng] = run_step(model, dd, g, device=DEV)
        assert (model._cache.kl is not None) == eng  # the path that ran is the one that was asked for
    big = bool(g.cfg.get("summary_only"))
    worst = {}
    for k, want in res[False].items():
        got = res[True][k]
        assert got.shape == want.shape, k
        if np.isnan(want).any() or np.linalg.norm(want.astype(np.float64)) == 0:
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nan_to_num(np.abs(got)).max() == 0 \
                or np.allclose(np.nan_to_num(got), np.nan_to_num(want)), k
            continue
        worst[k] = rel(got, want)
    print(name, {k: f"{v:.1e}" for k, v in worst.items()})
    for k, e in worst.items():
        tol = (3e-3 if big else 1e-4) if k.startswith("grad/") else 2e-6
        assert e <= tol, (k, e)


@pytest.mark.parametrize("name", CASES)
def test_engine_matches_reference_fp64(name):
    g = Golden(name)
    model, dd = build_model(g, device=DEV)
    assert model.use_step_engine
    res = run_step(model, dd, g, device=DEV)
    assert model._cache.kl is not None
    big = bool(g.cfg.get("summary_only"))
    bad, errs = compare(res, g, tol_out=1e-4, tol_grad=1e-4)
    print(name, {k: f"{v:.1e}" for k, v in errs.items()})
    assert not bad, bad


@pytest.mark.parametrize("name", [c for c in CASES if c.startswith(("c2", "c3", "c4", "c9"))])
def test_kept_products_match_recomputed(name):
    """training's forward keeps the data GPs' products Omega_l alpha and the backward streams them; with
    ``keep_products = False`` the forward is the half-price symmetric form and the backward recomputes them:
    same outputs and gradients (different kernels, so to rounding), and the arena is the smaller one"""
    g = Golden(name)
    res, arena = {}, {}
    for keep in (True, False):
        model, dd = build_model(g, device=DEV)
        model.keep_products = keep
        res[keep] = run_step(model, dd, g, device=DEV)
        plan = next(iter(model._step_plans.values()))
        arena[keep] = (plan.saved_bytes, plan.saved_bytes_nokeep)
    assert arena[True][0] > arena[True][1]  # this plan CAN keep (M <= 256)
    big = bool(g.cfg.get("summary_only"))
    for k, want in res[False].items():
        got = res[True][k]
        if np.isnan(want).any() or np.linalg.norm(want.astype(np.float64)) == 0:
            continue
        e = rel(got, want)
        assert e <= ((3e-3 if big else 3e-5) if k.startswith("grad/") else 2e-6), (k, e)


def test_arenas_do_not_pile_up():
    """no backward to follow: nothing is kept and nothing leaks between calls; a training step hands its arena back
    right after backward (the autograd node sits in a reference cycle only the cyclic collector would break)"""
    import gc

    g = Golden("c7_m200_conditioning")
    model, dd = build_model(g, device=DEV)
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: d["spatial_coords"] for m, d in dd.items()}
    gc.disable()
    try:
        with torch.no_grad():
            model.forward(Xs, view_idx=view_idx, Ns=Ns, S=2)
            base = torch.cuda.memory_allocated()
            for _ in range(3):
                model.forward(Xs, view_idx=view_idx, Ns=Ns, S=2)
            assert torch.cuda.memory_allocated() <= base + (1 << 20)
        # forwards in training mode whose backward never runs: the node holds the model weakly, so the previous
        # arena dies with the outputs it belongs to (at most two alive at a time)
        seen = []
        for _ in range(5):
            out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=2)
            del out
            seen.append(torch.cuda.memory_allocated())
        assert seen[-1] <= seen[1] + (1 << 20), seen
        used = []
        for _ in range(5):
            out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=2)
            loss = model.loss_fn(dd, out[3])
            model.zero_grad(set_to_none=True)
            loss.backward()
            del out, loss
            used.append(torch.cuda.memory_allocated())
        plan = [p for p in model._step_plans.values() if p.S == 2][0]
        assert plan.saved_bytes - plan.saved_bytes_nokeep > (4 << 20)
        assert used[-1] <= used[1] + (1 << 20), used
    finally:
        gc.enable()
    # a SECOND backward through one forward (retain_graph=True; the reference's graph allows it): the arena went back
    # with the first one, the node fills a fresh one from the same inputs and draws - the gradients simply accumulate
    for fuse in (True, False):
        model.fuse_elbo = fuse
        model.zero_grad(set_to_none=True)
        out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=2)
        loss = model.loss_fn(dd, out[3])
        loss.backward(retain_graph=True)
        once = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        loss.backward()
        for k, p in model.named_parameters():
            if k in once and float(once[k].norm()) > 0:
                err = float((p.grad - 2 * once[k]).norm() / (2 * once[k]).norm())
                assert err <= 1e-6, (fuse, k, err)
        with pytest.raises(RuntimeError, match="second time"):  # (torch's own refusal: nothing was retained)
            loss.backward()
        del out, loss


def test_engine_unequal_views_and_user_loss_on_G():
    """views of different sizes (padding columns of the view blocks), a loss that also uses G_means and
    G_samples directly, S = 1"""
    import spatial_alignment_amd as gp

    gen = torch.Generator().manual_seed(3)
    ns = [37, 90, 5]
    N = sum(ns)
    X = (10 * torch.rand(N, 2, generator=gen)).to(DEV)
    Y = torch.randn(N, 4, generator=gen).to(DEV)
    dd = {"expression": {"spatial_coords": X, "outputs": Y, "n_samples_list": ns}}
    outs = {}
    for eng in (True, False):
        torch.manual_seed(1)
        np.random.seed(1)
        model = gp.VariationalGPSA({"expression": {"spatial_coords": X.cpu(), "outputs": Y.cpu(), "n_samples_list": ns}},
                                   m_X_per_view=9, m_G=11, data_init=False, n_latent_gps={"expression": None},
                                   fixed_view_idx=None).to(DEV)
        with torch.no_grad():
            model.delta_G_list.add_(0.1 * torch.randn(model.delta_G_list.shape, generator=gen).to(DEV) * 0 + 0.05)
        model.use_step_engine = eng
        view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
        g2 = torch.Generator().manual_seed(11)
        eps_G = [torch.randn(1, n, 2, generator=g2) for n in ns]
        eps_F = {"expression": torch.randn(1, N, 4, generator=g2)}
        model.inject_noise(eps_G, eps_F)
        Gm, Gs, Fl, Fo = model.forward({"expression": X}, view_idx, Ns, S=1)
        loss = model.loss_fn(dd, Fo) + 3.0 * (Gm["expression"] ** 2).sum() + (Gs["expression"].sin()).sum()
        loss.backward()
        outs[eng] = dict(loss=loss.detach().cpu().numpy(), Gm=Gm["expression"].detach().cpu().numpy(),
                         F=Fo["expression"].detach().cpu().numpy(),
                         **{f"grad/{k}": p.grad.detach().cpu().numpy() for k, p in model.named_parameters()
                            if p.grad is not None})
    for k, want in outs[False].items():
        if np.linalg.norm(want) == 0:
            assert np.abs(outs[True][k]).max() == 0, k
            continue
        assert rel(outs[True][k], want) <= (1e-4 if k.startswith("grad/") else 2e-6), (k, rel(outs[True][k], want))


def test_numerics_check_raises_like_the_reference():
    """a non-positive-definite covariance: torch.linalg.LinAlgError from forward (no gradients wanted, or
    check_numerics == "strict"), from the backward of the same step before any gradient of the GPs' parameters exists (training:
    the wait is deferred so that the host keeps queueing), never with check_numerics == False"""
    g = Golden("c2_three_free_views")

    def broken():
        model, dd = build_model(g, device=DEV)
        with torch.no_grad():
            model.Xtilde[1, 3, 0] = float("nan")  # K_uu of view 1 loses positive definiteness
        view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
        Xs = {m: dd[m]["spatial_coords"] for m in g.mods}
        return model, dd, Xs, view_idx, Ns

    model, dd, Xs, view_idx, Ns = broken()
    with torch.no_grad(), pytest.raises(torch.linalg.LinAlgError):
        model.forward(Xs, view_idx, Ns, S=2)
    model, dd, Xs, view_idx, Ns = broken()
    model.check_numerics = "strict"
    with pytest.raises(torch.linalg.LinAlgError):
        model.forward(Xs, view_idx, Ns, S=2)
    model, dd, Xs, view_idx, Ns = broken()
    out = model.forward(Xs, view_idx, Ns, S=2)  # training: deferred
    loss = model.loss_fn(dd, out[3])
    with pytest.raises(torch.linalg.LinAlgError):
        loss.backward()
    # (the likelihood node ran first: only its own noise parameter has a gradient by then)
    assert all(p.grad is None for k, p in model.named_parameters() if k != "noise_variance")
    model, dd, Xs, view_idx, Ns = broken()
    out = model.forward(Xs, view_idx, Ns, S=2)  # backward never runs: the next forward reports it
    with pytest.raises(torch.linalg.LinAlgError):
        model.forward(Xs, view_idx, Ns, S=2)
    model, dd, Xs, view_idx, Ns = broken()
    model.check_numerics = False
    model.forward(Xs, view_idx, Ns, S=2)


def test_graphed_step_folds_the_engine_flag(monkeypatch):
    """Inside a capture nobody waits for the step engine's numerics word; the sticky word of GraphedTrainStep
    must fold it in: a non-positive pivot is replaced by 1 and the step trains on finite garbage.  The loss's own
    non-finite bit is disabled here so that only the engine's word can trip check()."""
    from spatial_alignment_amd.train import GraphedTrainStep

    g = Golden("c2_three_free_views")
    model, dd = build_model(g, device=DEV)
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
    real = torch.isfinite
    monkeypatch.setattr(torch, "isfinite", lambda t: torch.ones_like(t, dtype=torch.bool))
    gs = GraphedTrainStep(model, opt, dd, view_idx, Ns, S=2, warmup=3)
    monkeypatch.setattr(torch, "isfinite", real)
    assert gs.engine_flag is not None
    gs.step()
    gs.check()  # healthy parameters: nothing to report
    with torch.no_grad():
        model.Xtilde[1, 3, 0] = float("nan")  # K_uu of view 1 loses positive definiteness (the graph reads it in place)
    gs.step()
    torch.cuda.synchronize()
    assert int(gs.engine_flag.item()) != 0
    with pytest.raises(torch.linalg.LinAlgError):
        gs.check()


def test_engine_validates_what_it_hands_to_cxx():
    """shapes the C++ engine would trust blindly: a modality's G_test with another sample count, coordinates
    shorter than the views' row counts, injected draws of the wrong size -> ValueError, not an out-of-bounds read"""
    g = Golden("c5_two_modalities")
    model, dd = build_model(g, device=DEV)
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: dd[m]["spatial_coords"] for m in g.mods}
    D = Xs[g.mods[0]].shape[1]
    Gt = {g.mods[0]: torch.randn(2, 5, D, device=DEV), g.mods[1]: torch.randn(1, 5, D, device=DEV)}
    with pytest.raises(ValueError):
        model.forward(Xs, view_idx, Ns, S=2, G_test=Gt)
    short = dict(Xs)
    short[g.mods[1]] = Xs[g.mods[1]][:-1]
    with pytest.raises(ValueError):
        model.forward(short, view_idx, Ns, S=2)
    model.inject_noise(None, {m: torch.randn(2, 3, 1) for m in g.mods}, None)
    with pytest.raises(ValueError):
        model.forward(Xs, view_idx, Ns, S=2)
    model._noise = None
    model.forward(Xs, view_idx, Ns, S=2)  # and the well-formed call still runs


def test_deferred_check_belongs_to_its_own_forward():
    """forward A (healthy), forward B on broken coordinates, backward A: A's backward must not retire B's pending
    numerics check - the next forward still reports it"""
    g = Golden("c2_three_free_views")
    model, dd = build_model(g, device=DEV)
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: dd[m]["spatial_coords"] for m in g.mods}
    outA = model.forward(Xs, view_idx, Ns, S=2)
    lossA = model.loss_fn(dd, outA[3])
    bad = {m: x.clone() for m, x in Xs.items()}
    bad[g.mods[0]][-1, 0] = float("nan")         # a spot of the last (free) view: its warp variance is not > 0
    outB = model.forward(bad, view_idx, Ns, S=2)  # (checks A's word on the way in: healthy)
    lossA.backward()                             # A's own word again: healthy, and B's stays pending
    with pytest.raises(torch.linalg.LinAlgError):
        model.forward(Xs, view_idx, Ns, S=2)
    del outB


def test_step_graph_cache_with_fixed_buffers():
    """the engine's hipGraph cache (csrc/step.hip; experimental, off by default): a caller that hands gpsa_step_forward the SAME
    buffers again - here: stage 1 of a forward repeated on its own arena and outputs - gets the launch sequence replayed
    as one graph launch from the third call on, with bitwise the same results as the eager call"""
    import ctypes as C

    from spatial_alignment_amd import ops as ops_mod
    from spatial_alignment_amd import torch_ops as TO

    g = Golden("c2_three_free_views")
    model, dd = build_model(g, device=DEV)
    model.fuse_min_flops = 0
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: d["spatial_coords"] for m, d in dd.items()}
    model.inject_noise(g.eps_G, g.eps_F, None)
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=g.S)     # a training forward: stage 1 ran, loss_fn has not
    live = model._cache.fuse["live"]
    plan, io, prm, saved, tensors, ins = (live[k] for k in ("plan", "io", "prm", "saved", "tensors", "ins"))
    m = g.mods[0]
    Gm, Gs = out[0][m].detach(), out[1][m].detach()
    want = (Gm.clone(), Gs.clone())
    scratch = ops_mod.get_ops()._ws(plan.scratch_bytes, saved)
    stats = (C.c_longlong * 4)()
    assert plan.lib.gpsa_step_graph(plan.handle, 1, None) == 0
    try:
        for rep in range(4):  # 1st: first sighting (eager), 2nd: captured and launched, 3rd and 4th: replays
            Gm.zero_()
            Gs.zero_()
            call = TO.stash(dict(lib=plan.lib, handle=plan.handle, prm=prm, io=io))
            try:
                torch.ops.gpsa.step_forward(list(tensors), ins, [Gm, Gs], saved, scratch, call, 1)
            finally:
                TO.CALLS.pop(call, None)
            torch.cuda.synchronize()
            assert torch.equal(Gm, want[0]) and torch.equal(Gs, want[1]), rep
    finally:
        plan.lib.gpsa_step_graph(plan.handle, 0, stats)
    print("graph cache [replays, eager, captures, held]:", list(stats))
    assert stats[0] >= 2 and stats[2] >= 1, list(stats)


def test_step_graph_cache_training_trajectory_is_bitwise():
    """Two models in one process, each trained twice from the same seed with the host running ahead of the device: the
    engine's hipGraph cache on (replaying most calls, evicting the least recently used graphs of a deliberately small
    pool) against off - the loss trajectories agree bit for bit (tools/graph_stress.py is the long form)."""
    import ctypes as C

    from spatial_alignment_amd.optim import FusedAdam
    from spatial_alignment_amd.synthetic import make_grid_problem, make_model

    def run(seed, enable):
        torch.cuda.empty_cache()  # (a fresh allocator repeats its block pattern within a few steps: replays)
        torch.manual_seed(seed)
        dd = make_grid_problem(side=10, n_views=2, n_outputs=30, device="cpu")
        model = make_model(dd, m=25, device=DEV, fixed_view_idx=0)
        dd = {m: {"spatial_coords": d["spatial_coords"].to(DEV), "outputs": d["outputs"].to(DEV),
                  "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
        view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
        Xs = {m: d["spatial_coords"] for m, d in dd.items()}
        opt = FusedAdam(model.parameters(), lr=1e-2)
        kept = []
        for i in range(240):
            out = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=5)
            loss = model.loss_fn(dd, out[3])
            opt.zero_grad()
            loss.backward()
            opt.step()
            if i == 0:
                for plan in model._step_plans.values():
                    assert plan.lib.gpsa_step_graph(plan.handle, enable, None) == 0
            if i % 20 == 19:
                kept.append(loss.detach())
        torch.cuda.synchronize()
        tot = [0, 0, 0, 0]
        cnt = (C.c_longlong * 4)()
        for plan in model._step_plans.values():
            plan.lib.gpsa_step_graph(plan.handle, 0, cnt)
            tot = [a + int(b) for a, b in zip(tot, cnt)]
        return [float(x) for x in kept], tot

    for k in range(2):
        ref, _ = run(7 + k, 0)
        got, tot = run(7 + k, 1)
        print("graph cache [replays, eager, captures, held]:", tot)
        assert got == ref, (k, ref[-3:], got[-3:])
        # most calls were replays - or, in a process whose allocator never hands out the same blocks twice, the cache
        # captured its quota of never-replayed graphs and stood down
        assert tot[0] > 100 or tot[2] >= 9, tot


def test_index_mutated_in_place_between_forwards_leaves_the_engine_path():
    """the engine path ignores view_idx (it knows the views as consecutive row blocks), so the check that
    lets a forward take it must see an index array that was re-ordered IN PLACE between two forwards."""
    g = Golden("c2_three_free_views")
    model, dd = build_model(g, device=DEV)
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: dd[m]["spatial_coords"] for m in g.mods}
    outs = []
    for swap in (False, True):
        if swap:  # rows of views 0 and 1 trade places, the objects and their lengths stay
            m0 = g.mods[0]
            a, b = view_idx[m0][0].copy(), view_idx[m0][1].copy()
            n = min(len(a), len(b))
            view_idx[m0][0][:n], view_idx[m0][1][:n] = b[:n], a[:n]
        model.inject_noise(g.eps_G, g.eps_F, g.eps_F_test)
        with torch.no_grad():
            out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=g.S)
        outs.append((model._cache.kl is not None, out[0][g.mods[0]].detach().cpu().numpy().copy()))
    assert outs[0][0] and not outs[1][0]        # engine first, the per-layer (index-honouring) path after the write
    assert not np.allclose(outs[0][1], outs[1][1])  # and the rows did go to the other views' warps
