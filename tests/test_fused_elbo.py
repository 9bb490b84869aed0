"""The fused ELBO step (forward(_fuse_loss=data_dict): variance, draw, Gaussian likelihood, its gradient and the
backward's abar inside the data GP's one pass over the products, gpsa_quadform_elbo_f32) against the separate kernels
and against the reference's fp64 run: same loss, same parameter gradients."""
import numpy as np
import pytest
import torch

from golden_io import CASES, Golden, rel
from model_util import build_model, compare

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(model, dd, g, fuse, gscale=None):
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: dd[m]["spatial_coords"] for m in g.mods}
    model.inject_noise(g.eps_G, g.eps_F, None)
    model.zero_grad()
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=g.S, _fuse_loss=dd if fuse else None)
    loss = model.loss_fn(dd, out[3])
    (loss if gscale is None else loss * gscale).backward()
    res = {"loss": loss.detach().cpu().numpy()}
    for k, p in model.named_parameters():
        res[f"grad/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy()
    return res, model._cache.fuse


TRAIN = [c for c in CASES if Golden(c).G_test is None]


@pytest.mark.parametrize("name", TRAIN)
def test_fused_matches_separate_kernels(name):
    g = Golden(name)
    res = {}
    for fuse in (True, False):
        model, dd = build_model(g, device=DEV)
        res[fuse], rec = run(model, dd, g, fuse)
        if fuse:
            lmc = any(model.n_latent_gps[m] is not None for m in g.mods)
            if rec is None:  # nothing fusable in this case: say why
                assert lmc or model.Gtilde.shape[0] > 208, "an eligible case ran unfused"
                pytest.skip("no modality of this case can run fused (LMC / more than 13 row tiles)")
            assert any(rec["mods"])
        else:
            assert rec is None
    big = bool(g.cfg.get("summary_only"))
    worst = {}
    for k, want in res[False].items():
        got = res[True][k]
        assert got.shape == want.shape, k
        if np.isnan(want).any() or np.linalg.norm(want.astype(np.float64)) == 0:
            assert np.allclose(np.nan_to_num(got), np.nan_to_num(want)), k
            continue
        worst[k] = rel(got, want)
    print(name, {k: f"{v:.1e}" for k, v in worst.items()})
    for k, e in worst.items():
        assert e <= ((3e-3 if big else 3e-5) if k.startswith("grad/") else 2e-6), (k, e)


@pytest.mark.parametrize("name", TRAIN)
def test_fused_matches_reference_fp64(name):
    g = Golden(name)
    model, dd = build_model(g, device=DEV)
    res, rec = run(model, dd, g, True)
    if rec is None:
        pytest.skip("not fusable")
    big = bool(g.cfg.get("summary_only"))
    bad, errs = compare(res, g, tol_out=1e-4, tol_grad=1e-4 if not big else 3e-3)
    print(name, {k: f"{v:.1e}" for k, v in errs.items()})
    assert not bad, bad


def test_fused_upstream_gradient_scales_everything():
    """(loss * 0.37).backward(): the fused quantities are formed at upstream gradient 1 and scaled in the backward"""
    g = Golden(TRAIN[0])
    out = {}
    for fuse in (True, False):
        model, dd = build_model(g, device=DEV)
        out[fuse], _ = run(model, dd, g, fuse, gscale=0.37)
    for k, want in out[False].items():
        if np.linalg.norm(want.astype(np.float64)) == 0:
            continue
        assert rel(out[True][k], want) <= 3e-5, k


def test_fused_handles_are_only_for_loss_fn():
    g = Golden(TRAIN[0])
    model, dd = build_model(g, device=DEV)
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: dd[m]["spatial_coords"] for m in g.mods}
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=g.S, _fuse_loss=dd)
    if model._cache.fuse is None:
        pytest.skip("not fusable")
    other = {m: dict(d, outputs=d["outputs"].clone()) for m, d in dd.items()}
    with pytest.raises(ValueError):
        model.loss_fn(other, out[3])
    # the step's backward without the loss's: refused, not silently wrong
    model2, dd2 = build_model(g, device=DEV)
    out2 = model2.forward(Xs, view_idx=view_idx, Ns=Ns, S=g.S, _fuse_loss=dd2)
    with pytest.raises(RuntimeError):
        out2[3][g.mods[0]].sum().backward()
    # a hand-written loop (no _fuse_loss) gets real draws
    out3 = model2.forward(Xs, view_idx=view_idx, Ns=Ns, S=g.S)
    assert out3[3][g.mods[0]].dim() == 3 and model2._cache.fuse is None
