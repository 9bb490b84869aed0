"""lazy.LazyDraws on the CPU: the handle's mechanics with the engine call replaced (what it computes is the GPU tests'
business, tests/test_fused_elbo.py)."""
import pytest
import torch

from spatial_alignment_amd import lazy


def make(monkeypatch, state="lazy", shape=(2, 5, 3)):
    calls = []
    S, N, L = shape
    vals = torch.arange(S * N * L, dtype=torch.float32).reshape(S, N, L)

    def fake_values(rec, i, attach=False):
        calls.append(("values", i, attach))
        return vals.clone()

    monkeypatch.setattr(lazy, "materialize_values", fake_values)
    parts = torch.zeros(4, dtype=torch.float64, requires_grad=True)
    rec = dict(state=[state], shapes=[shape], FT=[vals.permute(2, 0, 1).reshape(L, S * N).contiguous()], dF=[None],
               F_real=[None], live={})
    h = lazy.LazyDraws(rec, 0, shape, torch.device("cpu"))
    h._parts = parts
    return h, rec, calls, vals


def test_metadata_costs_nothing(monkeypatch):
    h, rec, calls, vals = make(monkeypatch)
    assert isinstance(h, torch.Tensor) and torch.is_tensor(h)
    assert tuple(h.shape) == (2, 5, 3) and h.size(1) == 5 and h.dim() == 3 and h.ndim == 3 and h.numel() == 30
    assert h.dtype == torch.float32 and h.device.type == "cpu" and h.requires_grad and not h.is_cuda
    assert h.is_floating_point() and h.is_contiguous()
    assert calls == [] and not h.is_materialized and rec["state"] == ["lazy"]


def test_any_use_before_loss_fn_makes_the_modality_an_unfused_one(monkeypatch):
    h, rec, calls, vals = make(monkeypatch)
    y = h * 2.0 + 1.0                      # an operator
    assert calls == [("values", 0, True)] and rec["state"] == ["real"] and h.is_materialized
    assert torch.equal(y, vals * 2 + 1)
    assert torch.equal(h[1, 2], vals[1, 2]) and torch.equal(h.detach().cpu(), vals)   # an index, .detach().cpu()
    assert torch.equal(torch.mean(h, dim=0), vals.mean(0))                           # a torch.* function
    assert "tensor(" in repr(h)
    assert len(calls) == 1                 # computed once
    # the gradient that reaches the draws is left for the step's backward and autograd walks on to the step's node
    (h * h).sum().backward()
    assert torch.equal(rec["dF"][0], 2 * vals) and h._parts.grad is not None


def test_under_no_grad_the_handle_only_shows_values(monkeypatch):
    h, rec, calls, vals = make(monkeypatch)
    with torch.no_grad():
        assert torch.equal(h.sum(), vals.sum())
    assert calls == [("values", 0, False)] and rec["state"] == ["lazy"] and not h.is_materialized


def test_after_loss_fn_the_handle_shows_the_fused_pass_draws(monkeypatch):
    h, rec, calls, vals = make(monkeypatch, state="fused")
    assert torch.equal(h.detach().mean(0), vals.mean(0))          # from the pass's own [L, S N] output: no engine call
    assert calls == []
    with pytest.raises(RuntimeError, match="fused form"):        # a gradient through them is refused, not dropped
        h.sum().backward()
    rec["live"] = None                                            # ... and they are still there after backward
    with torch.no_grad():
        assert torch.equal(h[0], vals[0])


def test_untouched_draws_are_gone_with_the_arena():
    parts = torch.zeros(4, dtype=torch.float64, requires_grad=True)
    rec = dict(state=["lazy"], shapes=[(1, 2, 3)], FT=[None], dF=[None], F_real=[None], live=None)
    h = lazy.LazyDraws(rec, 0, (1, 2, 3), torch.device("cpu"))
    h._parts = parts
    with pytest.raises(RuntimeError, match="backward"):
        h.sum()
