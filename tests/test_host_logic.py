"""Host logic on CPU: model classes + autograd wiring + hand-derived backward formulas of
spatial_alignment_amd/engine.py, run on the TEST-ONLY fake backend (tests/fake_ops.py) and checked
against the reference's fp64 run (golden fixtures).  No HIP kernel is exercised here."""
import pytest
import torch

from fake_ops import FakeOps
from golden_io import SMALL_CASES, Golden
from model_util import build_model, compare, run_step
from spatial_alignment_amd import ops as ops_mod


@pytest.fixture(autouse=True)
def fake_backend():
    ops_mod.set_ops(FakeOps())
    yield
    ops_mod.set_ops(None)


@pytest.mark.parametrize("name", SMALL_CASES)
def test_forward_backward_vs_reference_fp64(name):
    g = Golden(name)
    model, dd = build_model(g)
    res = run_step(model, dd, g)
    bad, errs = compare(res, g, tol_out=1e-4, tol_grad=2e-3)
    assert not bad, bad


def test_state_dict_names_match_reference():
    g = Golden("c5_two_modalities")
    model, _ = build_model(g)
    assert set(model.state_dict().keys()) == set(g.state.keys())


def test_loss_before_forward_raises():
    g = Golden("c2_three_free_views")
    model, dd = build_model(g)
    with pytest.raises(AttributeError):
        model.loss_fn(dd, {})
