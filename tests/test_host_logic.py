"""Host logic on CPU: model classes + autograd wiring + hand-derived backward formulas of
spatial_alignment_amd/engine.py, run on the TEST-ONLY fake backend (tests/fake_ops.py) and checked
against the reference's fp64 run (golden fixtures).  No HIP kernel is exercised here."""
import numpy as np
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


def _custom_rbf(x1, x2, lengthscale_unconstrained, output_variance_unconstrained, diag=False):
    """a user plug-in (not one of the built-ins by identity): evaluated as-is inside the model"""
    import spatial_alignment_amd as gp

    return gp.rbf_kernel(x1, x2, lengthscale_unconstrained, output_variance_unconstrained, diag)


def test_custom_callable_plugin_matches_builtin():
    from golden_io import rel

    g = Golden("c2_three_free_views")
    model, dd = build_model(g)
    ref = run_step(model, dd, g)
    model2, dd2 = build_model(g)
    model2.kernel_func_warp = _custom_rbf
    model2.kernel_func_data = _custom_rbf
    got = run_step(model2, dd2, g)
    for k, v in ref.items():
        if np.linalg.norm(v) > 0:
            assert rel(got[k], v) < 2e-4, (k, rel(got[k], v))


def test_empty_view_is_skipped_and_stays_nan():
    import spatial_alignment_amd as gp

    torch.manual_seed(0)
    X = torch.rand(30, 2) * 10
    Y = torch.randn(30, 3)
    dd = {"expression": {"spatial_coords": X, "outputs": Y, "n_samples_list": [30, 0]}}
    model = gp.VariationalGPSA(dd, m_X_per_view=5, m_G=5, data_init=False, n_latent_gps={"expression": None})
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    out = model.forward({"expression": X}, view_idx, Ns, S=2)
    assert out[0]["expression"].shape == (30, 2) and torch.isfinite(out[0]["expression"]).all()


def test_constructor_errors_match_reference():
    import spatial_alignment_amd as gp

    X, Y = torch.rand(20, 2), torch.randn(20, 3)
    bad_views = {"a": {"spatial_coords": X, "outputs": Y, "n_samples_list": [10, 10]},
                 "b": {"spatial_coords": X, "outputs": Y, "n_samples_list": [20]}}
    with pytest.raises(ValueError, match="same number of views"):
        gp.VariationalGPSA(bad_views, 4, 4, data_init=False, n_latent_gps={"a": None, "b": None})
    bad_dims = {"a": {"spatial_coords": X, "outputs": Y, "n_samples_list": [10, 10]},
                "b": {"spatial_coords": torch.rand(20, 3), "outputs": Y, "n_samples_list": [10, 10]}}
    with pytest.raises(ValueError, match="spatial dimensions"):
        gp.VariationalGPSA(bad_dims, 4, 4, data_init=False, n_latent_gps={"a": None, "b": None})
    small = {"a": {"spatial_coords": X, "outputs": Y, "n_samples_list": [10, 10]}}
    with pytest.raises(ValueError):  # m_G > spots of the last view (np.random.choice, vgpsa.py:81-85)
        gp.VariationalGPSA(small, 4, 15, data_init=True, n_latent_gps={"a": None})


def test_initial_parameters_match_reference_rng_order():
    """same torch seed => the same freshly initialised parameters as the reference (the constructor
    consumes the RNG in the reference's order); fixture from tests/golden/make_init_golden.py"""
    import os

    import spatial_alignment_amd as gp
    from golden_io import GOLDEN_DIR

    z = np.load(os.path.join(GOLDEN_DIR, "init_state_seed1234.npz"))
    dd = {
        "rna": {"spatial_coords": torch.tensor(z["Xa"]), "outputs": torch.tensor(z["Ya"]), "n_samples_list": [25, 15]},
        "protein": {"spatial_coords": torch.tensor(z["Xb"]), "outputs": torch.tensor(z["Yb"]), "n_samples_list": [10, 20]},
    }
    torch.manual_seed(1234)
    model = gp.VariationalGPSA(dd, m_X_per_view=7, m_G=9, data_init=False, grid_init=False,
                               n_latent_gps={"rna": None, "protein": 2})
    sd = model.state_dict()
    ref = {k[6:]: z[k] for k in z.files if k.startswith("state/")}
    assert set(sd) == set(ref)
    for k, v in ref.items():
        assert np.array_equal(sd[k].numpy(), v), k


def test_fit_loop_matches_manual_steps_and_stops_on_checker():
    """train.fit == the reference's loop body repeated; LossNotDecreasingChecker ends it early"""
    from spatial_alignment_amd.train import fit, train_step

    g = Golden("c1_example_fixed0")
    traces = []
    for use_fit in (True, False):
        model, dd = build_model(g)
        torch.manual_seed(3)
        if use_fit:
            traces.append(fit(model, dd, n_epochs=6, lr=1e-2, S=2, sync_every=4))
        else:
            opt = torch.optim.Adam(model.parameters(), lr=1e-2)
            view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
            traces.append([float(train_step(model, opt, dd, view_idx, Ns, S=2)) for _ in range(6)])
    assert traces[0] == traces[1] and len(traces[0]) == 6
    model, dd = build_model(g)
    from spatial_alignment_amd import LossNotDecreasingChecker

    chk = LossNotDecreasingChecker(max_epochs=40, atol=1e9, window_size=3)  # any decrease is "too small"
    tr = fit(model, dd, n_epochs=40, S=1, sync_every=2, checker=chk)
    assert 3 <= len(tr) < 40


@pytest.mark.parametrize("branch", ["warp2d", "data3d"])
def test_compute_mean_and_var_matches_reference_formula(branch):
    """the public method of vgpsa.py:174-204 (reference arguments and return shapes) against the oracle's
    restatement of it, both branches"""
    from golden_io import rel
    from oracle import gpsa_oracle as orc

    g = Golden("c2_three_free_views")
    model, _ = build_model(g)
    gen = torch.Generator().manual_seed(3)
    f64 = torch.float64
    M, V, D, L, n, S = 12, 3, 2, 5, 40, 2
    Z = 3.0 * torch.randn(M, D, generator=gen, dtype=f64)
    K = orc.rbf_kernel(Z, Z, torch.tensor(1.0, dtype=f64), torch.tensor(0.3, dtype=f64)) + 1e-5 * torch.eye(M, dtype=f64)
    Lk = torch.linalg.cholesky(K)
    kff = torch.exp(torch.tensor(0.3, dtype=f64))
    if branch == "warp2d":
        X = 3.0 * torch.randn(n, D, generator=gen, dtype=f64)
        Kuf = orc.rbf_kernel(Z, X, torch.tensor(1.0, dtype=f64), torch.tensor(0.3, dtype=f64))
        A = 0.3 * torch.randn(V * D, M, M, generator=gen, dtype=f64)
        Lo = torch.linalg.cholesky(A @ A.transpose(-1, -2) + 1e-5 * torch.eye(M, dtype=f64))
        delta, mu_z = torch.randn(V, M, D, generator=gen, dtype=f64), torch.randn(V, M, D, generator=gen, dtype=f64)
        args = (kff, Kuf, Lk, X, mu_z, delta, Lo)
    else:
        X = 3.0 * torch.randn(S, n, D, generator=gen, dtype=f64)
        Kuf = orc.rbf_kernel(Z, X, torch.tensor(1.0, dtype=f64), torch.tensor(0.3, dtype=f64))
        A = 0.3 * torch.randn(L, M, M, generator=gen, dtype=f64)
        Lo = torch.linalg.cholesky(A @ A.transpose(-1, -2) + 1e-5 * torch.eye(M, dtype=f64))
        delta, mu_z = torch.randn(M, L, generator=gen, dtype=f64), torch.zeros(M, L, dtype=f64)
        args = (kff * torch.ones(S, n, dtype=f64), Kuf, Lk, torch.zeros(n, L, dtype=f64), mu_z, delta, Lo)
    want_mean, want_var = orc.conditional(*args)
    got_mean, got_var = model.compute_mean_and_var(*args)
    assert got_mean.shape == want_mean.shape and got_var.shape == want_var.shape
    assert rel(got_mean.numpy(), want_mean.numpy()) < 1e-6 and rel(got_var.numpy(), want_var.numpy()) < 1e-6


def test_empty_free_view_keeps_its_kl_terms():
    """a non-fixed view without rows in this call (a data-parallel rank's empty slice) is skipped by the
    warp GP but keeps its prior factorisation and KL terms: its parameters still receive the KL gradient"""
    import spatial_alignment_amd as gp

    gen = torch.Generator().manual_seed(0)
    X = 10 * torch.rand(30, 2, generator=gen)
    Y = torch.randn(30, 3, generator=gen)
    full = {"expression": {"spatial_coords": X, "outputs": Y, "n_samples_list": [15, 15]}}
    torch.manual_seed(1)
    model = gp.VariationalGPSA(full, m_X_per_view=6, m_G=6, data_init=False, n_latent_gps={"expression": None})
    with torch.no_grad():  # off the initial state delta_G == Xtilde, where the KL's mean term has no gradient
        model.delta_G_list.add_(0.1 * torch.randn(model.delta_G_list.shape, generator=gen))
    part = {"expression": {"spatial_coords": X[:15], "outputs": Y[:15], "n_samples_list": [15, 0]}}
    view_idx, Ns, _, _ = model.create_view_idx_dict(part)
    out = model.forward({"expression": X[:15]}, view_idx, Ns, S=2)
    assert out[1]["expression"].shape == (2, 15, 2)
    loss = model.loss_fn(part, out[3])
    loss.backward()
    assert torch.isfinite(loss)
    assert model.Xtilde.grad[1].abs().max() > 0 and model.delta_G_list.grad[1].abs().max() > 0
    assert model.Omega_sqt_G_list.grad.abs().sum((-1, -2)).min() > 0


@pytest.mark.parametrize("name", ["c1_example_fixed0", "c5_two_modalities"])
def test_handoff_attributes_match_reference(name):
    """Kuu_chol_list / curr_Omega_tril_list / Kuu_chol_F / curr_Omega_tril_F exist after forward"""
    from model_util import _check_handoff, _handoff_reference

    g = Golden(name)
    model, dd = build_model(g)
    with pytest.raises(AttributeError):
        model.Kuu_chol_F
    run_step(model, dd, g)
    _check_handoff(model, _handoff_reference(g))


def test_reference_import_lines_resolve():
    """every ``from gpsa... import ...`` line found in the reference's examples / experiments (grep over
    /root/reference, listed here as data) executes against the alias package: the import block of
    examples/grid_example.py:6-9 runs unchanged.  The four plotting callbacks are names only (out of scope)."""
    lines = [
        "from gpsa import VariationalGPSA",
        "from gpsa import matern12_kernel, rbf_kernel",
        "from gpsa.plotting import callback_twod",
        "from gpsa import VariationalGPSA, matern12_kernel, rbf_kernel, LossNotDecreasingChecker",
        "from gpsa import GPSA",
        "from gpsa import polar_warp",
        "from gpsa.util import rbf_kernel_numpy as rbf_covariance",
        "from gpsa.util.util import rbf_kernel_numpy",
        "from gpsa.util.util import rbf_kernel, matern12_kernel, matern32_kernel, polar_warp, get_st_coordinates, "
        "LossNotDecreasingChecker",
        "from gpsa.plotting import callback_oned, callback_twod, callback_twod_aligned_only",
        "from gpsa.plotting import callback_twod, callback_twod_multimodal",
        "from gpsa.plotting.callbacks import callback_oned, callback_twod, callback_twod_aligned_only, "
        "callback_twod_multimodal",
        "from gpsa.models.vgpsa import VariationalGPSA",
        "from gpsa.models.gpsa import GPSA",
        "from gpsa import callback_twod",
    ]
    ns = {}
    for line in lines:
        exec(line, ns)
    import spatial_alignment_amd as pkg

    assert ns["VariationalGPSA"] is pkg.VariationalGPSA and ns["GPSA"] is pkg.GPSA
    assert ns["rbf_kernel"] is pkg.rbf_kernel  # the plug-in identity the fused path recognises
    with pytest.raises(NotImplementedError, match="out of scope"):
        ns["callback_twod"](None, None, None)


def test_kl_owner_ranges_partition_the_terms():
    """parallel.own_kl_terms / step_engine.kl_own_range: over the ranks of any world the ranges are contiguous, disjoint
    and cover the V*D + sum L terms; a world of one (or no owner) means every term"""
    import types

    from spatial_alignment_amd import step_engine as SE

    for V, D, Ls in ((2, 2, [50]), (3, 1, [5, 7]), (8, 2, [2000]), (2, 2, [1])):
        n = V * D + sum(Ls)
        m = types.SimpleNamespace(n_views=V, n_spatial_dims=D, modality_names=[f"m{i}" for i in range(len(Ls))],
                                  n_latent_outputs={f"m{i}": L for i, L in enumerate(Ls)}, kl_owner=None)
        assert SE.kl_own_range(m) is None
        m.kl_owner = (0, 1)
        assert SE.kl_own_range(m) is None
        for world in (2, 3, 8, n, n + 3):
            edge, sizes = 0, []
            for r in range(world):
                m.kl_owner = (r, world)
                lo, hi = SE.kl_own_range(m)
                if hi > lo and (lo, hi) != (n, n):
                    assert lo == edge
                    edge = hi
                    sizes.append(hi - lo)
                else:  # more ranks than terms: the ranks left over own the empty range [n, n)
                    assert (lo, hi) == (n, n) and hi > 0
            assert edge == n and max(sizes) - min(sizes) <= 1


def test_bench_smi_sample_parsing(tmp_path):
    """bench.py's ``sustained`` leg: the rocm-smi sampler's lines inside the run's window -> clock and power statistics"""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = tmp_path / "samples.tsv"
    line = ("GPU[0]\t\t: sclk clock level: 1: (%dMhz) | ===== Power Consumption ===== | "
            "GPU[0]\t\t: Current Socket Graphics Package Power (W): %.1f")
    out.write_text("".join("%.3f\t%s\n" % (t, line % (c, p)) for t, c, p in
                           ((5.0, 2100, 300.0), (10.2, 2390, 990.0), (10.7, 2394, 1010.0), (20.0, 2000, 200.0))))

    class P:
        def wait(self, timeout=None):
            return 0

    res = bench.stop_smi_sampler(dict(proc=P(), out=str(out), stop=str(tmp_path / "stop")), 10.0, 11.0)
    assert res["samples"] == 2
    assert res["sclk_mhz"] == dict(min=2390, mean=2392.0, max=2394)
    assert res["power_w"] == dict(min=990.0, mean=1000.0, max=1010.0)
