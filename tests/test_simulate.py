"""Simulated-data generator (reference recipes data/simulated/generate_twod_data.py:17-88,
data/warps.py): shapes and conventions, the exact and random-feature GP draws against the covariance they
claim, the three warps against their closed forms.  CPU only, no HIP kernel involved."""
import math

import numpy as np
import pytest
import torch

from spatial_alignment_amd import simulate as sim


def test_lattice_matches_numpy_meshgrid_order():
    lin = np.linspace(0, 10, 7)
    x1, x2 = np.meshgrid(lin, lin)
    want = np.vstack([x1.ravel(), x2.ravel()]).T  # generate_twod_data.py:30-35
    got = sim.lattice_2d(7, dtype=torch.float64).numpy()
    assert np.allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("method,tol", [("exact", 0.08), ("rff", 0.12)])
def test_gp_draws_have_the_rbf_covariance(method, tol):
    x = sim.lattice_2d(5, dtype=torch.float64)  # 25 points
    g = torch.Generator().manual_seed(3)
    f = sim.gp_draws(x, 6000, variance=0.7, lengthscale=3.0, jitter=0.0 if method == "rff" else 1e-6,
                     generator=g, method=method, n_features=4096)
    emp = (f @ f.t()) / f.shape[1]
    K = sim.rbf_covariance(x, x, 0.7, 3.0)
    assert (emp - K).abs().max() < tol, float((emp - K).abs().max())
    assert abs(float(f.mean())) < 0.05


def test_gp_draw_mean_and_auto_switch():
    x = sim.lattice_2d(4, dtype=torch.float64)
    m = torch.arange(16, dtype=torch.float64)
    f = sim.gp_draws(x, 3, variance=1e-12, lengthscale=1.0, mean=m, jitter=0.0, method="rff")
    assert torch.allclose(f, m.unsqueeze(1).expand(-1, 3), atol=1e-4)
    a = sim.gp_draws(x, 2, generator=torch.Generator().manual_seed(1), exact_limit=16)
    b = sim.gp_draws(x, 2, generator=torch.Generator().manual_seed(1), method="exact")
    assert torch.equal(a, b)  # 16 points <= exact_limit: "auto" is the exact draw
    with pytest.raises(ValueError):
        sim.gp_draws(x, 1, method="nystrom")


def test_generate_twod_data_conventions():
    X, Y, nsl, vidx = sim.generate_twod_data(3, 5, 6, noise_variance=0.0, fixed_view_idx=1, seed=7)
    n = 36
    assert X.shape == (3 * n, 2) and Y.shape == (3 * n, 5) and nsl == [n] * 3
    assert [v.tolist() for v in vidx] == [list(range(k * n, (k + 1) * n)) for k in range(3)]
    assert torch.equal(X[vidx[1]], sim.lattice_2d(6))  # the fixed view keeps the lattice
    assert not torch.allclose(X[vidx[0]], sim.lattice_2d(6))
    assert torch.equal(Y[vidx[0]], Y[vidx[2]])  # identical outputs per view before noise
    X2, Y2, _, _ = sim.generate_twod_data(3, 5, 6, noise_variance=0.0, fixed_view_idx=1, seed=7)
    assert torch.equal(X, X2) and torch.equal(Y, Y2)  # seeded
    X3, _, _, _ = sim.generate_twod_data(3, 5, 6, seed=8)
    assert not torch.equal(X, X3)
    _, Yn, _, _ = sim.generate_twod_data(2, 4, 6, noise_variance=0.25, seed=7)
    d = Yn[:n] - Yn[n:]
    assert 0.3 < float(d.std()) < 1.1  # two independent N(0, 0.25) noises: std sqrt(0.5)


def test_latent_mixing_and_data_dict_feed_the_model():
    import spatial_alignment_amd as gp

    X, Y, nsl, _ = sim.generate_twod_data(2, 7, 5, n_latent_gps=3, seed=1)
    assert Y.shape == (50, 7)
    assert np.linalg.matrix_rank(Y[:25].double().numpy(), tol=1e-6) == 3  # 3 latent GPs mixed to 7 outputs
    dd = sim.as_data_dict(X, Y, nsl)
    model = gp.VariationalGPSA(dd, m_X_per_view=9, m_G=9, data_init=False, n_latent_gps={"expression": None})
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    assert Ns["expression"] == 50 and len(view_idx["expression"]) == 2


def test_linear_and_polar_warps_closed_forms():
    Xs = sim.lattice_2d(4, dtype=torch.float64) + 1.0
    Y0 = torch.zeros(16, 2, dtype=torch.float64)
    X, Y, nsl, vidx = sim.apply_linear_warp(Xs, Y0, 3, linear_slope_variance=0.2, linear_intercept_variance=0.3,
                                            noise_variance=0.0, generator=torch.Generator().manual_seed(5))
    for v in vidx:
        s = (X[v] - 0.3) / Xs  # per-coordinate slope, constant over the spots
        assert torch.allclose(s, s[0].expand_as(s), atol=1e-12)
        assert ((s[0] >= 0.8) & (s[0] <= 1.2)).all()
    assert torch.equal(Y, Y0.repeat(3, 1))
    Xp, _, _, vp = sim.apply_polar_warp(Xs, Y0, 2, linear_slope_variance=0.0, noise_variance=0.0)
    assert torch.allclose(Xp[vp[0]], Xs) and torch.allclose(Xp[vp[1]], Xs)  # B = 0: r = 0
    Xq, _, _, vq = sim.apply_polar_warp(Xs, Y0, 1, linear_slope_variance=0.1, noise_variance=0.0,
                                        generator=torch.Generator().manual_seed(2))
    B = (2.0 * torch.rand(2, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(2)) - 1.0) * 0.1
    p = Xs @ B
    want = torch.stack([Xs[:, 0] + p[:, 0] * torch.cos(p[:, 1]), Xs[:, 1] + p[:, 0] * torch.sin(p[:, 1])], 1)
    assert torch.allclose(Xq[vq[0]], want, atol=1e-12)
    with pytest.raises(ValueError):
        sim.apply_polar_warp(torch.zeros(4, 3), torch.zeros(4, 1), 1)


def test_gp_warp_scatters_around_the_mean_function():
    Xs = sim.lattice_2d(6, dtype=torch.float64)
    Y0 = torch.zeros(36, 1, dtype=torch.float64)
    X, _, _, vidx = sim.apply_gp_warp(Xs, Y0, 4, kernel_variance=0.01, kernel_lengthscale=5.0,
                                      mean_slope=2.0, mean_intercept=-1.0,
                                      generator=torch.Generator().manual_seed(0))
    for v in vidx:
        assert float((X[v] - (2.0 * Xs - 1.0)).abs().max()) < 0.6  # sd 0.1 draws around 2 x - 1
    assert not torch.allclose(X[vidx[0]], X[vidx[1]])


def test_semidefinite_covariance_is_sampled_through_its_square_root():
    """lengthscale 5 on a 20 x 20 lattice with no jitter: Cholesky fails in fp64, the draw must not"""
    Xs = sim.lattice_2d(20, dtype=torch.float64)
    f = sim.gp_draws(Xs, 3, variance=0.1, lengthscale=5.0, jitter=0.0, method="exact",
                     generator=torch.Generator().manual_seed(0))
    assert f.shape == (400, 3) and torch.isfinite(f).all()
    assert 0.05 < float(f.std()) < 0.8


def test_large_lattice_uses_features_and_is_fast():
    X, Y, nsl, _ = sim.generate_twod_data(2, 8, 80, seed=0, n_features=256)  # 6400 points per view > 4096
    assert X.shape == (12800, 2) and Y.shape == (12800, 8) and torch.isfinite(Y).all()
    assert 0.5 < float(Y.std()) < 1.5  # unit-variance prior
    assert math.isclose(float(X[:6400].mean()), 5.0, abs_tol=0.5)
