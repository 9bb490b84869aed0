"""Deterministic synthetic grids for benchmarking and full-size tests.

Replaces the reference's O(n^3) GP-draw simulator (data/simulated/generate_twod_data.py:17-88) with a
cheap generator of the same shape: a g x g lattice on [0,10]^2 per view, views v > 0 smoothly warped,
outputs = 32-term random Fourier features (RBF, lengthscale 1) + N(0, 0.1^2) noise, column-standardised.
"""
import numpy as np
import torch

from .models import VariationalGPSA


def make_grid_problem(side=100, n_views=2, n_outputs=50, device="cpu", modality="expression", compute_device=None):
    """``compute_device``: evaluate the random-feature sums there with torch (fp64) instead of numpy on the host -
    the same construction and the same random streams, values equal to rounding (cos differs by ulps); for the
    10^5-spot x 1000-output problems, where the host loop takes minutes."""
    lin = np.linspace(0, 10, side)
    x1, x2 = np.meshgrid(lin, lin)
    grid = np.vstack([x1.ravel(), x2.ravel()]).T  # row-major, as generate_twod_data.py:30-35
    rng = np.random.default_rng(1234)
    om = rng.standard_normal((n_outputs, 32, 2))
    a = rng.standard_normal((n_outputs, 32)) / np.sqrt(32.0)
    b = rng.uniform(0, 2 * np.pi, (n_outputs, 32))
    # 64 outputs at a time: the [n, p, 32] feature block of 10^5 spots x 1000 outputs would be 26 GB
    base = np.empty((grid.shape[0], n_outputs))
    for p0 in range(0, n_outputs, 64):
        sl = slice(p0, min(p0 + 64, n_outputs))
        if compute_device is None:
            base[:, sl] = np.einsum("pr,npr->np", a[sl], np.cos(np.einsum("nd,prd->npr", grid, om[sl]) + b[None, sl]))
        else:
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(compute_device)
            ph = torch.einsum("nd,prd->npr", t(grid), t(om[sl])) + t(b[sl])[None]
            base[:, sl] = torch.einsum("pr,npr->np", t(a[sl]), torch.cos(ph)).cpu().numpy()
    nrng = np.random.default_rng(4321)
    Xs, Ys = [], []
    for v in range(n_views):
        Xv = grid.copy()
        if v > 0:
            Xv[:, 0] += 0.3 * np.sin(2 * np.pi * grid[:, 1] / 10 + v)
            Xv[:, 1] += 0.3 * np.cos(2 * np.pi * grid[:, 0] / 10 + v)
        Xs.append(Xv)
        Ys.append(base + 0.1 * nrng.standard_normal(base.shape))
    X = np.concatenate(Xs).astype(np.float32)
    Y = np.concatenate(Ys)
    Y = ((Y - Y.mean(0)) / Y.std(0)).astype(np.float32)
    return {
        modality: {
            "spatial_coords": torch.from_numpy(X).to(device),
            "outputs": torch.from_numpy(Y).to(device),
            "n_samples_list": [side * side] * n_views,
        }
    }


def lattice(m, lo=0.0, hi=10.0):
    """m points on an a x b lattice (a*b == m, a >= b as square as possible) over [lo,hi]^2"""
    b = int(np.floor(np.sqrt(m)))
    while m % b:
        b -= 1
    a = m // b
    g1, g2 = np.meshgrid(np.linspace(lo, hi, a), np.linspace(lo, hi, b))
    return torch.tensor(np.vstack([g1.ravel(), g2.ravel()]).T, dtype=torch.float32)


def make_model(data_dict, m=200, n_latent_gps=None, fixed_view_idx=None, device="cpu", seed=0, **kw):
    """VariationalGPSA with deterministic lattice inducing points (no k-means), torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    cpu_dd = {
        k: {"spatial_coords": v["spatial_coords"].cpu(), "outputs": v["outputs"].cpu(),
            "n_samples_list": v["n_samples_list"]}
        for k, v in data_dict.items()
    }
    model = VariationalGPSA(cpu_dd, m_X_per_view=m, m_G=m, data_init=False, n_latent_gps=n_latent_gps,
                            fixed_view_idx=fixed_view_idx, **kw)
    lat = lattice(m)
    with torch.no_grad():
        model.Xtilde.copy_(lat.unsqueeze(0).expand_as(model.Xtilde))
        model.Gtilde.copy_(lat)
        model.delta_G_list.copy_(model.Xtilde)
    return model.to(device)
