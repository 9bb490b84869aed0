"""Deterministic input recipes shared by make_golden.py (build container) and the tests.

Pure numpy (``default_rng`` streams are stable across numpy versions); nothing here
touches the reference.
"""
import numpy as np

SLICE_STRIDE = 37  # stride of the flattened slices kept for "summary_only" fixtures


def _rff(X, n_out, rng, n_terms=16, ell=2.0):
    D = X.shape[1]
    Y = np.zeros((X.shape[0], n_out))
    for p in range(n_out):
        w = rng.standard_normal((n_terms, D)) / ell
        b = rng.uniform(0, 2 * np.pi, n_terms)
        a = rng.standard_normal(n_terms) / np.sqrt(n_terms)
        Y[:, p] = np.cos(X @ w.T + b) @ a
    return Y


def grid_views(side, n_views, n_out, seed, dims=2):
    """``n_views`` copies of a ``side**dims`` lattice on [0,10]^dims; views v>0 smoothly warped."""
    rng = np.random.default_rng(seed)
    axes = [np.linspace(0, 10, side)] * dims
    grid = np.stack([g.ravel() for g in np.meshgrid(*axes)], axis=1)
    Xs, Ys = [], []
    base = _rff(grid, n_out, rng)
    for v in range(n_views):
        Xv = grid.copy()
        if v > 0:
            for d in range(dims):
                Xv[:, d] += 0.3 * np.sin(2 * np.pi * grid[:, (d + 1) % dims] / 10 + v + d)
            Xv += 0.02 * rng.standard_normal(Xv.shape)
        Xs.append(Xv)
        Ys.append(base + 0.1 * rng.standard_normal(base.shape))
    X = np.concatenate(Xs).astype(np.float32)
    Y = np.concatenate(Ys)
    Y = ((Y - Y.mean(0)) / Y.std(0)).astype(np.float32)
    return X, Y, [grid.shape[0]] * n_views


def line_views(n, n_views, n_out, seed):
    rng = np.random.default_rng(seed)
    x = np.linspace(0, 10, n)[:, None]
    base = _rff(x, n_out, rng)
    Xs, Ys = [], []
    for v in range(n_views):
        xv = x + (0.4 * np.sin(x * 0.7 + v) if v > 0 else 0.0)
        Xs.append(xv)
        Ys.append(base + 0.1 * rng.standard_normal(base.shape))
    X = np.concatenate(Xs).astype(np.float32)
    Y = np.concatenate(Ys)
    Y = ((Y - Y.mean(0)) / Y.std(0)).astype(np.float32)
    return X, Y, [n] * n_views


def m200_state(X, n_out, n_views, dims, m, seed):
    """Full parameter set (reference state_dict names) for the M=200 conditioning case."""
    rng = np.random.default_rng(seed)
    n_v = X.shape[0] // n_views
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    Xt = np.stack(
        [X[v * n_v : (v + 1) * n_v][rng.choice(n_v, m, replace=False)] for v in range(n_views)]
    )
    Xt = Xt + 0.05 * rng.standard_normal(Xt.shape)
    Gt = X[rng.choice(X.shape[0], m, replace=False)] + 0.05 * rng.standard_normal((m, dims))
    return {
        "noise_variance": f32([-0.7, -1.3]),
        "warp_kernel_variances": f32(np.zeros(n_views)),
        "warp_kernel_lengthscales": f32(np.full(n_views, np.log(10.0))),
        "data_kernel_lengthscale": f32([0.3]),
        "data_kernel_variance": f32([0.2]),
        "Xtilde": f32(Xt),
        "Gtilde": f32(Gt),
        "Omega_sqt_G_list": f32(0.1 * rng.standard_normal((n_views * dims, m, m))),
        "delta_G_list": f32(Xt + 0.1 * rng.standard_normal(Xt.shape)),
        "Omega_sqt_F_dict.expression": f32(0.1 * rng.standard_normal((n_out, m, m))),
        "delta_F_dict.expression": f32(rng.standard_normal((m, n_out))),
    }
