"""Simulated alignment problems at scale: a lattice, per-view warps of it, outputs drawn
from a GP over the unwarped lattice.

Same recipes and return convention as the reference's simulators — ``generate_twod_data``
(data/simulated/generate_twod_data.py:17-88) and the GP / linear / polar warps (data/warps.py:17-70,
160-233, 236-304) — but written for the device the model trains on: everything is torch on ``device``
with an explicit ``torch.Generator`` (the reference draws from numpy's global state), and a GP draw is

* exact (``K + jitter I = L L^T``, ``f = m + L z``, fp64) while the lattice is small enough for the O(n^3)
  factorisation (``n <= exact_limit``), which is what the reference does through
  ``scipy.stats.multivariate_normal.rvs`` at any size, and
* a random-Fourier-feature draw of the same RBF prior beyond it (``f = m + sqrt(2 var / R) sum_r a_r
  cos(w_r.x + b_r)``, ``w ~ N(0, 1/ell^2)``, ``b ~ U[0, 2 pi)``, ``a ~ N(0,1)``): O(n R) time and memory, so
  the 100 x 100 and 316 x 316 lattices of BASELINE.json's configurations take milliseconds.

On a HIP device the exact draw runs on this package's own kernels (``_exact_draw_hip``: ``gpsa_kmat`` +
``gpsa_chol_inv*_f64`` + ``gpsa_gemm``); CPU tensors take torch's.  Returns ``(X [n_views * n, D], Y [n_views * n, P],
n_samples_list, view_idx)`` like the reference, as torch tensors.
"""
import math

import torch

__all__ = [
    "rbf_covariance",
    "gp_draws",
    "lattice_2d",
    "apply_gp_warp",
    "apply_linear_warp",
    "apply_polar_warp",
    "generate_twod_data",
]


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def rbf_covariance(x, xp, variance=1.0, lengthscale=1.0):
    """variance * exp(-1/2 |x - x'|^2 / lengthscale^2): gpsa/util/util.py:26-30 with
    ``kernel_params = [log variance, log lengthscale]``."""
    d2 = torch.cdist(x / lengthscale, xp / lengthscale).square()
    return variance * torch.exp(-0.5 * d2)


def _exact_draw_hip(x64, z, variance, lengthscale, jitter):
    """f = L z with K + jitter I = L L^T through THIS package's HIP kernels (data/warps.py:55-65 and
    generate_twod_data.py:49-61 do it with scipy): the covariance from ``gpsa_kmat`` (the fused RBF kernel of the
    hot path, fp64), the factor's inverse from ``gpsa_chol_inv_f64`` / ``gpsa_chol_inv_blocked_f64``, and
    L z = K (L^-T z) as two ``gpsa_gemm`` products.  A smooth kernel without jitter is numerically semi-definite
    (the GP warp asks for jitter 0): the factorisation then flags a non-positive pivot and the draw is retried with
    a floor of 1e-8 * variance on the diagonal (1e-4 of the prior standard deviation: far below the lattice
    spacing).  Returns None when even 1e-4 of the variance is flagged (the caller raises)."""
    from . import ops as _ops

    o = _ops.get_ops()
    dev = x64.device
    ls_u = torch.full((1,), math.log(float(lengthscale)), dtype=torch.float64, device=dev)
    var_u = torch.full((1,), math.log(float(variance)), dtype=torch.float64, device=dev)
    # the jitter is raised in decades up to 1e-4 of the variance before giving up (round 6: a smooth kernel on a dense
    # lattice is numerically semi-definite, and the fallback for that used to be torch.linalg.cholesky_ex / eigh -
    # rocSOLVER - on the device; a draw from N(0, K + 1e-4 s^2 I) instead of N(0, K) is white noise 40 dB under the
    # signal, which the simulator adds by the percent anyway: generate_twod_data.py:77)
    floor = float(variance)
    for jit in (float(jitter),) + tuple(max(float(jitter), f * floor) for f in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4)):
        K = o.kmat("rbf", x64, x64, ls_u, var_u, jitter=jit, dtype=torch.float64)
        Linv, _, info = o.chol_inv(K.unsqueeze(0))
        if int(info.item()) == 0:
            t = o.gemm(Linv[0], z, transA=True)  # L^-T z
            return o.gemm(K, t)                 # K L^-T z = L z
    return None


def gp_draws(x, n_draws, variance=1.0, lengthscale=1.0, mean=None, jitter=1e-3, generator=None,
             method="auto", exact_limit=4096, n_features=2048):
    """``n_draws`` independent draws f ~ GP(mean, RBF(variance, lengthscale)) at the rows of ``x`` [n, D].
    Returns [n, n_draws] in x's dtype.  ``mean``: None, [n] or [n, n_draws].  ``method``: "exact", "rff"
    or "auto" (exact up to ``exact_limit`` points)."""
    n, d = x.shape
    dev = x.device
    if generator is None:
        generator = _gen(0, dev)
    if method == "auto":
        method = "exact" if n <= exact_limit else "rff"
    f64 = torch.float64
    if method == "exact":
        x64 = x.to(f64)
        z = torch.randn(n, n_draws, dtype=f64, device=dev, generator=generator)
        f = _exact_draw_hip(x64, z, variance, lengthscale, jitter) if x.is_cuda else None
        if f is None and x.is_cuda:  # never a vendor solver on the device: the caller picks method="rff" or more jitter
            raise torch.linalg.LinAlgError(
                "gp_draws(method='exact'): the covariance is not positive definite even with a jitter of 1e-4 of the "
                "variance (this package's fp64 factorisation); use method='rff' or a larger jitter")
        if f is None:  # CPU tensors: plain torch
            K = rbf_covariance(x64, x64, variance, lengthscale)
            K.diagonal().add_(jitter)
            L, info = torch.linalg.cholesky_ex(K)
            if int(info) != 0:  # smooth kernel, no jitter: numerically semi-definite -> symmetric square root
                lam, V = torch.linalg.eigh(K)  # (scipy's multivariate_normal.rvs factors through the SVD as well)
                L = V * lam.clamp_min(0.0).sqrt()
            f = L @ z
    elif method == "rff":
        R = int(n_features)
        w = torch.randn(d, R, dtype=f64, device=dev, generator=generator) / lengthscale
        b = torch.rand(R, dtype=f64, device=dev, generator=generator) * (2.0 * math.pi)
        a = torch.randn(R, n_draws, dtype=f64, device=dev, generator=generator)
        f = math.sqrt(2.0 * variance / R) * (torch.cos(x.to(f64) @ w + b) @ a)
        if jitter:
            f = f + math.sqrt(jitter) * torch.randn(n, n_draws, dtype=f64, device=dev, generator=generator)
    else:
        raise ValueError(f"unknown method {method!r}")
    if mean is not None:
        m = mean.to(f64)
        f = f + (m.unsqueeze(1) if m.dim() == 1 else m)
    return f.to(x.dtype)


def lattice_2d(grid_size, lo=0.0, hi=10.0, device="cpu", dtype=torch.float32):
    """grid_size x grid_size lattice on [lo, hi]^2, row-major meshgrid order
    (generate_twod_data.py:30-35)."""
    lin = torch.linspace(lo, hi, grid_size, dtype=torch.float64, device=device)
    x2, x1 = torch.meshgrid(lin, lin, indexing="ij")  # np.meshgrid(x1s, x2s) + ravel: x1 runs fastest
    return torch.stack([x1.reshape(-1), x2.reshape(-1)], 1).to(dtype)


def _views(n, n_views, device):
    n_samples_list = [n] * n_views
    view_idx = [torch.arange(v * n, (v + 1) * n, device=device) for v in range(n_views)]
    return n_samples_list, view_idx


def _outputs(Y_single, n_views, noise_variance, generator):
    Y = Y_single.repeat(n_views, 1)
    if noise_variance:
        Y = Y + math.sqrt(noise_variance) * torch.randn(Y.shape, dtype=Y.dtype, device=Y.device,
                                                         generator=generator)
    return Y


def apply_gp_warp(X_single, Y_single, n_views, noise_variance=0.0, kernel_variance=1.0,
                  kernel_lengthscale=1.0, mean_slope=1.0, mean_intercept=0.0, generator=None, **gp_kw):
    """Every coordinate of every view is one GP draw around ``mean_slope * x + mean_intercept``
    (data/warps.py:17-70).  Returns X [n_views * n, D], Y, n_samples_list, view_idx."""
    n, D = X_single.shape
    generator = generator or _gen(0, X_single.device)
    mean = (X_single * mean_slope + mean_intercept).repeat(1, n_views)  # column v * D + s
    W = gp_draws(X_single, n_views * D, kernel_variance, kernel_lengthscale, mean=mean, jitter=0.0,
                 generator=generator, **gp_kw)
    X = torch.cat([W[:, v * D:(v + 1) * D] for v in range(n_views)], 0)
    n_samples_list, view_idx = _views(n, n_views, X.device)
    return X, _outputs(Y_single, n_views, noise_variance, generator), n_samples_list, view_idx


def apply_linear_warp(X_single, Y_single, n_views, linear_slope_variance=0.1, linear_intercept_variance=0.1,
                      noise_variance=0.01, generator=None):
    """Per view: x -> x * s + c with s ~ U[1 - slope_var, 1 + slope_var] per coordinate and the constant
    c = intercept_var (the reference draws c from U[v, v], data/warps.py:214-222)."""
    n, D = X_single.shape
    dev = X_single.device
    generator = generator or _gen(0, dev)
    parts = []
    for _ in range(n_views):
        u = torch.rand(D, dtype=torch.float64, device=dev, generator=generator)
        slopes = (1.0 - linear_slope_variance) + 2.0 * linear_slope_variance * u
        parts.append((X_single.double() * slopes + linear_intercept_variance).to(X_single.dtype))
    n_samples_list, view_idx = _views(n, n_views, dev)
    return torch.cat(parts, 0), _outputs(Y_single, n_views, noise_variance, generator), n_samples_list, view_idx


def apply_polar_warp(X_single, Y_single, n_views, linear_slope_variance=0.1, linear_intercept_variance=0.1,
                     noise_variance=0.01, generator=None):
    """Per view: (r, theta) = x B with B ~ U[-slope_var, slope_var]^{2x2}, then
    x -> x + r (cos theta, sin theta)   (data/warps.py:270-287, gpsa/util/util.py:69-70)."""
    n, D = X_single.shape
    if D != 2:
        raise ValueError("the polar warp is defined for 2 spatial dimensions")
    dev = X_single.device
    generator = generator or _gen(0, dev)
    parts = []
    x = X_single.double()
    for _ in range(n_views):
        B = (2.0 * torch.rand(2, 2, dtype=torch.float64, device=dev, generator=generator) - 1.0) * linear_slope_variance
        p = x @ B
        r, th = p[:, 0], p[:, 1]
        parts.append(torch.stack([x[:, 0] + r * torch.cos(th), x[:, 1] + r * torch.sin(th)], 1).to(X_single.dtype))
    n_samples_list, view_idx = _views(n, n_views, dev)
    return torch.cat(parts, 0), _outputs(Y_single, n_views, noise_variance, generator), n_samples_list, view_idx


def generate_twod_data(n_views, n_outputs, grid_size, n_latent_gps=None, kernel_variance=0.1,
                       kernel_lengthscale=5, noise_variance=0.0, fixed_view_idx=None, warp="gp",
                       device="cpu", dtype=torch.float32, seed=0, **gp_kw):
    """A grid_size x grid_size lattice on [0, 10]^2 seen in ``n_views`` warped copies; the outputs are
    ``n_outputs`` draws of a unit RBF GP over the unwarped lattice (or ``n_latent_gps`` draws mixed by a
    standard-normal W), identical in every view before noise (generate_twod_data.py:17-88).

    ``warp``: "gp" (the reference's choice), "linear" or "polar"; ``fixed_view_idx``: that view keeps the
    unwarped lattice.  Unlike the reference — which always warps exactly two views whatever ``n_views``
    says (generate_twod_data.py:76-83) — all ``n_views`` views are produced."""
    g = _gen(seed, device)
    Xs = lattice_2d(grid_size, device=device, dtype=dtype)
    nY = n_outputs if n_latent_gps is None else n_latent_gps
    Y0 = gp_draws(Xs, nY, 1.0, 1.0, generator=g, **gp_kw)
    if n_latent_gps is not None:
        Wm = torch.randn(n_latent_gps, n_outputs, dtype=torch.float64, device=device, generator=g)
        Y0 = (Y0.double() @ Wm).to(dtype)
    if warp == "gp":
        X, Y, nsl, vidx = apply_gp_warp(Xs, Y0, n_views, noise_variance=noise_variance,
                                        kernel_variance=kernel_variance, kernel_lengthscale=kernel_lengthscale,
                                        generator=g, **gp_kw)
    elif warp == "linear":
        X, Y, nsl, vidx = apply_linear_warp(Xs, Y0, n_views, noise_variance=noise_variance, generator=g)
    elif warp == "polar":
        X, Y, nsl, vidx = apply_polar_warp(Xs, Y0, n_views, noise_variance=noise_variance, generator=g)
    else:
        raise ValueError(f"unknown warp {warp!r}")
    if fixed_view_idx is not None:
        X[vidx[fixed_view_idx]] = Xs
    return X, Y, nsl, vidx


def as_data_dict(X, Y, n_samples_list, modality="expression"):
    """the ``data_dict`` the model classes take (gpsa/models/gpsa.py:12, examples/grid_example.py:34-40)"""
    return {modality: {"spatial_coords": X, "outputs": Y, "n_samples_list": list(n_samples_list)}}
