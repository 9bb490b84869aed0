"""Covariance-function plugins (the ``kernel_func_warp`` / ``kernel_func_data`` API).

Same call signature as the reference's gpsa/util/util.py:8-66:

    k(x1, x2, lengthscale_unconstrained, output_variance_unconstrained, diag=False) -> Tensor

When one of the three built-ins below is handed to ``VariationalGPSA`` it is recognised by identity
and the model runs the fused HIP covariance kernels (csrc/kmat.hip) instead of calling it; called
directly by user code (e.g. for plotting) they evaluate with ordinary tensor ops on whatever device
the inputs live on.  Any other callable with this signature is evaluated as-is inside the model and
its matrices are fed to the HIP layer kernels.
"""
import math

import torch


def _diffs(x1, x2, diag):
    return x1 - x2 if diag else x1.unsqueeze(-2) - x2.unsqueeze(-3)


def rbf_kernel(x1, x2, lengthscale_unconstrained, output_variance_unconstrained, diag=False):
    """sigma^2 exp(-1/2 |x1-x2|^2 / ell^2)  (util.py:8-23)."""
    ell = torch.exp(lengthscale_unconstrained)
    var = torch.exp(output_variance_unconstrained)
    u = _diffs(x1, x2, diag) / ell
    return var * torch.exp(-0.5 * (u * u).sum(-1))


def matern12_kernel(x1, x2, lengthscale_unconstrained, output_variance_unconstrained, diag=False):
    """sigma^2 exp(-0.5 d / ell), d = sqrt(|x1-x2|^2 + 1e-10)  (util.py:33-47; note the 0.5)."""
    ell = torch.exp(lengthscale_unconstrained)
    var = torch.exp(output_variance_unconstrained)
    u = _diffs(x1, x2, diag)
    d = torch.sqrt((u * u).sum(-1) + 1e-10)
    return var * torch.exp(-0.5 * d / ell)


def matern32_kernel(x1, x2, lengthscale_unconstrained, output_variance_unconstrained, diag=False):
    """sigma^2 (1 + sqrt3 d/ell) exp(-sqrt3 d/ell)  (util.py:50-66)."""
    ell = torch.exp(lengthscale_unconstrained)
    var = torch.exp(output_variance_unconstrained)
    u = _diffs(x1, x2, diag)
    d = torch.sqrt((u * u).sum(-1) + 1e-10)
    z = math.sqrt(3.0) * d / ell
    return var * (1.0 + z) * torch.exp(-z)


BUILTIN_KIND = {rbf_kernel: "rbf", matern12_kernel: "matern12", matern32_kernel: "matern32"}


def builtin_kind(fn):
    """'rbf' | 'matern12' | 'matern32' for the built-ins (by identity), else None."""
    try:
        return BUILTIN_KIND.get(fn)
    except TypeError:  # unhashable callable
        return None
