"""Inducing-point initialisation on the device.

``kmeans(X, K)``: Lloyd iterations on HIP kernels (csrc/kmeans.hip) from a seeded random subset of the
points.  Used by ``VariationalGPSA(data_init=True)`` when the coordinates are HIP tensors; CPU tensors
keep the reference's scikit-learn path (gpsa/models/vgpsa.py:61-92).
"""
import torch

from . import ops as _ops_mod


def kmeans(X, K, iters=25, seed=0):
    """X [N,D] fp32 device tensor -> centres [K,D] fp32 (same device).  Deterministic given ``seed``."""
    o = _ops_mod.get_ops()
    X = X.detach().float().contiguous()
    N = X.shape[0]
    if K > N:
        raise ValueError(f"Cannot take a larger sample than population: K={K} > N={N}")
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    idx = torch.randperm(N, generator=g)[:K].to(X.device)
    centres = X[idx].clone().contiguous()
    for _ in range(iters):
        assign, _ = o.kmeans_assign(X, centres)
        o.kmeans_update(X, assign, centres)
    return centres
