"""GPU: seeded random model configurations (views, spatial dims, unequal view sizes, inducing counts, covariance
kinds, latent mixing, fixed views, S) through the HIP path and through the CPU oracle in fp64, noise drawn here.
Complements the golden cases (tests/test_hip_parity.py), which pin the reference itself."""
import random

import numpy as np
import pytest
import torch

import spatial_alignment_amd as gp
from golden_io import rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KERNELS = {"rbf": gp.rbf_kernel, "matern12": gp.matern12_kernel, "matern32": gp.matern32_kernel}


def _config(seed):
    r = random.Random(seed)
    V, D = r.choice([2, 3]), r.choice([1, 2, 2, 3])
    ns = [r.randint(40, 150) for _ in range(V)]
    P = r.randint(2, 6)
    latent = r.choice([None, None, 2])
    return dict(V=V, D=D, ns=ns, P=P, latent=latent, MX=r.choice([6, 9, 16, 25, 40]), MG=r.choice([6, 9, 16, 25, 40]),
                kw=r.choice(list(KERNELS)), kd=r.choice(list(KERNELS)), fixed=r.choice([None, None, 0, V - 1]),
                S=r.choice([1, 3]))


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_configuration_matches_oracle(seed):
    from oracle import gpsa_oracle as orc

    c = _config(seed)
    m = "expression"
    gen = torch.Generator().manual_seed(1000 + seed)
    N = sum(c["ns"])
    X = torch.rand(N, c["D"], generator=gen) * 10.0
    Y = torch.randn(N, c["P"], generator=gen)
    dd = {m: {"spatial_coords": X, "outputs": Y, "n_samples_list": c["ns"]}}
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = gp.VariationalGPSA(dd, m_X_per_view=c["MX"], m_G=c["MG"], data_init=False,
                               n_latent_gps={m: c["latent"]}, kernel_func_warp=KERNELS[c["kw"]],
                               kernel_func_data=KERNELS[c["kd"]], fixed_view_idx=c["fixed"])
    with torch.no_grad():  # inducing points inside the data's range, moderately conditioned covariances
        model.Xtilde.copy_(torch.rand(model.Xtilde.shape, generator=gen) * 10.0)
        model.Gtilde.copy_(torch.rand(model.Gtilde.shape, generator=gen) * 10.0)
        model.delta_G_list.copy_(model.Xtilde + 0.1 * torch.randn(model.Xtilde.shape, generator=gen))
        model.warp_kernel_lengthscales.fill_(float(np.log(2.0)))
        model.data_kernel_lengthscale.fill_(float(np.log(2.0)))
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for name in ("mean_slopes", "mean_intercepts"):
        state.setdefault(name, getattr(model, name).detach().clone())
    model = model.to(DEV)
    assert model.exact_inducing_grad is None  # the default: exact
    S, L = c["S"], (c["latent"] or c["P"])
    free = [v for v in range(c["V"]) if v != c["fixed"]]
    eps_G = [torch.randn(S, c["ns"][v], c["D"], generator=gen) for v in free]
    eps_F = {m: torch.randn(S, N, L, generator=gen)}
    ddd = {m: {"spatial_coords": X.to(DEV), "outputs": Y.to(DEV), "n_samples_list": c["ns"]}}
    view_idx, Ns, _, _ = model.create_view_idx_dict(ddd)
    model.inject_noise(eps_G, eps_F, None)
    out = model.forward({m: ddd[m]["spatial_coords"]}, view_idx=view_idx, Ns=Ns, S=S)
    loss = model.loss_fn(ddd, out[3])
    loss.backward()
    cfg = dict(modality_names=[m], n_views=c["V"], n_spatial_dims=c["D"], kernel_warp=c["kw"], kernel_data=c["kd"],
               n_latent_gps={m: c["latent"]}, fixed_view_idx=c["fixed"])
    ref = orc.evaluate(state, cfg, {m: X}, {m: Y}, {m: c["ns"]}, S, eps_G, eps_F, dtype=torch.float64)
    assert rel(out[0][m].detach().cpu().numpy(), ref["G_means"][m].numpy()) < 1e-5, c
    assert rel(out[3][m].detach().cpu().numpy(), ref["F_obs"][m].numpy()) < 1e-4, c
    assert rel(loss.detach().cpu().numpy(), ref["loss"].numpy()) < 1e-5, c
    for k, p in model.named_parameters():
        if k in ref["grads"] and p.grad is not None and float(ref["grads"][k].norm()) > 0:
            assert rel(p.grad.cpu().numpy(), ref["grads"][k].numpy()) < 2e-4, (k, c)
