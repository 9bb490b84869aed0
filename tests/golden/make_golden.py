"""Generate golden fixtures by RUNNING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the unmodified reference package from /root/reference (with a stub for the
absent plotting dependency ``seaborn``), runs ``VariationalGPSA.forward`` +
``loss_fn`` + ``backward`` in fp32 and — same parameters, same noise — in fp64
(``torch.set_default_dtype(torch.float64)``; SURVEY.md §8c "fp64 arbiter"), and
stores inputs, parameters, the recorded Gaussian noise, outputs, loss and all
parameter gradients as ``tests/golden/<case>.npz``.

Only DATA is written (arrays + a small JSON config); no reference source travels.
The GPU box never runs this script: tests there read the committed ``.npz`` files.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipes  # noqa: E402  (shared deterministic input recipes)

warnings.filterwarnings("ignore")
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
sys.path.insert(0, "/root/reference")
import gpsa  # noqa: E402
from gpsa import VariationalGPSA  # noqa: E402

torch.autograd.set_detect_anomaly(False)  # reference turns it on at import (vgpsa.py:9)

KERNELS = {
    "rbf": gpsa.rbf_kernel,
    "matern12": gpsa.matern12_kernel,
    "matern32": gpsa.matern32_kernel,
}


class NoiseTap:
    """Records (mode='rec') or replays (mode='play') every Gaussian draw of forward()."""

    def __init__(self):
        self.mode, self.tape, self.pos = "off", [], 0
        self._sn = torch.distributions.normal._standard_normal
        self._randn = torch.randn
        tap = self

        def standard_normal(shape, dtype, device):
            return tap._draw(lambda: tap._sn(shape, dtype, device), ("G", tuple(shape)), dtype)

        def randn(*a, **k):
            if tap.mode == "off":
                return tap._randn(*a, **k)
            shape = tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)
            return tap._draw(lambda: tap._randn(*a, **k), ("F", shape), torch.get_default_dtype())

        torch.distributions.normal._standard_normal = standard_normal
        torch.randn = randn

    def _draw(self, fresh, tag, dtype):
        if self.mode == "rec":
            t = fresh()
            self.tape.append((tag, t.detach().clone()))
            return t
        if self.mode == "play":
            want, t = self.tape[self.pos]
            assert want == tag, (want, tag)
            self.pos += 1
            return t.to(dtype).clone()
        return fresh()


TAP = NoiseTap()


def build(case, dtype):
    torch.set_default_dtype(dtype)
    mods = case["mods"]
    dd = {}
    for m in mods:
        dd[m] = {
            "spatial_coords": torch.tensor(case["X"][m], dtype=dtype),
            "outputs": torch.tensor(case["Y"][m], dtype=dtype),
            "n_samples_list": list(case["n_samples"][m]),
        }
    np.random.seed(case["seed"])
    torch.manual_seed(case["seed"])
    model = VariationalGPSA(
        dd,
        m_X_per_view=case["m_X"],
        m_G=case["m_G"],
        data_init=True,
        n_latent_gps=case["n_latent_gps"],
        kernel_func_warp=KERNELS[case["kernel_warp"]],
        kernel_func_data=KERNELS[case["kernel_data"]],
        fixed_view_idx=case["fixed_view_idx"],
        fixed_warp_kernel_variances=case.get("fixed_warp_kernel_variances"),
        fixed_warp_kernel_lengthscales=case.get("fixed_warp_kernel_lengthscales"),
        fixed_data_kernel_lengthscales=case.get("fixed_data_kernel_lengthscales"),
    )
    return model, dd


def step(model, dd, case, dtype):
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: dd[m]["spatial_coords"] for m in case["mods"]}
    G_test = None
    if case.get("G_test") is not None:
        G_test = {m: torch.tensor(g, dtype=dtype) for m, g in case["G_test"].items()}
    model.zero_grad()
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=case["S"], G_test=G_test)
    loss = model.loss_fn(dd, out[3])
    loss.backward()
    res = {"loss": loss.detach().numpy()}
    names = ["G_means", "G_samples", "F_latent", "F_obs", "F_latent_test", "F_obs_test"]
    for nm, o in zip(names, out):
        for m in case["mods"]:
            res[f"{nm}/{m}"] = o[m].detach().numpy()
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        res[f"grad/{k}"] = g.detach().numpy()
    return res


def run_case(name, case):
    print("case", name)
    # fp32: build, perturb away from the trivial initial state, record
    model, dd = build(case, torch.float32)
    if case.get("state_override") is not None:
        sd = model.state_dict()
        for k, v in case["state_override"].items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
            sd[k] = torch.tensor(v, dtype=torch.float32)
        model.load_state_dict(sd)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=3e-2)
        for _ in range(case.get("pre_steps", 3)):
            view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
            o = model.forward({m: dd[m]["spatial_coords"] for m in case["mods"]}, view_idx, Ns, S=2)
            l = model.loss_fn(dd, o[3])
            opt.zero_grad()
            l.backward()
            opt.step()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    TAP.mode, TAP.tape, TAP.pos = "rec", [], 0
    r32 = step(model, dd, case, torch.float32)
    TAP.mode = "off"
    tape = TAP.tape

    # fp64: same parameters, same noise, unmodified reference under float64 default
    model64, dd64 = build(case, torch.float64)
    model64.load_state_dict({k: v.double() for k, v in state.items()})
    for attr in ("warp_kernel_variances", "warp_kernel_lengthscales", "data_kernel_lengthscale"):
        t = getattr(model, attr)
        if not isinstance(t, torch.nn.Parameter):
            setattr(model64, attr, t.double())
    TAP.mode, TAP.pos = "play", 0
    r64 = step(model64, dd64, case, torch.float64)
    assert TAP.pos == len(tape)
    TAP.mode = "off"
    torch.set_default_dtype(torch.float32)

    eps_G = [t.numpy() for (tag, t) in tape if tag[0] == "G"]
    eps_F = [t.numpy() for (tag, t) in tape if tag[0] == "F"]
    S = case["S"]
    n_free = len(eps_G) // S if S else 0
    arrays = {}
    for v in range(n_free):  # S consecutive draws per non-fixed view
        arrays[f"eps_G/{v}"] = np.stack(eps_G[v * S : (v + 1) * S])
    i = 0
    for m in case["mods"]:
        arrays[f"eps_F/{m}"] = eps_F[i]
        i += 1
        if case.get("G_test") is not None:
            arrays[f"eps_F_test/{m}"] = eps_F[i]
            i += 1
    assert i == len(eps_F)

    big = case.get("summary_only", False)
    for m in case["mods"]:
        if not big or True:
            arrays[f"in/X/{m}"] = np.asarray(case["X"][m], dtype=np.float32)
            arrays[f"in/Y/{m}"] = np.asarray(case["Y"][m], dtype=np.float32)
        if case.get("G_test") is not None:
            arrays[f"in/G_test/{m}"] = np.asarray(case["G_test"][m], dtype=np.float32)
    for k, v in state.items():
        if big and v.numel() > 20000:
            continue  # regenerated from recipes.py in the test
        arrays[f"state/{k}"] = v.numpy()
    for attr in ("warp_kernel_variances", "warp_kernel_lengthscales", "data_kernel_lengthscale"):
        t = getattr(model, attr)
        if not isinstance(t, torch.nn.Parameter):
            arrays[f"fixed/{attr}"] = t.detach().numpy()
    for tag, r in (("ref32", r32), ("ref64", r64)):
        for k, v in r.items():
            if big and v.size > 20000:
                arrays[f"{tag}/norm/{k}"] = np.array(np.linalg.norm(v.astype(np.float64)))
                arrays[f"{tag}/slice/{k}"] = v.reshape(-1)[:: recipes.SLICE_STRIDE].copy()
            else:
                arrays[f"{tag}/{k}"] = v
    cfg = dict(
        modality_names=case["mods"],
        n_views=int(model.n_views),
        n_spatial_dims=int(model.n_spatial_dims),
        kernel_warp=case["kernel_warp"],
        kernel_data=case["kernel_data"],
        n_latent_gps=case["n_latent_gps"],
        fixed_view_idx=case["fixed_view_idx"],
        n_samples={m: [int(x) for x in case["n_samples"][m]] for m in case["mods"]},
        S=S,
        m_X=case["m_X"],
        m_G=case["m_G"],
        summary_only=big,
        recipe=case.get("recipe"),
        fixed_warp_kernel_variances=case.get("fixed_warp_kernel_variances"),
        fixed_warp_kernel_lengthscales=case.get("fixed_warp_kernel_lengthscales"),
        fixed_data_kernel_lengthscales=case.get("fixed_data_kernel_lengthscales"),
    )
    arrays["cfg_json"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300)
    m0 = case["mods"][0]
    print(
        "  %s: %.1f KB; fp32-vs-fp64 rel: G_means %.2e G_samples %.2e F %.2e loss %.2e"
        % (
            name,
            os.path.getsize(path) / 1024,
            rel(r32[f"G_means/{m0}"], r64[f"G_means/{m0}"]),
            rel(r32[f"G_samples/{m0}"], r64[f"G_samples/{m0}"]),
            rel(r32[f"F_latent/{m0}"], r64[f"F_latent/{m0}"]),
            rel(r32["loss"], r64["loss"]),
        )
    )


def h5ad_example():
    """examples/synthetic_data.h5ad walked by raw offsets (no h5py here)."""
    b = open("/root/reference/examples/synthetic_data.h5ad", "rb").read()
    Y = np.frombuffer(b, dtype="<f4", count=200 * 30, offset=2048).reshape(200, 30)
    batch = np.frombuffer(b, dtype="<i8", count=200, offset=39488)
    X = np.frombuffer(b, dtype="<f8", count=400, offset=47680).reshape(200, 2)
    assert (batch[:100] == 0).all() and (batch[100:] == 1).all()
    return X.astype(np.float32), Y.copy()


def main():
    cases = {}
    X, Y = h5ad_example()
    cases["c1_example_fixed0"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": [100, 100]}, m_X=25, m_G=25, S=5, seed=1,
        n_latent_gps={"expression": None}, kernel_warp="rbf", kernel_data="rbf",
        fixed_view_idx=0,
    )
    X, Y, ns = recipes.grid_views(side=9, n_views=3, n_out=6, seed=11)
    cases["c2_three_free_views"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": ns}, m_X=12, m_G=14, S=3, seed=2,
        n_latent_gps={"expression": None}, kernel_warp="rbf", kernel_data="rbf",
        fixed_view_idx=None,
    )
    X, Y, ns = recipes.grid_views(side=10, n_views=2, n_out=12, seed=12)
    cases["c3_lmc_matern12_warp"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": ns}, m_X=15, m_G=15, S=4, seed=3,
        n_latent_gps={"expression": 3}, kernel_warp="matern12", kernel_data="rbf",
        fixed_view_idx=None,
    )
    X, Y, ns = recipes.grid_views(side=5, n_views=2, n_out=5, seed=13, dims=3)
    gt = np.random.default_rng(130).uniform(0, 10, size=(1, 17, 3)).astype(np.float32)
    cases["c4_3d_gtest"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": ns}, m_X=20, m_G=20, S=2, seed=4,
        n_latent_gps={"expression": None}, kernel_warp="rbf", kernel_data="rbf",
        fixed_view_idx=[0], G_test={"expression": gt},
    )
    Xa, Ya, nsa = recipes.grid_views(side=8, n_views=2, n_out=7, seed=14)
    Xb, Yb, nsb = recipes.grid_views(side=6, n_views=2, n_out=4, seed=15)
    cases["c5_two_modalities"] = dict(
        mods=["rna", "protein"], X={"rna": Xa, "protein": Xb}, Y={"rna": Ya, "protein": Yb},
        n_samples={"rna": nsa, "protein": nsb}, m_X=16, m_G=18, S=3, seed=5,
        n_latent_gps={"rna": None, "protein": 2}, kernel_warp="rbf", kernel_data="matern32",
        fixed_view_idx=None,
    )
    X, Y, ns = recipes.line_views(n=60, n_views=2, n_out=5, seed=16)
    cases["c6_one_dim"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": ns}, m_X=10, m_G=10, S=3, seed=6,
        n_latent_gps={"expression": None}, kernel_warp="matern32", kernel_data="rbf",
        fixed_view_idx=None,
    )
    X, Y, ns = recipes.grid_views(side=9, n_views=2, n_out=5, seed=17)
    cases["c8_fixed_hyper"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": ns}, m_X=12, m_G=12, S=2, seed=8,
        n_latent_gps={"expression": None}, kernel_warp="rbf", kernel_data="rbf",
        fixed_view_idx=1, fixed_warp_kernel_variances=[0.5, 0.7],
        fixed_warp_kernel_lengthscales=[4.0, 6.0], fixed_data_kernel_lengthscales=[1.5],
    )
    # BASELINE config 4 in miniature: 8 views, view 0 fixed (the 8-view loop of vgpsa.py:259-273)
    X, Y, ns = recipes.grid_views(side=6, n_views=8, n_out=5, seed=19)
    cases["c9_eight_views_fixed0"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": ns}, m_X=14, m_G=16, S=2, seed=9,
        n_latent_gps={"expression": None}, kernel_warp="rbf", kernel_data="rbf",
        fixed_view_idx=0,
    )
    # round 6: unequal view sizes with TWO fixed views given as a list (vgpsa.py:230-234, 262-273), Matern-1/2 data
    # kernel, S = 1 (forward's default)
    X, Y, ns = recipes.grid_views(side=8, n_views=4, n_out=5, seed=20)
    keep = [64, 50, 57, 40]
    rows = np.concatenate([np.arange(v * 64, v * 64 + k) for v, k in enumerate(keep)])
    cases["c10_unequal_two_fixed"] = dict(
        mods=["expression"], X={"expression": X[rows]}, Y={"expression": Y[rows]},
        n_samples={"expression": keep}, m_X=12, m_G=13, S=1, seed=10,
        n_latent_gps={"expression": None}, kernel_warp="rbf", kernel_data="matern12",
        fixed_view_idx=[0, 2],
    )
    # round 6: LMC + G_test with two test samples (the 6-tuple of vgpsa.py:437-489) on unequal views, Matern-3/2 warp
    X, Y, ns = recipes.grid_views(side=9, n_views=2, n_out=9, seed=21)
    keep = [81, 60]
    rows = np.concatenate([np.arange(v * 81, v * 81 + k) for v, k in enumerate(keep)])
    gt = np.random.default_rng(210).uniform(0, 10, size=(2, 11, 2)).astype(np.float32)
    cases["c11_lmc_gtest_unequal"] = dict(
        mods=["expression"], X={"expression": X[rows]}, Y={"expression": Y[rows]},
        n_samples={"expression": keep}, m_X=14, m_G=12, S=3, seed=11,
        n_latent_gps={"expression": 3}, kernel_warp="matern32", kernel_data="rbf",
        fixed_view_idx=None, G_test={"expression": gt},
    )
    # conditioning study: M=200 (warp-GP K_uu cond ~ 2e7), parameters from a seeded recipe
    rc = dict(side=50, n_views=2, n_out=4, seed=18, m=200, state_seed=180)
    X, Y, ns = recipes.grid_views(side=rc["side"], n_views=2, n_out=rc["n_out"], seed=rc["seed"])
    cases["c7_m200_conditioning"] = dict(
        mods=["expression"], X={"expression": X}, Y={"expression": Y},
        n_samples={"expression": ns}, m_X=200, m_G=200, S=2, seed=7,
        n_latent_gps={"expression": None}, kernel_warp="rbf", kernel_data="rbf",
        fixed_view_idx=None, summary_only=True, recipe=rc,
        state_override=recipes.m200_state(X, rc["n_out"], 2, 2, rc["m"], rc["state_seed"]),
    )
    only = sys.argv[1:]
    for name, case in cases.items():
        if only and name not in only:
            continue
        run_case(name, case)


if __name__ == "__main__":
    main()
