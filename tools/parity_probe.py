"""Diagnostic (GPU): the bench model after ``--train`` Adam steps, one training step at S in --S against the fp64
oracle in both inducing-gradient modes; prints every gradient's relative error and, for the scalar parameters, the
values themselves (a relative error on a gradient that is passing through zero says little).
    python tools/parity_probe.py --train 200 --S 5 1"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=200)
    ap.add_argument("--S", type=int, nargs="+", default=[5])
    ap.add_argument("--side", type=int, default=100)
    ap.add_argument("--outputs", type=int, default=50)
    ap.add_argument("--M", type=int, default=200)
    ap.add_argument("--modes", type=int, nargs="+", default=[1, 0])
    args = ap.parse_args()
    import __graft_entry__ as ge

    ge.build()
    from oracle import gpsa_oracle as orc
    from spatial_alignment_amd.optim import FusedAdam
    from spatial_alignment_amd.synthetic import make_grid_problem, make_model
    from spatial_alignment_amd.train import train_step

    dev = torch.device("cuda:0")
    m = "expression"
    dd_cpu = make_grid_problem(side=args.side, n_views=2, n_outputs=args.outputs, device="cpu")
    model = make_model(dd_cpu, m=args.M, device=dev)
    dd = {m: {"spatial_coords": dd_cpu[m]["spatial_coords"].to(dev), "outputs": dd_cpu[m]["outputs"].to(dev),
              "n_samples_list": dd_cpu[m]["n_samples_list"]}}
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    opt = FusedAdam(model.parameters(), lr=1e-2)
    torch.manual_seed(1000)
    for _ in range(args.train):
        train_step(model, opt, dd, view_idx, Ns, S=5)
    torch.cuda.synchronize()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for name in ("mean_slopes", "mean_intercepts"):
        state.setdefault(name, getattr(model, name).detach().cpu().clone())
    N, L = dd_cpu[m]["spatial_coords"].shape[0], args.outputs
    cfg = dict(modality_names=[m], n_views=2, n_spatial_dims=2, kernel_warp="rbf", kernel_data="rbf",
               n_latent_gps={m: None}, fixed_view_idx=None)
    rel = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / b.double().norm())
    for S in args.S:
        gen = torch.Generator().manual_seed(7)
        eps_G = [torch.randn(S, n_v, 2, generator=gen) for n_v in dd_cpu[m]["n_samples_list"]]
        eps_F = {m: torch.randn(S, N, L, generator=gen)}
        t0 = time.time()
        ref = orc.evaluate(state, cfg, {m: dd_cpu[m]["spatial_coords"]}, {m: dd_cpu[m]["outputs"]},
                           {m: dd_cpu[m]["n_samples_list"]}, S, eps_G, eps_F, want_grads=True, dtype=torch.float64)
        print(f"S={S}: fp64 oracle {time.time() - t0:.1f} s, loss {float(ref['loss']):.6e}", flush=True)
        for exact in args.modes:
            for fuse in (True, False):
                model.exact_inducing_grad, model.fuse_elbo = bool(exact), fuse
                model.inject_noise([e.to(dev) for e in eps_G], {m: eps_F[m].to(dev)}, None)
                model.zero_grad(set_to_none=True)
                out = model.forward({m: dd[m]["spatial_coords"]}, view_idx=view_idx, Ns=Ns, S=S)
                loss = model.loss_fn(dd, out[3])
                loss.backward()
                gerr = {k: rel(p.grad, ref["grads"][k]) for k, p in model.named_parameters()
                        if p.grad is not None and k in ref["grads"] and float(ref["grads"][k].norm()) > 0}
                print(f"  exact={exact} fuse={fuse}: F {rel(out[3][m], ref['F_obs'][m]):.1e} loss "
                      f"{rel(loss.reshape(1), ref['loss'].reshape(1)):.1e}  " +
                      " ".join(f"{k.split('.')[0]}={v:.1e}" for k, v in gerr.items()), flush=True)
                for k, p in model.named_parameters():
                    if p.numel() <= 2 and k in ref["grads"]:
                        print(f"      {k}: ours {p.grad.detach().cpu().double().tolist()} ref {ref['grads'][k].tolist()}")
    model.exact_inducing_grad = None


if __name__ == "__main__":
    main()
