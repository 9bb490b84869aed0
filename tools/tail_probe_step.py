"""Does a training step read or write past the end of anything it allocates?  Every device tensor the package (and this
script) gets from torch.empty is placed at the very END of a device allocation of its own (PYTORCH_NO_CUDA_MEMORY_CACHING=1:
a hipMalloc per tensor; 16-byte aligned start, so the slack behind it is < 16 bytes): an access >= 16 bytes past an arena,
a scratch buffer, an output, the flat gradient buffer or the draws leaves the mapping and faults.  One child process per
configuration (a fault takes the process down).  Round 6: written after tools/fuzz_kernels.py found the panel kernels'
staging ring reading past the kept-products workspace.
usage: python tools/tail_probe_step.py            -> one line per configuration: ok / FAULT"""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "headline-shape, small (M=200, 2x1024 spots, 7 outputs, S=3)": dict(side=32, views=2, outputs=7, M=200, S=3),
    "odd sizes (M=50, 2x961 spots, 5 outputs, S=2)": dict(side=31, views=2, outputs=5, M=50, S=2),
    "three views, one fixed (M=30, 3x225 spots, 4 outputs, S=5)": dict(side=15, views=3, outputs=4, M=30, S=5, fixed=0),
    "LMC (M=64, 2x400 spots, 12 outputs through 3 latent GPs)": dict(side=20, views=2, outputs=12, M=64, S=3, latent=3),
    "large M (M=300, 2x400 spots, 5 outputs, S=1)": dict(side=20, views=2, outputs=5, M=300, S=1),
    "large M, odd columns (M=300, 2x169 spots, 5 outputs, S=1)": dict(side=13, views=2, outputs=5, M=300, S=1),
    # the headline's column count (C = 100k: the projection that forms K_uf itself, the thin update, the covariance
    # backward's resident grid - none of which the small shapes reach) at a few outputs
    "headline columns (M=200, 2x10000 spots, 6 outputs, S=5)": dict(side=100, views=2, outputs=6, M=200, S=5),
    "headline columns, odd count (M=197, 2x9801 spots, 3 outputs, S=5)": dict(side=99, views=2, outputs=3, M=197, S=5),
}


def child(name):
    import torch

    cfg = CONFIGS[name]
    orig_empty = torch.empty
    keep = []

    def tail_empty(*size, **kw):
        dev = kw.get("device")
        if dev is None or torch.device(dev).type != "cuda":
            return orig_empty(*size, **kw)
        meta = orig_empty(*size, **{**kw, "device": "meta"})
        nbytes = meta.numel() * meta.element_size()
        if nbytes == 0:
            return orig_empty(*size, **kw)
        gran = 2 << 20
        tot = (nbytes + gran - 1) // gran * gran + gran
        base = orig_empty(tot, dtype=torch.uint8, device=dev)
        keep.append(base)
        if len(keep) > 64:
            del keep[0]
        start = (tot - nbytes) // 16 * 16
        return base[start:start + nbytes].view(meta.dtype).view(meta.shape)

    torch.empty = tail_empty
    import spatial_alignment_amd as gp
    from spatial_alignment_amd.optim import FusedAdam
    from spatial_alignment_amd.synthetic import make_grid_problem, make_model

    dev = torch.device("cuda:0")
    dd = make_grid_problem(side=cfg["side"], n_views=cfg["views"], n_outputs=cfg["outputs"], device="cpu")
    model = make_model(dd, m=cfg["M"], device=dev, fixed_view_idx=cfg.get("fixed"),
                       n_latent_gps={"expression": cfg.get("latent")})

    def at_tail(t):
        x = tail_empty(*t.shape, dtype=t.dtype, device=dev)
        x.copy_(t)
        return x

    dd = {m: {"spatial_coords": at_tail(d["spatial_coords"]), "outputs": at_tail(d["outputs"]),
              "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
    with torch.no_grad():
        for p in model.parameters():
            p.data = at_tail(p.data)
    opt = FusedAdam(model.parameters(), lr=1e-2)
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: d["spatial_coords"] for m, d in dd.items()}
    for fuse in (True, False):
        model.fuse_elbo = fuse
        for _ in range(2):
            out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=cfg["S"])
            loss = model.loss_fn(dd, out[3])
            opt.zero_grad()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
    with torch.no_grad():
        model.forward(Xs, view_idx=view_idx, Ns=Ns, S=cfg["S"], prediction_mode=True)
    torch.cuda.synchronize()
    print(f"{name}: ok (loss {float(loss):.4g})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1")
        for name in CONFIGS:
            r = subprocess.run([sys.executable, __file__, name], capture_output=True, text=True, env=env, timeout=900)
            out = [ln for ln in r.stdout.splitlines() if ln.endswith(")") and ": ok" in ln]
            if out:
                print(out[-1], flush=True)
            else:
                err = [ln for ln in (r.stderr + r.stdout).splitlines() if "fault" in ln.lower() or "Error" in ln]
                print(f"{name}: FAULT / error (rc {r.returncode}) {err[-1][:200] if err else ''}", flush=True)
