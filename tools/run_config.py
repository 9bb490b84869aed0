"""Step time of the other BASELINE.json configurations (or the largest single-GPU cut of them) on one MI355X.
Diagnostic: these are parity-test configurations, not bench lines.
Names follow BASELINE.json's 1-based numbering: 3, 4cut, 5cut (LMC cuts of rounds 1-2), and 4, 5 = the
configurations at their STATED size (independent outputs, P = L = 2000 / 1000; S = 1).
usage: python tools/run_config.py 3|4cut|5cut|4|5 [steps] [warmup]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
import spatial_alignment_amd as gp  # noqa: E402
from spatial_alignment_amd.synthetic import make_grid_problem, make_model  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
CFG = {
    # 4 views x 10k spots, 500 outputs through 10 latent GPs, Matern-1/2 warp / RBF data
    "3": dict(side=100, views=4, outputs=500, M=200, latent=10, fixed=None, warp=gp.matern12_kernel, S=5),
    # Visium-scale: 8 views x 5k spots (71 x 71), 2000 genes through 20 latent GPs, M = 500, view 0 fixed
    "4cut": dict(side=71, views=8, outputs=2000, M=500, latent=20, fixed=0, warp=gp.rbf_kernel, S=5),
    # Slide-seq-scale cut that fits one GPU step: 2 views x 100k spots (316 x 316), 1000 genes / 10 latent, M = 1000
    "5cut": dict(side=316, views=2, outputs=1000, M=1000, latent=10, fixed=None, warp=gp.rbf_kernel, S=1),
    # BASELINE config 4 as stated: 8 views x 5041 spots, 2000 genes = 2000 independent outputs, M = 500, view 0 fixed
    "4": dict(side=71, views=8, outputs=2000, M=500, latent=None, fixed=0, warp=gp.rbf_kernel, S=1),
    # BASELINE config 5 as stated: 2 views x 99 856 spots, 1000 independent outputs, M = 1000
    "5": dict(side=316, views=2, outputs=1000, M=1000, latent=None, fixed=None, warp=gp.rbf_kernel, S=1),
}[which]
dd_cpu = make_grid_problem(side=CFG["side"], n_views=CFG["views"], n_outputs=CFG["outputs"], device="cpu")
model = make_model(dd_cpu, m=CFG["M"], n_latent_gps={"expression": CFG["latent"]}, fixed_view_idx=CFG["fixed"],
                   device=dev, kernel_func_warp=CFG["warp"], kernel_func_data=gp.rbf_kernel)
dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev),
          "n_samples_list": d["n_samples_list"]} for m, d in dd_cpu.items()}
if os.environ.get("GPSA_NOCHECK") == "1":
    model.check_numerics = False
view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
Xs = {m: d["spatial_coords"] for m, d in dd.items()}
from spatial_alignment_amd.optim import FusedAdam  # noqa: E402
from spatial_alignment_amd.train import train_step  # noqa: E402

opt = FusedAdam(model.parameters(), lr=1e-2)


MB = int(os.environ.get("GPSA_MICROBATCHES", "1"))  # K > 1: one step = K accumulating passes over row slices
if MB > 1:
    from spatial_alignment_amd.train import Microbatches  # noqa: E402

    mb = Microbatches(model, dd, MB)


def step():
    if MB > 1:
        return mb.step(opt, S=CFG["S"])
    return train_step(model, opt, dd, view_idx, Ns, S=CFG["S"])  # (the reference loop body; lets the engine fuse the ELBO)


for _ in range(warmup):  # allocator growth and first-use code loading settle within the first steps
    l0 = step()
torch.cuda.synchronize()
import ctypes  # noqa: E402

plans = [p for p in model.__dict__.get("_step_plans", {}).values() if p.S == CFG["S"]]
if os.environ.get("GPSA_NOTIMING") == "1" or MB > 1:  # (per-contraction rates are per pass: not meaningful over K passes)
    plans = []
for p in plans:
    p.lib.gpsa_step_timing(p.handle, steps)
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# the three contraction ops of the (first modality's) data GP, timed by the engine on its launch stream:
# op-level rates on the nominal 2*C*L*M^2 flops (M > 256: an op is several launches - operand preparation,
# the tiled product with its triangle mode, the closing reduction)
for p in plans:
    buf = (ctypes.c_float * (3 * steps))()
    n = p.lib.gpsa_step_timing_read(p.handle, buf, steps)
    p.lib.gpsa_step_timing(p.handle, 0)
    if n > 0:
        C_, L_, M_ = CFG["S"] * CFG["views"] * CFG["side"] ** 2, CFG["latent"] or CFG["outputs"], CFG["M"]
        fl = 2.0 * C_ * L_ * M_ * M_
        kept = p.saved_bytes - p.saved_bytes_nokeep  # training keeps the products Omega_l alpha (M <= 256)
        names = ("variance form a^T Omega a" + (" (full product, kept)" if kept else ""),
                 "its alpha-gradient (" + ("streams the kept products" if kept else "ACCUM") + ")", "its Omega-gradient (Gram)")
        for k, name in enumerate(names):
            ms = sum(buf[i * 3 + k] for i in range(n)) / n
            if kept and k == 1:
                print(f"  {name:46s} {ms:8.3f} ms   {kept / ms / 1e6:7.0f} GB/s = {kept / ms / 1e6 / 8000:.2f} of the HBM peak")
            else:
                print(f"  {name:46s} {ms:8.3f} ms   {fl / ms / 1e9:7.1f} TF nominal = {fl / ms / 1e9 / 157.3:.2f} of the fp32-MFMA peak")
print(f"{which}: {CFG['views']} views x {CFG['side'] ** 2} spots, {CFG['outputs']} outputs via {CFG['latent'] or 'no'} latent GPs, "
      f"M={CFG['M']}, S={CFG['S']}: {dt * 1e3:.2f} ms/step, loss {float(l0):.4g} -> {float(loss):.4g}, "
      f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB" + (f", {MB} microbatches per step" if MB > 1 else ""),
      flush=True)
