"""EXPLORATORY (round-5 verdict item 9; never the headline, the timed dtype stays f32): the three-way bf16 split of the
fp32 contraction ON THE GPU - a kernel-level parity table and the loop-level rate.

* parity: W_l = Omega_l alpha through ``gpsa_experiment_split_bf16_product`` (csrc/split_bf16.hip: real
  v_mfma_f32_16x16x32_bf16 instructions, operands split in registers) for the variants 6 / 4 / 3 / 1 products and the fp32
  instruction on the same tiling, on operands shaped like the headline step's (alpha = K^-1 K_uf of an RBF layer with
  M = 200 inducing points, Omega_l = A A^T + 1e-5 I), against fp64 on the same fp32-rounded operands (the contract of the
  fp32 kernels) - W, v[l,c] = alpha_c . W_l[:,c] and abar = sum_l g_l W_l, norm-wise relative errors;
* rate: the MFMA + LDS-fragment-read loop of panel_elbo_kernel's tile in fp32 and in split form
  (``gpsa_experiment_split_bf16_rate``), HIP-event timed.

usage: python tools/split_bf16_parity.py [--json] [M] [C] [L] [lengthscale]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def operands(M, C, L, ls, seed=0):
    f64 = torch.float64
    g = torch.Generator().manual_seed(seed)
    side = int(M ** 0.5 + 0.999)
    lin = torch.linspace(0, 10, side, dtype=f64)
    Z = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1).reshape(-1, 2)[:M]
    Z = Z + 0.05 * torch.randn(Z.shape, generator=g, dtype=f64)
    X = 10 * torch.rand(C, 2, generator=g, dtype=f64)
    k = lambda a, b: torch.exp(-0.5 * torch.cdist(a / ls, b / ls).square())  # noqa: E731
    Kuu = k(Z, Z) + 1e-5 * torch.eye(M, dtype=f64)
    alpha = torch.linalg.solve(Kuu, k(Z, X))
    A = 0.1 * torch.randn(L, M, M, generator=g, dtype=f64) + 0.3 * torch.eye(M, dtype=f64)
    Om = A @ A.transpose(1, 2) + 1e-5 * torch.eye(M, dtype=f64)
    gw = torch.randn(L, C, generator=g, dtype=f64)
    return alpha, Om, gw, float(torch.linalg.cond(Kuu))


def run(M=200, C=4096, L=6, ls=1.6):
    import __graft_entry__ as ge

    ge.build()
    from spatial_alignment_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    alpha, Om, gw, cond = operands(M, C, L, ls)
    a32, O32 = alpha.float(), Om.float()
    W_r = O32.double() @ a32.double()
    v_r = (a32.double() * W_r).sum(1)
    ab_r = (gw.unsqueeze(1) * W_r).sum(0)
    ad, Od = a32.to(dev).contiguous(), O32.to(dev).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())  # noqa: E731
    names = {0: "fp32 instruction (v_mfma_f32_16x16x4_f32), same tiling", 6: "bf16 x 3, six products (i + j <= 4)",
             4: "bf16 x 2, four products", 3: "bf16 x 2, three products (no a2 b2)", 1: "plain bf16 operands"}
    parity = {}
    for nprod in (0, 6, 4, 3, 1):
        W = torch.empty(L, M, C, dtype=torch.float32, device=dev)
        _lib.check(lib.gpsa_experiment_split_bf16_product(Od.data_ptr(), ad.data_ptr(), M, C, L, nprod, W.data_ptr(), st),
                   "gpsa_experiment_split_bf16_product")
        torch.cuda.synchronize()
        Wc = W.cpu()
        v = (a32 * Wc).sum(1)
        ab = (gw.float().unsqueeze(1) * Wc).sum(0)
        parity[names[nprod]] = dict(W=rel(Wc, W_r), v=rel(v, v_r), abar=rel(ab, ab_r))
    # loop-level rate
    out = torch.empty(65536, dtype=torch.float32, device=dev)
    outputs = 400
    rate = {}
    for nprod in (0, 6, 4, 3):
        best = 1e30
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.gpsa_experiment_split_bf16_rate(nprod, outputs, out.data_ptr(), st), "rate")
            e1.record()
            torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1))
        rate[names[nprod]] = dict(us_per_output_and_workgroup=1e3 * best / outputs,
                                  tflops_fp32_equivalent_M200=2.0 * 200 * 200 * 128 * outputs * 256 / best / 1e9)
    base = rate[names[0]]["us_per_output_and_workgroup"]
    for v in rate.values():
        v["speedup_vs_fp32_loop"] = base / v["us_per_output_and_workgroup"]
    return dict(shape=dict(M=M, C=C, L=L, cond_Kuu=cond, max_abs_alpha=float(alpha.abs().max())),
                parity_vs_fp64_on_the_rounded_operands=parity, loop_rate=rate,
                note="EXPERIMENT, not the timed path: the timed step computes in the fp32 matrix instructions; parity = "
                     "norm-wise relative error of W_l = Omega_l alpha, v = alpha . W and abar = sum_l g_l W_l; loop_rate = "
                     "the MFMA + LDS-read loop of the fused ELBO kernel's tile only (no staging, no closing)")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--json"]
    M = int(args[0]) if len(args) > 0 else 200
    C = int(args[1]) if len(args) > 1 else 4096
    L = int(args[2]) if len(args) > 2 else 6
    ls = float(args[3]) if len(args) > 3 else 1.6
    res = run(M, C, L, ls)
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        s = res["shape"]
        print(f"M = {s['M']}, C = {s['C']}, L = {s['L']}: cond(K_uu) = {s['cond_Kuu']:.1e}, max |alpha| = {s['max_abs_alpha']:.1e}")
        print(f"{'arithmetic (GPU, real matrix instructions)':62s} |    W        v       abar   (vs fp64 on the rounded operands)")
        for k, v in res["parity_vs_fp64_on_the_rounded_operands"].items():
            print(f"{k:62s} | {v['W']:.1e}  {v['v']:.1e}  {v['abar']:.1e}")
        print(f"{'loop of the fused ELBO tile (13 x 2 accumulators)':62s} | us / output / workgroup   TF fp32-equivalent   speed-up")
        for k, v in res["loop_rate"].items():
            print(f"{k:62s} | {v['us_per_output_and_workgroup']:8.2f} {v['tflops_fp32_equivalent_M200']:18.1f} {v['speedup_vs_fp32_loop']:14.2f}")
