"""EXPLORATORY (round-5 verdict item 9; never the headline): would a three-way bf16 split of the fp32 contraction keep the
1e-4 contract?  fp32 MFMA is 1/16 of the bf16 rate on gfx950; writing a = a1 + a2 + a3 with bf16 pieces (8 + 8 + 8
significant bits) and keeping the six products a_i b_j with i + j <= 4, accumulated in fp32, reproduces the fp32 product
to ~2^-22 per term at 6/16 of the matrix time.  This script EMULATES that arithmetic on the CPU (bf16 x bf16 products are
exact in fp32; the accumulation is an fp32 matmul per product pair, the six partial results added smallest first) on
operands shaped like the headline step's - alpha = K^-1 K_uf of an RBF layer with M = 200 inducing points (cond ~1e7,
entries of alternating sign), Omega_l = A A^T + 1e-5 I - and prints, against an fp64 evaluation OF THE SAME fp32-ROUNDED
OPERANDS (the contract of the fp32 kernels: docs/LAB_NOTES.md) and against fp64 on the unrounded ones:
    W_l = Omega_l alpha,  v[l,c] = alpha_c . W_l[:,c],  abar = sum_l g_l W_l
and the Gram product of the backward, dOmega_l = sum_c g[l,c] alpha_c alpha_c^T (the left operand g o alpha is formed in
fp32 registers and split there).
usage: python tools/split_bf16_emulation.py [M] [C] [L] [lengthscale]"""
import sys

import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
LS = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
f32, f64, bf = torch.float32, torch.float64, torch.bfloat16


def split3(x):
    """x (fp32) -> three bf16-valued fp32 tensors with x ~= a1 + a2 + a3 (round to nearest even each time)"""
    a1 = x.to(bf).to(f32)
    r = x - a1
    a2 = r.to(bf).to(f32)
    a3 = (r - a2).to(bf).to(f32)
    return a1, a2, a3


def split_matmul(A, B, terms=((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))):
    """sum of the products A_i B_j for the listed (i, j), each an fp32 matmul, added smallest first"""
    As, Bs = split3(A), split3(B)
    out = None
    for i, j in terms:
        p = As[i] @ Bs[j]
        out = p if out is None else out + p
    return out


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def main():
    g = torch.Generator().manual_seed(0)
    # an RBF layer like the data GP's at initialisation: M inducing points on a jittered lattice of [0, 10]^2, lengthscale
    # as the model's (exp(log 1) ... the bench problem trains towards ~1), C spots uniform in the square
    side = int(M ** 0.5 + 0.999)
    lin = torch.linspace(0, 10, side, dtype=f64)
    Z = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1).reshape(-1, 2)[:M]
    Z = Z + 0.05 * torch.randn(Z.shape, generator=g, dtype=f64)
    X = 10 * torch.rand(C, 2, generator=g, dtype=f64)
    ls = LS

    def k(a, b):
        return torch.exp(-0.5 * torch.cdist(a / ls, b / ls).square())

    Kuu = k(Z, Z) + 1e-5 * torch.eye(M, dtype=f64)
    alpha64 = torch.linalg.solve(Kuu, k(Z, X))                      # [M, C], the projection (fp64 in the step)
    A = 0.1 * torch.randn(L, M, M, generator=g, dtype=f64) + 0.3 * torch.eye(M, dtype=f64)
    Om64 = A @ A.transpose(1, 2) + 1e-5 * torch.eye(M, dtype=f64)   # [L, M, M]
    gw = torch.randn(L, C, generator=g, dtype=f64)
    print(f"M = {M}, C = {C}, L = {L}: cond(K_uu) = {float(torch.linalg.cond(Kuu)):.1e}, max |alpha| = "
          f"{float(alpha64.abs().max()):.1e}")
    a32, O32 = alpha64.to(f32), Om64.to(f32)
    # references
    W_r = O32.double() @ a32.double()                               # fp64 on the ROUNDED operands (the kernels' contract)
    W_t = Om64 @ alpha64                                            # fp64 on the unrounded ones
    v_r, v_t = (a32.double() * W_r).sum(1), (alpha64 * W_t).sum(1)
    ab_r, ab_t = (gw.unsqueeze(1) * W_r).sum(0), (gw.unsqueeze(1) * W_t).sum(0)
    rows = []
    for name, fn in (
        ("fp32 matmul (what the fp32 MFMA path computes)", lambda: O32 @ a32),
        ("bf16 x 3, six products (i + j <= 4)", lambda: torch.stack([split_matmul(O32[l], a32) for l in range(L)])),
        ("bf16 x 3, three products (i + j <= 3)",
         lambda: torch.stack([split_matmul(O32[l], a32, ((1, 0), (0, 1), (0, 0))) for l in range(L)])),
        ("bf16 x 2, three products", lambda: torch.stack([split_matmul(O32[l], a32, ((1, 1), (1, 0), (0, 1), (0, 0))) for l in range(L)])),
        ("bf16 x 1 (plain bf16 operands)", lambda: torch.stack([split_matmul(O32[l], a32, ((0, 0),)) for l in range(L)])),
    ):
        W = fn()
        v = (a32 * W).sum(1)                       # the kernels close v in fp32 from the accumulators
        ab = (gw.float().unsqueeze(1) * W).sum(0)
        rows.append((name, rel(W, W_r), rel(v, v_r), rel(ab, ab_r), rel(W, W_t), rel(v, v_t), rel(ab, ab_t)))
    print(f"{'arithmetic':52s} | vs fp64 on the rounded operands: W, v, abar | vs fp64 on the unrounded: W, v, abar")
    for r in rows:
        print(f"{r[0]:52s} | {r[1]:.1e} {r[2]:.1e} {r[3]:.1e} | {r[4]:.1e} {r[5]:.1e} {r[6]:.1e}")
    # the Gram product of one output: (g o alpha) alpha^T, [M, C] x [C, M]
    ga = (gw[0].float() * a32)
    G_r = ga.double() @ a32.double().t()
    print(f"{'Gram product dOmega = (g o alpha) alpha^T':52s} | vs fp64 on the rounded operands")
    for name, terms in (("fp32 matmul", None), ("bf16 x 3, six products", ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))),
                        ("bf16 x 2, four products", ((1, 1), (1, 0), (0, 1), (0, 0))),
                        ("bf16 x 2, three products", ((1, 0), (0, 1), (0, 0)))):
        Gm = ga @ a32.t() if terms is None else split_matmul(ga, a32.t().contiguous(), terms)
        print(f"{name:52s} | {rel(Gm, G_r):.1e}")


if __name__ == "__main__":
    main()
