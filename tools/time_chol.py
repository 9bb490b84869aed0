"""HIP-event timing of the batched Cholesky + inverse (57 matrices of 200 x 200, the headline step's batch)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spatial_alignment_amd.ops import get_ops
o = get_ops()
for M, B in ((200, 57), (200, 3), (128, 20), (64, 57)):
    g = torch.Generator().manual_seed(0)
    A = torch.randn(B, M, M, generator=g, dtype=torch.float64)
    K = (A @ A.transpose(1, 2) / M + 0.05 * torch.eye(M, dtype=torch.float64)).cuda()
    for _ in range(3):
        o.chol_inv(K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        o.chol_inv(K)
    e1.record()
    torch.cuda.synchronize()
    print(f"GPSA_CHOL_BLOCKED={os.environ.get('GPSA_CHOL_BLOCKED', '1')} M={M} batch={B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
