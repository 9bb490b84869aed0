import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spatial_alignment_amd.ops import get_ops
o = get_ops()
torch.manual_seed(0)
for M in (64, 200):
    A = torch.randn(4, M, M, dtype=torch.float64)
    K = A @ A.transpose(1, 2) / M + 2.0 * torch.eye(M, dtype=torch.float64)   # cond ~ 3
    Linv, logdet, info = o.chol_inv(K.cuda())
    Li = Linv.cpu()
    ref = torch.linalg.inv(torch.linalg.cholesky(K))
    res = (Li @ K @ Li.transpose(1, 2) - torch.eye(M, dtype=torch.float64)).abs().max()
    print(f"M={M}: max |Linv - ref| / max|ref| = {float((Li - ref).abs().max() / ref.abs().max()):.2e}, residual {float(res):.2e}, "
          f"logdet err {float((logdet.cpu() - torch.logdet(K)).abs().max()):.2e}")
