"""Time the data GP's covariance backward (gpsa_kmat_bwd_x64_axpy) alone at the headline size and compare the
register-accumulating kernel with the older one (env GPSA_KMAT_BWD_D2 / _ROWS / _PER are read once per process:
run this script once per setting).  usage: python tools/microbench/kmat_bwd_time.py [M C]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatial_alignment_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
D = 2
lib = _lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.rand(M, D, device=dev, generator=g) * 10
X = (torch.rand(C, D, device=dev, generator=g) * 10).double()
ls, var = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
Kbar = torch.randn(M, C, device=dev, generator=g)
X2 = torch.randn(M, C, device=dev, generator=g)
d = torch.randn(C, device=dev, generator=g)
dZ = torch.empty(M, D, dtype=torch.float64, device=dev)
dX = torch.empty(C, D, dtype=torch.float64, device=dev)
dp = torch.empty(2, dtype=torch.float64, device=dev)
wsb = lib.gpsa_kmat_bwd_workspace(1, M, C, D)  # GPSA_F64 = 1
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())


def run():
    rc = lib.gpsa_kmat_bwd_x64_axpy(0, p(Z), M, p(X), C, D, p(ls), p(var), p(Kbar), p(X2), p(d), 2.0, p(dZ), p(dX),
                                    p(dp), p(ws), wsb, ctypes.c_void_p(st))
    assert rc == 0, rc


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
tag = {k: os.environ[k] for k in os.environ if k.startswith("GPSA_KMAT")}
print(tag, "us per call", round(e0.elapsed_time(e1) * 20, 1), "dZ", dZ.sum().item(), "dX", dX.abs().sum().item(), "dp", dp.tolist())
