"""The data GP's projection alone (gpsa_whiten_f64: fp64 K_uf -> fp32 alpha + q; gpsa_whiten_axpy_f32: the backward's
solve) at the headline size.  usage: python tools/microbench/whiten_time.py [M C]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatial_alignment_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
lib = _lib.load()
dev = "cuda"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, M, device=dev, generator=g, dtype=torch.float64)
Kinv = A @ A.t() / M + torch.eye(M, device=dev, dtype=torch.float64)
Kuf = torch.randn(M, C, device=dev, generator=g, dtype=torch.float64)
alpha = torch.empty(M, C, device=dev, dtype=torch.float32)
q = torch.empty(C, device=dev, dtype=torch.float64)
wsb = lib.gpsa_whiten_workspace(M)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
abar = torch.randn(M, C, device=dev, generator=g)
qbar = torch.randn(C, device=dev, generator=g)
gamma = torch.empty(M, C, device=dev)


def fwd(pack=True):
    rc = lib.gpsa_whiten_f64(p(Kinv) if pack else None, 1, p(Kuf), M, C, 0, p(alpha), p(q), p(ws), wsb, st)
    assert rc == 0, rc


def bwd():
    rc = lib.gpsa_whiten_axpy_f32(None, p(abar), M, C, p(alpha), p(qbar), 2.0, p(gamma), p(ws), wsb, st)
    assert rc == 0, rc


def timeit(f, reps=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


fwd(True)
tag = {k: os.environ[k] for k in os.environ if k.startswith("GPSA_WH")}
tf = lambda us: 2.0 * (16 * ((M + 15) // 16)) ** 2 * C / us * 1e-6
uf, ub = timeit(lambda: fwd(False)), timeit(bwd)
ref = (Kinv @ Kuf[:, :512])
err = ((alpha[:, :512].double() - ref).abs().max() / ref.abs().max()).item()
print(tag, f"M {M} C {C}  forward {uf:6.1f} us ({tf(uf):4.1f} TF fp64)  backward {ub:6.1f} us ({tf(ub):4.1f} TF)  err {err:.1e}")
