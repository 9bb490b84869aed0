"""fp64 products of the warp GPs' backward / forward alone (gpsa_gemm), over split-K counts.
usage: python tools/microbench/gemm64_time.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatial_alignment_amd import _lib

lib = _lib.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())


def time_gemm(ta, tb, m, n, k, batch, sk, reps=30):
    A = torch.randn(batch, *((k, m) if ta else (m, k)), dtype=torch.float64, device=dev)
    B = torch.randn(batch, *((n, k) if tb else (k, n)), dtype=torch.float64, device=dev)
    C = torch.empty(batch, m, n, dtype=torch.float64, device=dev)
    wsb = lib.gpsa_gemm_workspace(1, m, n, batch, sk)
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    lda, ldb = A.shape[2], B.shape[2]

    def run():
        rc = lib.gpsa_gemm(1, ta, tb, m, n, k, 1.0, p(A), lda, A.shape[1] * A.shape[2], p(B), ldb,
                           B.shape[1] * B.shape[2], 0.0, p(C), n, m * n, batch, sk, p(ws), wsb, ctypes.c_void_p(st))
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    ref = torch.matmul(A.transpose(1, 2) if ta else A, B.transpose(1, 2) if tb else B)
    err = ((C - ref).abs().max() / ref.abs().max()).item()
    tf = 2.0 * m * n * k * batch / us * 1e-6
    return us, tf, err


tag = {k: os.environ[k] for k in os.environ if k.startswith("GPSA_GEMM")}
print(tag)
for name, (ta, tb, m, n, k, b), sks in [
    ("gram   NT 400x200x10000 b2", (0, 1, 400, 200, 10000, 2), (8, 16, 32, 64)),
    ("dKuu   NT 200x200x10000 b2", (0, 1, 200, 200, 10000, 2), (8, 16, 32, 64)),
    ("Wk     NN 200x10000x200 b4", (0, 0, 200, 10000, 200, 4), (1,)),
    ("dresid NT 200x2x10000 b2", (0, 1, 200, 2, 10000, 2), (32, 128, 256)),
    ("KinvSp NN 200x200x200 b3", (0, 0, 200, 200, 200, 3), (1, 2, 4)),
]:
    for sk in sks:
        us, tf, err = time_gemm(ta, tb, m, n, k, b, sk)
        print(f"{name}  splitk {sk:3d}  {us:7.1f} us  {tf:5.1f} TF  err {err:.1e}")
