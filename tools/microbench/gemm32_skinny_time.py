"""The data GP's skinny fp32 products alone (gpsa_gemm): mean = delta^T alpha  (TN, [L x M][M x C]) and
d delta = alpha dmean^T (NT, [M x C][L x C]^T, split-K).  usage: python tools/microbench/gemm32_skinny_time.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatial_alignment_amd import _lib

lib = _lib.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())


def time_gemm(ta, tb, m, n, k, sk, reps=30, beta=0.0):
    A = torch.randn(*((k, m) if ta else (m, k)), dtype=torch.float32, device=dev)
    B = torch.randn(*((n, k) if tb else (k, n)), dtype=torch.float32, device=dev)
    C = torch.zeros(m, n, dtype=torch.float32, device=dev)
    wsb = lib.gpsa_gemm_workspace(0, m, n, 1, sk)
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)

    def run():
        rc = lib.gpsa_gemm(0, ta, tb, m, n, k, 1.0, p(A), A.shape[1], 0, p(B), B.shape[1], 0, beta, p(C), n, 0, 1, sk,
                           p(ws), wsb, ctypes.c_void_p(st))
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
    C.zero_()
    run()
    ref = torch.matmul((A.t() if ta else A).double(), (B.t() if tb else B).double())
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    gb = (A.numel() + B.numel() + C.numel()) * 4 / us * 1e-3
    return us, gb, err


print({k: os.environ[k] for k in os.environ if k.startswith("GPSA_GEMM")})
for name, (ta, tb, m, n, k), sks in [
    ("mean   TN  50x100000x200", (1, 0, 50, 100000, 200), (1,)),
    ("mean   TN  10x200000x200", (1, 0, 10, 200000, 200), (1,)),
    ("ddelta NT 200x50x100000", (0, 1, 200, 50, 100000), (64, 128, 256)),
    ("ddelta NT 200x10x200000", (0, 1, 200, 10, 200000), (128, 256)),
    ("abar   NN 200x100000x50", (0, 0, 200, 100000, 50), (1,)),
    ("abar   NN 200x200000x10", (0, 0, 200, 200000, 10), (1,)),
    ("mean   TN  20x100000x200", (1, 0, 20, 100000, 200), (1,)),
    ("dresid NT 200x2x10000", (0, 1, 200, 2, 10000), (32, 128)),
    ("LMC    NN 200000x500x10", (0, 0, 200000, 500, 10), (1,)),
    ("LMCb   NT 200000x10x500", (0, 1, 200000, 10, 500), (1,)),
    ("LMCw   TN 10x500x200000", (1, 0, 10, 500, 200000), (64, 256)),
]:
    for sk in sks:
        us, gb, err = time_gemm(ta, tb, m, n, k, sk)
        print(f"{name}  splitk {sk:3d}  {us:7.1f} us  {gb:6.0f} GB/s  err {err:.1e}")

for name, (ta, tb, m, n, k) in [("abar   NN 200x100000x50  beta=1", (0, 0, 200, 100000, 50)),
                                ("abar   NN 200x200000x10  beta=1", (0, 0, 200, 200000, 10))]:
    us, gb, err = time_gemm(ta, tb, m, n, k, 1, beta=1.0)
    print(f"{name}  {us:7.1f} us  err {err:.1e}")
