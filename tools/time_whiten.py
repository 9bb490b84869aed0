"""Time the fp64 projection kernels (csrc/whiten.hip, csrc/proj64.hip) at the shapes of the step.
usage: [GPSA_PROJ64=0|1] [GPSA_PROJ64_MIN_TILES=n] python tools/time_whiten.py [reps]
The packed inverse is built once; the timed calls pass Kinv = NULL (as the engine's later passes do)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from spatial_alignment_amd.ops import get_ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
hip = get_ops()
lib = hip.lib
f64, f32 = torch.float64, torch.float32


def p(t):
    return None if t is None else t.data_ptr()


def st():
    return torch.cuda.current_stream().cuda_stream


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def case(name, M, C, kind, B=1):
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(B, M, M, dtype=f64, generator=g).to(dev)
    Kinv = (A @ A.transpose(1, 2) / M + torch.eye(M, dtype=f64, device=dev)).contiguous()
    X = torch.randn(B, M, C, dtype=f64, generator=g).to(dev)
    if os.environ.get("GPSA_TW_CONST") == "1":  # constant operands (the microbenchmarks' regime): is the rate data-dependent?
        X = torch.ones_like(X)
        Kinv = torch.full_like(Kinv, 2.0)
    wsb = int(lib.gpsa_whiten_workspace(M)) * B
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    a64 = torch.empty(B, M, C, dtype=f64, device=dev)
    q = torch.empty(B, C, dtype=f64, device=dev)
    if kind == "dual":
        a32 = torch.empty(M, C, dtype=f32, device=dev)
        assert lib.gpsa_whiten_f64_dual(p(Kinv), p(X), M, C, p(a64), p(a32), p(q), p(ws), wsb, st()) == 0
        fn = lambda: lib.gpsa_whiten_f64_dual(None, p(X), M, C, p(a64), p(a32), p(q), p(ws), wsb, st())  # noqa: E731
        ref = Kinv[0] @ X[0]
    elif kind == "bwd":  # fp32 right-hand side, fp64 result, no q
        X32 = X[0].float().contiguous()
        assert lib.gpsa_whiten_f64(p(Kinv), 0, p(X32), M, C, 1, p(a64), None, p(ws), wsb, st()) == 0
        fn = lambda: lib.gpsa_whiten_f64(None, 0, p(X32), M, C, 1, p(a64), None, p(ws), wsb, st())  # noqa: E731
        ref = Kinv[0] @ X32.double()
    else:  # batched fp64 -> fp64 with q (the warp GPs)
        assert lib.gpsa_whiten_batched_f64(p(Kinv), M * M, p(X), M, C, M * C, p(a64), p(q), B, p(ws), wsb, st()) == 0
        fn = lambda: lib.gpsa_whiten_batched_f64(None, M * M, p(X), M, C, M * C, p(a64), p(q), B, p(ws), wsb, st())  # noqa: E731
        ref = Kinv[0] @ X[0]
    med, mn = timed(fn)
    err = float((a64[0] - ref).norm() / ref.norm())
    fl = 2.0 * B * (16 * ((M + 15) // 16)) ** 2 * C
    print(f"{name:34s} M={M} C={C} B={B}: median {med:7.1f} us  min {mn:7.1f} us   {fl / med / 1e6:6.1f} TF executed "
          f"= {fl / med / 1e6 / 78.6:.2f} of the fp64-MFMA peak   rel.err {err:.1e}", flush=True)


print("GPSA_PROJ64 =", os.environ.get("GPSA_PROJ64"), " GPSA_PROJ64_MIN_TILES =", os.environ.get("GPSA_PROJ64_MIN_TILES"),
      " GPSA_PROJ64_SKIP =", os.environ.get("GPSA_PROJ64_SKIP"))
case("data GP forward (dual, q)", 200, 100000, "dual")
case("data GP backward (fp32 in)", 200, 100000, "bwd")
if os.environ.get("GPSA_TW_SHORT") == "1":
    sys.exit(0)
if os.environ.get("GPSA_TW_SHORT") == "3":  # the row pitch: one 400 000-column panel against 40 panels of 10 000 columns
    case("1 x 400 000 columns", 200, 400000, "batched", 1)
    case("40 x 10 000 columns", 200, 10000, "batched", 40)
    case("1 x 100 000 columns", 200, 100000, "batched", 1)
    case("10 x 10 000 columns", 200, 10000, "batched", 10)
    case("25 x 4 000 columns", 200, 4000, "batched", 25)
    sys.exit(0)
if os.environ.get("GPSA_TW_SHORT") == "2":  # long panels only: ramp and tail amortised
    case("backward, C = 400 000", 200, 400000, "bwd")
    case("forward, C = 400 000", 200, 400000, "dual")
    sys.exit(0)
case("config 3 forward (dual, q)", 200, 200000, "dual")
case("S=1 forward (dual, q)", 200, 20000, "dual")
case("1/8 shard forward (dual, q)", 200, 12500, "dual")
case("warp GPs (2 views)", 200, 10000, "batched", 2)
case("M=100 long", 100, 100000, "dual")
