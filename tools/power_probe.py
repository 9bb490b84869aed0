"""Package power and shader clock while one kernel runs back to back (rocm-smi sampled from a child process).
usage: [GPSA_PROJ64_SKIP=n] [GPSA_TW_CONST=1] python tools/power_probe.py whiten_bwd|whiten_fwd|elbo|idle [seconds]"""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from spatial_alignment_amd.ops import get_ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "whiten_bwd"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dev = torch.device("cuda:0")
hip = get_ops()
lib = hip.lib
f64, f32 = torch.float64, torch.float32
M, C = 200, 400000
g = torch.Generator(device="cpu").manual_seed(1)
const = os.environ.get("GPSA_TW_CONST") == "1"


def p(t):
    return None if t is None else t.data_ptr()


def st():
    return torch.cuda.current_stream().cuda_stream


if what.startswith("whiten"):
    A = torch.randn(M, M, dtype=f64, generator=g).to(dev)
    Kinv = (A @ A.t() / M + torch.eye(M, dtype=f64, device=dev)).contiguous()
    X = torch.randn(M, C, dtype=f64, generator=g).to(dev)
    if const:
        X, Kinv = torch.ones_like(X), torch.full_like(Kinv, 2.0)
    wsb = int(lib.gpsa_whiten_workspace(M))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    a64 = torch.empty(M, C, dtype=f64, device=dev)
    if what == "whiten_fwd":
        a32, q = torch.empty(M, C, dtype=f32, device=dev), torch.empty(C, dtype=f64, device=dev)
        lib.gpsa_whiten_f64_dual(p(Kinv), p(X), M, C, p(a64), p(a32), p(q), p(ws), wsb, st())
        fn = lambda: lib.gpsa_whiten_f64_dual(None, p(X), M, C, p(a64), p(a32), p(q), p(ws), wsb, st())  # noqa: E731
    else:
        X32 = X.float().contiguous()
        lib.gpsa_whiten_f64(p(Kinv), 0, p(X32), M, C, 1, p(a64), None, p(ws), wsb, st())
        fn = lambda: lib.gpsa_whiten_f64(None, 0, p(X32), M, C, 1, p(a64), None, p(ws), wsb, st())  # noqa: E731
elif what == "copy":  # an HBM-bound kernel for scale
    src = torch.randn(1 << 28, dtype=f32, device=dev)
    dst = torch.empty_like(src)
    fn = lambda: dst.copy_(src)  # noqa: E731
else:
    fn = None


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return "rocm-smi failed: %r" % (e,)
    keep = [ln.strip() for ln in out.splitlines() if ("Power" in ln or "sclk" in ln or "mclk" in ln or "fclk" in ln)]
    return " | ".join(keep)


print(what, "const" if const else "random", "GPSA_PROJ64_SKIP =", os.environ.get("GPSA_PROJ64_SKIP"), flush=True)
print("idle:", smi(), flush=True)
if fn is not None:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    one = e0.elapsed_time(e1) * 1e-3
    n = max(10, int(secs / one))
    t0 = time.perf_counter()
    e0.record()
    for i in range(n):
        fn()
    e1.record()
    for k in range(3):  # the queue is seconds deep: sample while it drains
        time.sleep(secs / 6)
        print("busy:", smi(), flush=True)
    torch.cuda.synchronize()
    print(f"{n} launches, {e0.elapsed_time(e1) / n * 1e3:.1f} us each", flush=True)
