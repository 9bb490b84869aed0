"""HIP-event timing of the streaming alpha-gradient over the kept products (5.5-6.3 TB/s in isolation at any row-tile grouping: at the HBM ceiling)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spatial_alignment_amd.ops import get_ops
o = get_ops()
M, L = 200, 50
for C in (100000, 12500):
    nb = o.lib.gpsa_quadform_keep_f32_bytes(M, C, L)
    W = torch.randn(nb // 4, device="cuda")
    g = torch.randn(L, C, device="cuda"); dc = torch.randn(M, L, device="cuda"); dm = torch.randn(L, C, device="cuda")
    out = torch.empty(M, C, device="cuda")
    st = o._stream(out)
    run = lambda: o.lib.gpsa_quadform_bwd_alpha_kept_f32(W.data_ptr(), g.data_ptr(), M, C, L, dc.data_ptr(), dm.data_ptr(), out.data_ptr(), st)
    best = 1e9
    for rnd in range(3):
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"C={C}: {best*1e3:.1f} us  {nb/best/1e6:.0f} GB/s")
