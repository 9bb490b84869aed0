"""The optimiser of the reference's training loop (torch.optim.Adam, examples/grid_example.py:59-78) as ONE
HIP launch over all parameters (``gpsa_adam_step``, csrc/step.hip): same update, same fp32 arithmetic, step
counter on the device (so that the step can be captured into a hipGraph)."""
import ctypes as C

import torch

from . import _lib

_raw_stream = torch._C._cuda_getCurrentRawStream


class FusedAdam(torch.optim.Optimizer):
    """Adam without weight decay / amsgrad over fp32 HIP parameters; one launch per step (plus the step
    counter's).  Parameters without a gradient in a step keep their value and their moments (as torch's)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._lib = _lib.load()
        self._step_dev = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise _lib.GpsaHipError("FusedAdam: parameters must live on a HIP device")
            gs = []
            for p in ps:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.GpsaHipError("FusedAdam: contiguous fp32 parameters only")
                g = p.grad
                gs.append(g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous())
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            key = (gi, dev.index)
            if key not in self._step_dev:
                self._step_dev[key] = torch.zeros(1, dtype=torch.float32, device=dev)
            n = len(ps)
            arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            numel = (C.c_longlong * n)(*[p.numel() for p in ps])
            b1, b2 = group["betas"]
            rc = self._lib.gpsa_adam_step(n, arr(ps), arr(gs), arr([self.state[p]["exp_avg"] for p in ps]),
                                          arr([self.state[p]["exp_avg_sq"] for p in ps]), numel, float(group["lr"]),
                                          float(b1), float(b2), float(group["eps"]),
                                          self._step_dev[key].data_ptr(), _raw_stream(dev.index))
            _lib.check(rc, "gpsa_adam_step")
        return loss
