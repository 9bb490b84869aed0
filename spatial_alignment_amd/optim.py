"""The optimiser of the reference's training loop (torch.optim.Adam, examples/grid_example.py:59-78) as ONE
HIP launch over all parameters (``gpsa_adam_step``, csrc/step.hip): same update, same fp32 arithmetic, step
counter on the device (so that the step can be captured into a hipGraph)."""
import ctypes as C

import torch

from . import _lib
from . import torch_ops  # noqa: F401  (registers torch.ops.gpsa.*)

_raw_stream = torch._C._cuda_getCurrentRawStream


class FusedAdam(torch.optim.Optimizer):
    """Adam without weight decay / amsgrad over fp32 HIP parameters; one launch per step (plus the step
    counter's).  Parameters without a gradient in a step keep their value and their moments (as torch's).

    State layout = ``torch.optim.Adam(capturable=True)``'s: ``state[p] = {"step": fp32 device tensor, "exp_avg",
    "exp_avg_sq"}``, so ``state_dict()`` / ``load_state_dict()`` round-trip the step count and interchange with
    torch's Adam.  The parameters of a group that have a gradient in a step share ONE step word (the kernel reads
    one counter per launch): a parameter that sat a step out is moved to a word of its own, so its count does not
    advance (torch's behaviour)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # torch.optim.Adam's group keys ride along (at the only values this optimiser implements) so that a saved
        # state_dict loads into torch's Adam and back
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=True, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self._lib = _lib.load()

    def _step_words(self, ps, dev):
        """the parameters of this launch grouped by the device word that counts their steps.
        ``self._sharers``: id(word) -> [word, parameters sharing it]; the entry holds the word, so neither its id nor
        its address can be taken by another tensor while it is counted (round 3 keyed this by data_ptr and never
        removed entries: a freed word's address, reused, looked shared and was cloned at every step)."""
        share = self.__dict__.setdefault("_sharers", {})
        groups = {}
        for p in ps:
            w = self.state[p].get("step")
            if torch.is_tensor(w) and id(w) in share and w.device == dev:
                groups.setdefault(("word", id(w)), []).append(p)
            elif torch.is_tensor(w) and w.is_cuda:  # a device count nobody registered (foreign state): no host sync
                groups.setdefault(("tensor", id(w)), []).append(p)
            else:  # no count yet, a Python number or a CPU tensor
                groups.setdefault(("value", float(w) if w is not None else 0.0), []).append(p)
        out = []
        for (kind, val), members in groups.items():
            if kind == "word":
                word, n = share[val]
                if n > len(members):  # some sharers have no gradient this step: these move to a word of their own
                    share[val][1] = n - len(members)
                    word = word.clone()
                    share[id(word)] = [word, len(members)]
                    for p in members:
                        self.state[p]["step"] = word
            else:
                if kind == "tensor":
                    word = self.state[members[0]]["step"].detach().to(device=dev, dtype=torch.float32).reshape(1).clone()
                else:
                    word = torch.full((1,), val, dtype=torch.float32, device=dev)
                share[id(word)] = [word, len(members)]
                for p in members:
                    self.state[p]["step"] = word
            out.append((word, members))
        return out

    def state_dict(self):
        """torch's layout: every parameter gets a 0-dim step tensor of its own (the live state shares words)"""
        sd = super().state_dict()
        sd["state"] = {k: dict(v) for k, v in sd["state"].items()}
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().reshape(()).clone()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__["_sharers"] = {}
        for group in self.param_groups:  # loaded step counts are separate tensors: re-share equal ones per group
            words = {}
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    v = float(st["step"])
                    if v not in words:
                        words[v] = torch.full((1,), v, dtype=torch.float32, device=p.device)
                    st["step"] = words[v]
            for w in words.values():
                self._sharers[id(w)] = [w, sum(1 for p in group["params"]
                                               if self.state.get(p) and self.state[p].get("step") is w)]

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
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise _lib.GpsaHipError("FusedAdam: plain Adam only (no weight decay, amsgrad or maximize)")
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
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            grad_of = {id(p): g for p, g in zip(ps, gs)}
            b1, b2 = group["betas"]
            for word, members in self._step_words(ps, dev):  # one launch per step word: one in the usual case
                torch.ops.gpsa.adam_step(members, [grad_of[id(p)] for p in members],
                                         [self.state[p]["exp_avg"] for p in members],
                                         [self.state[p]["exp_avg_sq"] for p in members], word, float(group["lr"]),
                                         float(b1), float(b2), float(group["eps"]))
        return loss
