"""Lazy handles for the draws of a training forward whose likelihood can ride in the data GP's pass.

The reference's loop is ``forward(X, view_idx, Ns, S)`` then ``loss_fn(data_dict, F_samples)``
(examples/grid_example.py:62-78); the observations arrive with the second call (vgpsa.py:532-538), and in every call
site of the reference's examples and experiments the training ``F_samples`` go nowhere but into ``loss_fn``.  So a
training ``forward`` runs everything up to the data GPs (stage 1 of ``gpsa_step_forward``) and returns, for every
modality the fused kernel covers, a ``LazyDraws`` as ``F_samples[m]``: a tensor (``isinstance`` holds; shape, dtype
and device answer without any work) that

* ``loss_fn`` recognises: it then enqueues that modality's data GP with the observations it was given
  (``gpsa_quadform_elbo_f32``: variance, draw, Gaussian likelihood, its gradient and the backward's alpha-gradient in
  one pass over the products - the draws never reach HBM), and that
* MATERIALISES the moment anything else touches it (an operator, an index, ``.cpu()``, ``print``): the engine runs
  that modality's data GP with the separate kernels on the forward's saved state and draws, so the numbers are the
  ones the unfused step produces.

Before ``loss_fn`` has consumed the handle the materialised tensor is a differentiable output of the step (the
modality simply is an unfused one: its backward takes the gradient that reaches the tensor).  Afterwards - at any
time, also after backward() and the optimiser step, as experiments/expression/visium/visium_component_analysis.py
uses its training draws - the handle shows the draws the fused pass made (it writes them out next to the likelihood:
20 MB at the headline size); the likelihood's gradient is committed to the fused form by then, so a gradient that
reaches them raises instead of being dropped.
"""
import torch

from . import _lib
from . import ops as _ops_mod
from . import torch_ops as TO

_DISABLE = torch._C.DisableTorchFunctionSubclass


def _meta_functions():
    T = torch.Tensor
    fns = {T.size, T.dim, T.numel, T.nelement, T.is_floating_point, T.is_complex, T.element_size, T.stride,
           T.is_contiguous, T.get_device, T.type}
    for name in ("shape", "dtype", "device", "requires_grad", "ndim", "is_cuda", "layout", "is_leaf", "is_sparse",
                 "is_quantized", "is_meta", "names", "is_cpu"):
        prop = getattr(T, name, None)
        if prop is not None and hasattr(prop, "__get__"):
            fns.add(prop.__get__)
    return fns


_META = None


class _Lazy(torch.Tensor):
    """a tensor that is computed the first time anything but its metadata is asked of it (subclasses: materialize())"""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        global _META
        if _META is None:
            _META = _meta_functions()
        kwargs = kwargs or {}
        if func in _META:  # answered from the handle's own metadata: nothing is computed
            with _DISABLE():
                return func(*args, **kwargs)
        from torch.utils._pytree import tree_map

        swap = lambda x: x.materialize() if isinstance(x, _Lazy) else x
        with _DISABLE():
            return func(*tree_map(swap, args), **tree_map(swap, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # (wrapper subclasses must have one.  Everything that goes through the Python API is answered by
        #  __torch_function__ above; what still arrives here - an ATen call made below it - gets the values too)
        from torch.utils._pytree import tree_map

        swap = lambda x: x.materialize() if isinstance(x, _Lazy) else x
        return func(*tree_map(swap, args), **tree_map(swap, kwargs or {}))


class LazyProduct(_Lazy):
    """``F_observed_samples[m]`` of an LMC modality in training: F_latent[m] @ W[m]  ([S, N, L] x [L, P]), formed when
    somebody asks for it (an ordinary differentiable product then).  ``loss_fn`` does not: it runs the likelihood, its
    gradient and the two LMC gradient products in one pass over (F_latent, W, Y) without the [S, N, P] tensor
    (gpsa_lmc_loglik_fused_f32) - 400 MB each way at BASELINE config 3."""

    @staticmethod
    def __new__(cls, thunk, shape, device, F_latent, W):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=device,
                                                requires_grad=True)
        t._thunk, t._real, t._peek, t._lmc = thunk, None, None, (F_latent, W)
        return t

    @property
    def is_materialized(self):
        return self._real is not None

    def materialize(self):
        if self._real is not None:
            return self._real
        if torch.is_grad_enabled():
            self._real = self._thunk()
            return self._real
        if self._peek is None:
            self._peek = self._thunk()
        return self._peek


class LazyDraws(_Lazy):
    """``F_latent_samples[m]`` / ``F_observed_samples[m]`` of a fused training forward (see the module docstring)."""

    @staticmethod
    def __new__(cls, rec, i, shape, device):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=device,
                                                requires_grad=True)
        t._rec, t._i, t._real, t._parts, t._peek = rec, i, None, None, None
        return t

    @property
    def is_materialized(self):
        return self._real is not None

    def materialize(self):
        """-> the draws [S, N, L] as a real tensor (computed once, then cached on the handle)"""
        rec, i = self._rec, self._i
        if self._real is not None:
            return self._real
        if rec["state"][i] == "lazy" and torch.is_grad_enabled():
            # not consumed by loss_fn yet: this modality becomes an unfused one of the same step
            self._real = MaterializeFn.apply(rec, i, self._parts)
            rec["state"][i] = "real"
            return self._real
        if rec["state"][i] == "fused" and torch.is_grad_enabled():
            # the likelihood's gradient is committed to the fused form: values to look at; a gradient that reaches
            # them raises (the dummy edge to the step's node is what makes autograd come back here)
            self._real = _ValuesOnlyFn.apply(self._peek_values(), self._parts)
            return self._real
        return self._peek_values()  # under no_grad: plain values, the handle stays what it was

    def _peek_values(self):
        rec, i = self._rec, self._i
        if rec["state"][i] == "fused":
            # the fused pass wrote its draws next to the likelihood, transposed: [L, S N] seen as [S, N, L] (a view; the
            # buffer lives as long as the handle, before and after backward and the optimiser step)
            S, N, L = rec["shapes"][i]
            return rec["FT"][i].view(L, S, N).permute(1, 2, 0)
        if self._peek is None:
            self._peek = materialize_values(rec, i)
        return self._peek


def materialize_values(rec, i, attach=False):
    """Run modality ``i``'s data GP unfused on the saved state of the forward ``rec`` belongs to -> F [S, N, L].
    ``attach``: the forward's own io is switched over too (its backward then treats the modality as unfused)."""
    live = rec.get("live")
    if live is None:
        raise RuntimeError(
            "GPSA: the draws of a training forward whose likelihood was fused into the data GP's pass were never "
            "stored, and the state they could be computed from went with backward().  Look at F_samples before "
            "backward(), run forward under torch.no_grad() for draws to keep, or set model.fuse_elbo = False")
    plan, io, prm, saved, tensors, ins = (live[k] for k in ("plan", "io", "prm", "saved", "tensors", "ins"))
    S, N, L = rec["shapes"][i]
    dev = saved.device
    F = torch.empty(S, N, L, dtype=torch.float32, device=dev)
    io2 = _lib.StepIO.from_buffer_copy(io)
    io2.Y[i] = None
    io2.F_latent[i] = F.data_ptr()
    # attached: the forward's own io (and its keep_products) drives this pass's backward, which then streams the kept
    # products back - they must be written now (the arena was sized for them: StepFn.forward's ``need``); a peek
    # leaves nothing behind
    io2.keep_products = io.keep_products if attach else 0
    o = _ops_mod.get_ops()
    scratch = o._ws(plan.scratch_bytes, saved)
    call = TO.stash(dict(lib=plan.lib, handle=plan.handle, prm=prm, io=io2))
    try:
        torch.ops.gpsa.step_forward(list(tensors), ins, [F], saved, scratch, call, 2 | 4 | (1 << (8 + i)))
    finally:
        TO.CALLS.pop(call, None)
    if attach:
        io.Y[i] = None
        io.F_latent[i] = F.data_ptr()
        rec["F_real"][i] = F  # (the io struct points into it)
    return F


class MaterializeFn(torch.autograd.Function):
    """partial sums of the fused pass -> the draws themselves; the backward leaves the draws' gradient where the
    step's backward looks for it (StepFn.backward: ``fuse["dF"]``) and walks on to the step's node"""

    @staticmethod
    def forward(ctx, rec, i, parts):
        ctx.rec, ctx.i, ctx.n = rec, i, parts.numel()
        ctx.dev = parts.device
        return materialize_values(rec, i, attach=True)

    @staticmethod
    def backward(ctx, dF):
        from .step_engine import placeholder_grad

        ctx.rec["dF"][ctx.i] = dF if (dF.dtype == torch.float32 and dF.is_contiguous()) else dF.float().contiguous()
        return None, None, placeholder_grad(ctx.dev, ctx.n)


class _ValuesOnlyFn(torch.autograd.Function):
    """identity whose backward refuses: the draws were materialised after loss_fn had taken the fused likelihood"""

    @staticmethod
    def forward(ctx, F, parts):
        return F.view_as(F)

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError(
            "GPSA: a gradient reached draws that were materialised AFTER loss_fn consumed them in fused form; use "
            "F_samples before calling loss_fn (the step then runs unfused for that modality) or set "
            "model.fuse_elbo = False")


def run_fused(rec, idx, Ys, parts):
    """loss_fn's half of the step: the data GP of the modalities ``idx`` with their likelihood folded in, on the
    observations ``Ys`` (gpsa_step_forward, stage 2 restricted to these modalities; forward left them out).
    ``parts``: their partial-sum outputs (held by the handles, not by ``rec``: rec -> outputs -> node -> rec would be
    a cycle that keeps the arena of a forward without a backward until the cyclic collector runs)"""
    live = rec.get("live")
    if live is None:
        raise RuntimeError("GPSA: loss_fn on the F_samples of a forward whose backward has already run")
    plan, io, prm, saved, tensors, ins = (live[k] for k in ("plan", "io", "prm", "saved", "tensors", "ins"))
    mask = 0
    for i, Y in zip(idx, Ys):
        io.Y[i] = Y.data_ptr()
        rec["Y"][i] = Y  # (alive until the backward has run: the io struct points into it)
        mask |= 1 << (8 + i)
    o = _ops_mod.get_ops()
    scratch = o._ws(plan.scratch_bytes, saved)
    outs = [p.detach() for p in parts]
    call = TO.stash(dict(lib=plan.lib, handle=plan.handle, prm=prm, io=io))
    try:
        torch.ops.gpsa.step_forward(list(tensors), ins + list(Ys) + [rec["noise"]], outs, saved, scratch, call, 2 | mask)
    finally:
        TO.CALLS.pop(call, None)
