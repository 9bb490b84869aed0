"""Training-loop helpers around the hot path: the reference's loop body
(examples/grid_example.py:62-78) as a function, and the same step captured once into a hipGraph and
replayed — ~470 kernel launches per step become one graph launch, which is what matters for the
launch-bound small configurations (config 1, S=1).
"""
import torch


_SEED = {}


def backward(loss):
    """``loss.backward()`` with the seed gradient (a one) kept on the device between steps: autograd otherwise
    launches a fill for it at every step."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _SEED.get(key)
    if one is None:
        one = _SEED[key] = torch.ones_like(loss)
    loss.backward(one)


def train_step(model, optimizer, data_dict, view_idx, Ns, S=5, reducer=None, static_grads=False):
    """forward(S) + loss_fn + backward + optimizer step; returns the loss tensor (no host sync).
    ``static_grads``: keep the .grad buffers (zeroed, accumulated into) instead of letting autograd hand
    over fresh ones: needed under graph capture, ~one extra launch per parameter otherwise."""
    Xs = {m: d["spatial_coords"] for m, d in data_dict.items()}
    out = model.forward(Xs, view_idx=view_idx, Ns=Ns, S=S)
    loss = model.loss_fn(data_dict, out[3])
    optimizer.zero_grad(set_to_none=not static_grads)
    backward(loss)
    if reducer is not None:
        reducer()
    optimizer.step()
    return loss


class Microbatches:
    """One optimiser step as K forward / ELBO / backward passes over row slices of every view, gradients
    accumulated: the "minibatched K_NM" of BASELINE.json's Slide-seq-scale configuration.

    Everything N-scaled is independent per spot given the M x M factors, so slice k contributes its spots'
    likelihood and the first slice the KL terms - summed over k that is the full negative ELBO and its gradient,
    the decomposition of K data-parallel ranks (parallel.shard_data_dict), only one after the other on ONE GPU.
    What it buys: the data GPs' products Omega_l alpha of a SLICE fit the device (L M C/K floats), so every slice
    takes the kept-products path - one full product per slice instead of the block-triangular form plus the
    recomputed alpha-gradient (BASELINE config 5 on one MI355X: 800 GB of products -> 8 slices of 100 GB).  What it
    costs: little - the slices share one M x M stage (the engine reuses the first slice's factorisations from the
    parked arena, ``gpsa_step_io.reuse_mm``; slices of another shape, e.g. a remainder, redo it)."""

    def __init__(self, model, data_dict, K, quantum=4):
        """``quantum``: slice sizes are multiples of it (the last slice of a view takes the remainder): the LDS-DMA
        kernels of the M > 256 data GP want 16-byte aligned rows of the [M, C] panels, i.e. C % 4 == 0"""
        self.model, self.K = model, int(K)
        if self.K < 1:
            raise ValueError(f"Microbatches: K = {K} slices")
        # every slice must hold rows OF EVERY MODALITY: a modality without rows in a slice drops that slice's data-GP
        # pass from its plan, and the accumulator the slices' backwards share (gpsa_step_io.bwd_acc) mirrors the
        # per-pass gradient pieces byte for byte across the slices' plans.  With sizes rounded up to ``quantum``,
        # slice k of a view starts at k * per: a modality's last slice is non-empty iff its largest view reaches it
        for mod, d in data_dict.items():
            biggest = max(int(n) for n in d["n_samples_list"])
            per_max = -(-(-(-biggest // self.K)) // quantum) * quantum
            if (self.K - 1) * per_max >= biggest:
                raise ValueError(f"Microbatches: K = {self.K} slices of multiples of {quantum} rows leave the last "
                                 f"slice(s) without rows of modality {mod!r} (largest view: {biggest} rows); use "
                                 f"K <= {max(1, -(-biggest // quantum))}")
        self.slices, self.bounds = [], []  # bounds[k][mod] = [(lo, hi) of slice k within view v]
        for k in range(self.K):
            dd = {}
            self.bounds.append({})
            for mod, d in data_dict.items():
                rows, new_ns, off = [], [], 0
                self.bounds[k][mod] = []
                for n in (int(x) for x in d["n_samples_list"]):
                    per = -(-n // self.K)
                    per = -(-per // quantum) * quantum
                    lo, hi = min(k * per, n), min((k + 1) * per, n)
                    self.bounds[k][mod].append((lo, hi))
                    rows.append(torch.arange(off + lo, off + hi))
                    new_ns.append(hi - lo)
                    off += n
                idx = torch.cat(rows).to(d["spatial_coords"].device)
                dd[mod] = {"spatial_coords": d["spatial_coords"][idx].contiguous(),
                           "outputs": d["outputs"][idx].contiguous(), "n_samples_list": new_ns}
            vi, Ns, _, _ = model.create_view_idx_dict(dd)
            self.slices.append((dd, vi, Ns))

    def _fold_buffer(self):
        """the accumulator the slices' backwards share (None: per-layer path / CPU: every slice closes by itself)"""
        model = self.model
        if len(self.slices) < 2 or not getattr(model, "use_step_engine", False) or not model.Xtilde.is_cuda:
            return None
        plans = list(model.__dict__.get("_step_plans", {}).values())
        need = max([p.bwd_acc_bytes for p in plans] or [0])
        if need <= 0:  # (first step: no plan yet - it closes per slice, the next one folds)
            return None
        buf = self.__dict__.get("_fold")
        if buf is None or buf.numel() < need:
            buf = self.__dict__["_fold"] = torch.empty(need, dtype=torch.uint8, device=model.Xtilde.device)
        return buf

    def _ensure_plans(self, S):
        """the slices' step plans exist before the first forward: the shared arena and the accumulator are sized from
        them (otherwise the first step runs with an arena per plan - twice 100 GB at BASELINE config 5)"""
        model = self.model
        if self.__dict__.get("_planned") == S or not getattr(model, "use_step_engine", False) or not model.Xtilde.is_cuda:
            return
        from . import step_engine as SE

        for dd, vi, Ns in self.slices:
            Xs = {m: d["spatial_coords"] for m, d in dd.items()}
            if not SE.eligible(model, Xs, vi, None):
                return
            rows = SE.view_rows(model, vi, Ns)
            if rows is None:
                return
            SE.get_plan(model, rows, S, None, want_kl=True)
        self.__dict__["_planned"] = S

    def _shared_arena(self):
        """ONE arena for all slices of a step (the largest any of their plans wants): slices of different shapes - the
        remainder - are different plans, and with an arena parked per plan the second plan found the device full and
        fell back to recomputing its products (BASELINE config 5: 100 GB per arena; one slice in eight ran the
        no-keep path, +220 ms per step) and redid the M x M stage it could have found in the first slice's arena"""
        model = self.model
        if len(self.slices) < 2 or not getattr(model, "use_step_engine", False) or not model.Xtilde.is_cuda:
            return None
        plans = list(model.__dict__.get("_step_plans", {}).values())
        if not plans:
            return None
        need = max(p.saved_bytes for p in plans)
        buf = self.__dict__.get("_arena")
        if buf is None or buf.numel() < need:
            self.__dict__["_arena"] = buf = None
            from .step_engine import release_arenas

            release_arenas(model)  # (the per-plan parked blocks of earlier steps make room)
            try:
                buf = self.__dict__["_arena"] = torch.empty(need, dtype=torch.uint8, device=model.Xtilde.device)
            except torch.OutOfMemoryError:
                return None
        return buf

    def step(self, optimizer, S=5, reducer=None, noise=None):
        """zero_grad, K accumulating passes, (all-reduce,) optimizer.step(); returns the summed loss (device tensor).
        ``noise``: per slice ``(eps_G, eps_F)`` for ``model.inject_noise`` (tests / reproducibility)."""
        model = self.model
        scale0 = model.kl_scale
        optimizer.zero_grad(set_to_none=True)
        total = None
        self._stepno = getattr(self, "_stepno", 0) + 1
        try:
            # every slice runs on the same parameters: the engine keeps the M x M stage of the first slice (same plan,
            # same parked arena) for the others and CLOSES the backward once, with the last slice (gpsa_step_io.bwd_acc:
            # the slices' N-scaled gradient pieces meet in an accumulator; KL backward, the prior covariances' backward,
            # dOmega -> dA and the finalisation run once per step instead of once per slice); the KL terms are charged
            # once, to the LAST slice - the one whose backward closes
            model.__dict__["_mm_epoch"] = (id(self), self._stepno)
            self._ensure_plans(S)
            fold = self._fold_buffer()
            model.__dict__["_mb_arena"] = self._shared_arena()
            for k, (dd, vi, Ns) in enumerate(self.slices):
                last = k == len(self.slices) - 1
                model.kl_scale = scale0 if last else 0.0
                model.__dict__["_bwd_acc"] = None if fold is None else (fold, 3 if last else (1 if k == 0 else 2))
                if noise is not None:
                    model.inject_noise(noise[k][0], noise[k][1], None)
                out = model.forward({m: d["spatial_coords"] for m, d in dd.items()}, view_idx=vi, Ns=Ns, S=S)
                loss = model.loss_fn(dd, out[3])
                backward(loss)  # .grad accumulates across the slices
                total = loss.detach() if total is None else total + loss.detach()
                del out, loss
        finally:
            model.kl_scale = scale0
            model.__dict__["_mm_epoch"] = None
            model.__dict__["_bwd_acc"] = None
            model.__dict__["_mb_arena"] = None
        if reducer is not None:
            reducer()
        optimizer.step()
        return total


def fit(model, data_dict, n_epochs, lr=1e-2, S=5, optimizer=None, checker=None, sync_every=10,
        callback=None, graphed=False):
    """The reference's training loop (examples/grid_example.py:59-78 and the convergence test of
    gpsa/util/util.py:257-278 used by the experiment scripts) as one call.

    Runs ``n_epochs`` steps of forward(S) + loss_fn + backward + Adam on ``data_dict``; returns the loss
    trace (a list of floats, one per step).  The loss stays on the device and is brought to the host
    every ``sync_every`` steps only, so the queue is not drained at each iteration; ``checker`` (a
    ``LossNotDecreasingChecker``) is evaluated on the synced values and stops the loop early;
    ``callback(step, model, loss_trace)`` is called at every sync.  ``graphed=True`` replays the step as
    one hipGraph (single GPU)."""
    model.train()
    view_idx, Ns, _, _ = model.create_view_idx_dict(data_dict)
    if optimizer is None:
        ps = list(model.parameters())
        if all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
            from .optim import FusedAdam  # torch.optim.Adam's update as one HIP launch (capturable)

            optimizer = FusedAdam(ps, lr=lr)
        else:  # CPU, model.double(), ...: what the reference's loop uses
            optimizer = torch.optim.Adam(ps, lr=lr, capturable=bool(graphed))
    stepper = GraphedTrainStep(model, optimizer, data_dict, view_idx, Ns, S=S) if graphed else None
    trace, pending = [], []
    # what exists by now (torch, the model, the plans) is long-lived: out of the cyclic collector's way for the loop.
    # A generation-2 collection over that heap costs 40-100 ms (round 6: the one slow block of every bench run, at the
    # process's ~45th step, was exactly that: `gc: done, 73888 unreachable, 0.10 s`) - 15 steps of the headline problem
    import gc

    gc.collect()
    gc.freeze()

    def drain():
        trace.extend(float(v) for v in torch.stack(pending).tolist())
        pending.clear()

    for it in range(n_epochs):
        loss = stepper.step() if graphed else train_step(model, optimizer, data_dict, view_idx, Ns, S)
        pending.append(loss.detach().clone() if graphed else loss.detach())
        if (it + 1) % sync_every == 0 or it + 1 == n_epochs:
            first = len(trace)
            drain()
            if graphed:
                stepper.check()
            if callback is not None:
                callback(it, model, trace)
            if checker is not None and any(checker.check_loss(i, trace) for i in range(first, len(trace))):
                break
    gc.unfreeze()
    return trace


class GraphedTrainStep:
    """One training step captured into a hipGraph (torch.cuda.CUDAGraph) and replayed.

    * every kernel of the step (HIP kernels of this package, Adam, RNG) is inside the graph; replaying
      does all the work of an eager step on the same static buffers;
    * the per-forward numerics check cannot sync inside a capture: every replay folds its flags (a
      non-positive pivot, a non-positive warp variance, a non-finite loss) into a STICKY device word, and
      ``check()`` raises if any step since the last check tripped it (call it every N steps);
    * the optimizer must be capturable (``optim.FusedAdam`` or ``torch.optim.Adam(..., capturable=True)``);
    * ``reducer`` (parallel.GradAllReducer): the data-parallel all-reduce of the gradients is captured too -
      RCCL collectives are capturable - so every rank of a sharded job replays ONE graph per step.
    """

    def __init__(self, model, optimizer, data_dict, view_idx, Ns, S=5, warmup=3, reducer=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainStep needs a HIP device")
        self.model, self.optimizer = model, optimizer
        saved = (model.check_numerics, model.overlap_views)
        model.check_numerics = False
        model.overlap_views = True  # independent views become parallel branches of the graph
        self.sticky = torch.zeros((), dtype=torch.int32, device=next(model.parameters()).device)
        # drop every reference to an earlier autograd graph (capture needs fresh AccumulateGrad nodes
        # on the capture stream)
        model._cache = None
        model.F_latent_samples, model.F_observed_samples = {}, {}
        for p in model.parameters():
            p.grad = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up off the default stream, as torch's capture rules ask
            for _ in range(warmup):
                train_step(model, optimizer, data_dict, view_idx, Ns, S, reducer=reducer, static_grads=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # with a collective inside, other threads (the process group's watchdog polling its events) must stay
        # legal during the capture: thread-local capture mode
        mode = {"capture_error_mode": "thread_local"} if reducer is not None else {}
        with torch.cuda.graph(self.graph, **mode):
            self.loss = train_step(model, optimizer, data_dict, view_idx, Ns, S, reducer=reducer, static_grads=True)
            flags = [f.reshape(-1).to(torch.int32) for f in model._cache.flags]
            # the step engine folds its Cholesky infos and variance flags into ONE device word (check_numerics is
            # off inside the capture, so nobody else reads it): a non-positive pivot is replaced by 1 and trains on
            # finite garbage - this word is the only trace of it
            self.engine_flag = getattr(model._cache, "engine_flag", None)
            if self.engine_flag is not None:
                flags.append(self.engine_flag.reshape(-1).to(torch.int32))
            flags.append((~torch.isfinite(self.loss.detach())).reshape(-1).to(torch.int32))
            self.sticky.copy_(torch.maximum(self.sticky, torch.cat(flags).abs().max()))
        # eager forwards after this keep their own numerics check and stream behaviour
        model.check_numerics, model.overlap_views = saved

    def step(self):
        self.graph.replay()
        return self.loss

    def check(self):
        """host sync: raise if ANY replay since the last check hit a non-PD covariance, a non-positive warp
        variance or a non-finite loss (the flag is sticky across replays and reset here)"""
        bad = int(self.sticky.item())
        if bad != 0:
            self.sticky.zero_()
            raise torch.linalg.LinAlgError(
                "GPSA graphed step: non-positive-definite covariance, non-positive warp variance or "
                "non-finite loss in a step since the last check")
