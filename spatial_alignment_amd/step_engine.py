"""Host side of the C++ step engine (csrc/step.hip, declared in include/gpsa_hip.h).

``forward`` of the model becomes ONE autograd node whose forward and backward are one C call each
(``gpsa_step_forward`` / ``gpsa_step_backward``): the launch sequence of the whole step is enqueued from
C++, the warp GPs of all free views share their launches, and every parameter gradient is accumulated in
fp64 and rounded once.  ``loss_fn`` is a second node (``gpsa_elbo_loss_fwd`` / ``_bwd``).

PyTorch remains plumbing: it owns the tensors (parameters, outputs, the two arenas), the stream and the
autograd bookkeeping between the two nodes and the optimiser.
"""
import ctypes as C
import os

import torch

from . import _lib
from . import ops as _ops_mod
from . import torch_ops as TO
from .kernels import builtin_kind

# the flat gradient buffer of the most recent backward per device (its views are the parameters' .grad):
# parallel.GradAllReducer reduces it in place
STATS = {"mm_reused": 0}  # forwards that skipped their M x M stage (io.reuse_mm): tests look at it
LAST_FLAT = {}
LAST_USED = {}  # floats of LAST_FLAT the gradient views span (each view starts on a 256-byte boundary)

MAXM = _lib.MAX_MODS
KINDS = _ops_mod.KINDS
_raw_stream = torch._C._cuda_getCurrentRawStream


def _p(t):
    return 0 if t is None else t.data_ptr()


class StepPlan:
    """a ``gpsa_step_create`` handle + the shape it was made for"""

    def __init__(self, lib, key, desc, keep):
        self.lib, self.key, self._keep = lib, key, keep
        self.handle = lib.gpsa_step_create(C.byref(desc))
        if not self.handle:
            raise _lib.GpsaHipError(f"gpsa_step_create refused the problem description {key}")
        self.saved_bytes = int(lib.gpsa_step_saved_bytes(self.handle))
        self.saved_bytes_nokeep = int(lib.gpsa_step_saved_bytes_nokeep(self.handle))
        self.scratch_bytes = int(lib.gpsa_step_scratch_bytes(self.handle))
        self.bwd_acc_bytes = int(lib.gpsa_step_bwd_acc_bytes(self.handle))
        self.n_kl = int(lib.gpsa_step_n_kl(self.handle))
        self.eps_g_numel = int(lib.gpsa_step_eps_g_numel(self.handle))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.gpsa_step_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# Arenas beyond this size are parked on their plan between a backward and the next forward instead of going back
# to torch's caching allocator: a 160 GB block (BASELINE config 4's kept products) that other allocations have split
# in the meantime cannot be had again - the allocator reports it as "reserved but unallocated" and fails.
ARENA_PARK_BYTES = 1 << 30


def _take_arena(plan, nbytes, dev, must=False, siblings=None):
    """-> arena, or None when it cannot be had.  ``plan._took_parked`` says whether it is the plan's own parked block
    (nobody else has written into it since it was handed back: its contents are the previous forward's).
    ``siblings``: the model's other plans - their parked blocks are let go before an allocation is given up."""
    plan._took_parked = False
    parked = plan.__dict__.get("_parked")
    # Under stream capture the arena's address is baked into the graph: it must come from the graph's private pool, not
    # be an eagerly allocated block that an eager forward between two replays could take from the plan and free
    # (later replays would then write into memory other tensors own)
    capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if parked is not None and not capturing and parked.device == dev and parked.numel() >= nbytes:
        if 2 * nbytes < parked.numel():  # a forward without kept products (no backward will hand the arena back):
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)  # leave the large one parked
        plan._parked = None
        plan._took_parked = True
        return parked
    if not capturing:
        plan._parked = None  # (too small: let it go before asking for the larger one)
    del parked
    try:
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    except torch.OutOfMemoryError:
        # memory may sit idle on a sibling plan of the same model (the S = 1 leg, prediction_mode, a remainder
        # microbatch shape): let every parked block go and ask once more
        freed = False
        for other in (siblings or ()):
            if other.__dict__.get("_parked") is not None and not capturing:
                other._parked = None
                freed = True
        if freed:
            torch.cuda.empty_cache()
            try:
                return torch.empty(nbytes, dtype=torch.uint8, device=dev)
            except torch.OutOfMemoryError:
                pass
        if must:
            raise
        return None


def _give_arena(plan, arena, force=False):
    """``force``: park whatever its size (the slices of a microbatched step share the arena's M x M stage)"""
    if arena is None or (arena.is_cuda and torch.cuda.is_current_stream_capturing()):
        return  # (a captured step's arena belongs to the graph's pool)
    if (force or arena.numel() >= ARENA_PARK_BYTES) and plan.__dict__.get("_parked") is None:
        plan._parked = arena


def release_arenas(model):
    """let go of every arena parked on the model's plans (VariationalGPSA.release_arenas; eval() calls it)"""
    for plan in model.__dict__.get("_step_plans", {}).values():
        plan.__dict__["_parked"] = None


def eligible(model, X_spatial, view_idx, G_test):
    """the engine covers: built-in covariance functions, fp32 HIP tensors, views that are consecutive row
    blocks covering each modality (what create_view_idx_dict produces), <= 4 modalities"""
    if builtin_kind(model.kernel_func_warp) is None or builtin_kind(model.kernel_func_data) is None:
        return False
    mods = model.modality_names
    if len(mods) > MAXM or model.n_spatial_dims > 4:
        return False
    if not model.Xtilde.is_cuda or model.Xtilde.dtype != torch.float32:
        return False
    for m in mods:
        x = X_spatial[m]
        if not x.is_cuda or x.dtype != torch.float32 or x.requires_grad:
            return False
        if G_test is not None and G_test[m].requires_grad:
            return False
    return True


def view_rows(model, view_idx, Ns):
    """rows of view v in modality m when the views are consecutive row blocks covering 0..N, else None"""
    import numpy as np

    V = model.n_views
    # a training loop passes the same index objects every step.  The answer is remembered only for what cannot change
    # under the memo's feet: a tensor is keyed by identity, storage and autograd version counter (every in-place write
    # bumps ``_version``); an ndarray has no such counter, so its CONTENT is compared with the block it stood for on
    # every hit (one vectorised compare: ~5 us for 10^4 rows); anything else (lists) is walked again each call.
    # The engine path ignores view_idx altogether, so a stale "consecutive blocks" answer would silently assign rows to
    # the wrong views.
    items = [view_idx[m][v] for m in model.modality_names for v in range(V)]
    if not all(torch.is_tensor(x) or isinstance(x, np.ndarray) for x in items):
        return _view_rows(model, view_idx, Ns, np)
    memo = model.__dict__.setdefault("_view_rows_memo", {})
    key = tuple((id(x), len(x), x._version, x.data_ptr()) if torch.is_tensor(x) else (id(x), len(x)) for x in items) + \
        tuple(int(Ns[m]) for m in model.modality_names)
    hit = memo.get(key)
    if hit is not None:
        res, _, blocks = hit
        if all(b is None or (x.shape == b.shape and np.array_equal(x, b)) for x, b in zip(items, blocks)):
            return res
        del memo[key]
    res = _view_rows(model, view_idx, Ns, np)
    if len(memo) >= 8:
        memo.clear()
    # (the objects are held alive by the memo so that an id cannot come back on another object; an ndarray is stored
    # next to a private copy of what it held when it was validated)
    memo[key] = (res, items, [None if torch.is_tensor(x) else np.array(x, copy=True) for x in items])
    return res


def _view_rows(model, view_idx, Ns, np):
    V = model.n_views
    out = []
    for m in model.modality_names:
        edge = 0
        for v in range(V):
            idx = view_idx[m][v]
            a = np.asarray(idx.cpu() if torch.is_tensor(idx) else idx)
            n = int(a.shape[0])
            if n > 0 and (int(a[0]) != edge or int(a[-1]) != edge + n - 1 or
                          (n > 1 and not bool(np.all(np.diff(a) == 1)))):
                return None
            out.append(n)
            edge += n
        if edge != int(Ns[m]):
            return None
    return tuple(out)


def keep_budget_bytes(model, desc, keep_gb=None):
    """HBM the plan may spend on the data GPs' kept products (gpsa_step_desc.keep_budget_bytes).

    ``model.keep_budget_gb`` (or GPSA_KEEP_GB) when set; otherwise what this process can still get from the device -
    free memory plus the blocks torch's allocator holds unused - minus the plan's own arenas, minus three more copies
    of the parameters (their gradient and Adam's two moments may not exist yet), with a tenth of the device left
    alone: a 288 GB MI355X keeps the 160 GB of BASELINE config 4, a smaller or shared device falls back to recomputing
    instead of dying in the allocator."""
    import os

    if keep_gb is None and os.environ.get("GPSA_KEEP_GB"):
        keep_gb = float(os.environ["GPSA_KEEP_GB"])
    if keep_gb is not None:
        return max(1, int(float(keep_gb) * 2**30)) if float(keep_gb) > 0 else -1
    dev = model.Xtilde.device
    free, total = torch.cuda.mem_get_info(dev)
    avail = free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    out = (C.c_longlong * 8)()
    desc.keep_budget_bytes = -1
    if _lib.load().gpsa_step_describe(C.byref(desc), out) != 0:
        return -1
    # memory this model's own plans hold idle is memory this plan can have: arenas parked on sibling plans are let go
    # on demand (_take_arena), and a microbatched step's slices share ONE arena (train.Microbatches) whatever their plan
    held = sum(q.__dict__["_parked"].numel() for q in model.__dict__.get("_step_plans", {}).values()
               if q.__dict__.get("_parked") is not None)
    shared = model.__dict__.get("_mb_arena")
    held += shared.numel() if shared is not None else 0
    params = sum(p.numel() * p.element_size() for p in model.parameters())
    budget = int(avail + held - 0.1 * total) - int(out[6]) - int(out[1]) - 3 * params
    return budget if budget > 0 else -1


def kl_own_range(model):
    """[lo, hi) of the KL terms this process evaluates (``model.kl_owner = (rank, world)``: a data-parallel rank owns a
    contiguous share of the V*D + sum L_m terms, weight 1; the gradient all-reduce sums the shares - parallel.py), or
    None: every term (one process, or the 1/world weighting of ``kl_scale``).  Term order: Omega_G rows r = j*V + v,
    then every modality's outputs (vgpsa.py:498-530)."""
    owner = getattr(model, "kl_owner", None)
    if owner is None:
        return None
    rank, world = int(owner[0]), int(owner[1])
    if world <= 1:
        return None
    n = model.n_views * model.n_spatial_dims + sum(int(model.n_latent_outputs[m]) for m in model.modality_names)
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    # (hi > 0 says "a range" to gpsa_step_desc; a rank left without a term - more ranks than terms - gets the empty
    #  range [n, n))
    return (lo, hi) if hi > lo else (n, n)


def get_plan(model, rows, S, test_shapes, want_kl):
    """plan for this model / data shape (cached on the model)"""
    mods = model.modality_names
    V, D = model.n_views, model.n_spatial_dims
    fixed = tuple(1 if model._is_fixed(v) else 0 for v in range(V))
    L = tuple(int(model.n_latent_outputs[m]) for m in mods)
    P = tuple(int(model.Ps[m]) for m in mods)
    lmc = tuple(1 if model.n_latent_gps[m] is not None else 0 for m in mods)
    N = tuple(sum(rows[i * V:(i + 1) * V]) for i in range(len(mods)))
    s_test = test_shapes[0] if test_shapes else 0
    n_test = tuple(test_shapes[1]) if test_shapes else tuple(0 for _ in mods)
    kw, kd = KINDS[builtin_kind(model.kernel_func_warp)], KINDS[builtin_kind(model.kernel_func_data)]
    keep_gb = getattr(model, "keep_budget_gb", None)
    exact = getattr(model, "exact_inducing_grad", None)
    if exact is None and os.environ.get("GPSA_EXACT_GRAD") in ("0", "1"):
        exact = os.environ["GPSA_EXACT_GRAD"] == "1"
    if exact is None:  # the default: every gradient within 1e-4 of the reference's fp64 run
        exact = True
    exact = int(bool(exact))
    own = kl_own_range(model)
    key = (V, D, len(mods), int(S), int(model.Xtilde.shape[1]), int(model.Gtilde.shape[0]), kw, kd, L, P, lmc, N,
           s_test, n_test, int(bool(want_kl)), fixed, rows, model.Xtilde.device.index, keep_gb, exact, own)
    cache = model.__dict__.setdefault("_step_plans", {})
    plan = cache.get(key)
    if plan is not None:
        return plan
    d = _lib.StepDesc()
    d.n_views, d.n_dims, d.n_mods, d.n_samples = V, D, len(mods), int(S)
    d.m_x, d.m_g, d.kind_warp, d.kind_data = key[4], key[5], kw, kd
    for i in range(len(mods)):
        d.n_latent[i], d.n_out[i], d.has_lmc[i], d.n_rows[i], d.n_test[i] = L[i], P[i], lmc[i], N[i], n_test[i]
    d.s_test, d.want_kl = int(s_test), int(bool(want_kl))
    vf = (C.c_int * V)(*fixed)
    vr = (C.c_longlong * len(rows))(*rows)
    d.view_fixed, d.view_rows = vf, vr
    d.exact_inducing_grad = exact
    d.kl_own_lo, d.kl_own_hi = own if own is not None else (0, 0)
    d.keep_budget_bytes = keep_budget_bytes(model, d, keep_gb)
    with torch.cuda.device(model.Xtilde.device):
        plan = StepPlan(_lib.load(), key, d, (vf, vr))
    plan.mods, plan.V, plan.D, plan.S, plan.L, plan.P, plan.lmc, plan.N = mods, V, D, int(S), L, P, lmc, N
    plan.s_test, plan.n_test, plan.fixed, plan.rows = s_test, n_test, fixed, rows
    plan.exact, plan.kl_own = bool(exact), own  # (what the plan was built with: bench.py reports the timed mode)
    cache[key] = plan
    return plan


def _param_list(model):
    """the tensors the step differentiates, in gpsa_step_params order"""
    mods = model.modality_names
    ps = [model.Xtilde, model.delta_G_list, model.Omega_sqt_G_list, model.warp_kernel_lengthscales,
          model.warp_kernel_variances, model.Gtilde, model.data_kernel_lengthscale, model.data_kernel_variance]
    ps += [model.Omega_sqt_F_dict[m] for m in mods]
    ps += [model.delta_F_dict[m] for m in mods]
    ps += [model.W_dict[m] for m in mods if model.n_latent_gps[m] is not None]
    return ps


def _fill_params(st, tensors, model, plan):
    nm = len(plan.mods)
    (st.Xtilde, st.delta_G, st.Omega_sqt_G, st.warp_ls, st.warp_var, st.Gtilde, st.data_ls,
     st.data_var) = (_p(t) for t in tensors[:8])
    for i in range(nm):
        st.Omega_sqt_F[i] = _p(tensors[8 + i])
        st.delta_F[i] = _p(tensors[8 + nm + i])
    k = 8 + 2 * nm
    for i in range(nm):
        if plan.lmc[i]:
            st.W[i] = _p(tensors[k])
            k += 1
    return st


_PLACEHOLDER = {}


def placeholder_grad(dev, n):
    """a gradient of the right shape and type for a fused modality's partial sums, so that autograd walks on to the
    step's node; never read (the real upstream gradients travel through the ``fuse`` record)"""
    key = (dev, n)
    d = _PLACEHOLDER.get(key)
    if d is None:
        d = _PLACEHOLDER[key] = torch.zeros(1, dtype=torch.float64, device=dev).expand(n)
    return d


class StepFn(torch.autograd.Function):
    """(parameters) -> G_means[m].., G_samples[m].., F_latent[m].., F_obs[m] (LMC).., test draws.., kl [T].
    ``aux``: everything that is not differentiated (plan, coordinates, draws, stream policy)."""

    @staticmethod
    def forward(ctx, aux, *tensors):
        plan, model = aux["plan"], aux["model"]()
        lib = plan.lib
        dev = tensors[0].device
        nm, S, D = len(plan.mods), plan.S, plan.D
        f32 = torch.float32
        for t in tensors:
            if t.dtype != f32 or not t.is_contiguous():
                raise _lib.GpsaHipError("step engine: parameters must be contiguous fp32 tensors")
        o = _ops_mod.get_ops()
        prm = _fill_params(_lib.StepParams(), tensors, model, plan)
        prm.slopes, prm.intercepts = _p(aux["slopes"]), _p(aux["intercepts"])
        io = _lib.StepIO()
        empty = lambda *sh: torch.empty(*sh, dtype=f32, device=dev)
        outs = {"Gm": [], "Gs": [], "Fl": [], "Fo": [], "Flt": [], "Fot": []}
        # fused ELBO (aux["fuse"], set up by VariationalGPSA.forward in training): modality i's data GP is left for
        # loss_fn, which knows the observations (lazy.run_fused) - its "F_latent" output is the vector of partial sums
        # the likelihood finishes from, the draws are never materialised
        fuse = aux.get("fuse")
        fused = fuse["mods"] if fuse is not None else [False] * nm
        lazy_obs = aux.get("lazy_obs") or [False] * nm
        if fuse is not None:
            io.fuse_elbo = 1
            nparts = int(lib.gpsa_quadform_elbo_parts())
        for i, m in enumerate(plan.mods):
            N, L, P = plan.N[i], plan.L[i], plan.P[i]
            io.X[i] = _p(aux["X"][i])
            io.eps_F[i] = _p(aux["eps_F"][i])
            Gm, Gs = empty(N, D), empty(S, N, D)
            if fused[i]:
                Fl = torch.empty(nparts, dtype=torch.float64, device=dev)
                io.noise_u[i], io.ll_part[i] = fuse["noise_ptr"][i], _p(Fl)  # (io.Y[i]: set by loss_fn)
                # the draws themselves leave the fused pass too, transposed ([L, S N]: 20 MB at the headline size, two
                # more 64-byte stores per output and wave): what the handle shows to anyone who looks after loss_fn
                fuse["FT"][i] = torch.empty(L, S * N, dtype=f32, device=dev)
                io.F_fused_T[i] = _p(fuse["FT"][i])
            else:
                Fl = empty(S, N, L)
                io.F_latent[i] = _p(Fl)
            outs["Gm"].append(Gm); outs["Gs"].append(Gs); outs["Fl"].append(Fl)
            io.G_means[i], io.G_samples[i] = _p(Gm), _p(Gs)
            if plan.lmc[i]:
                # training through loss_fn: F_obs = F_latent W is not formed (lazy.LazyProduct forms it on demand,
                # loss_fn runs gpsa_lmc_loglik_fused_f32 instead); the output slot stays, empty
                Fo = torch.empty(0, dtype=f32, device=dev) if lazy_obs[i] else empty(S, N, P)
                outs["Fo"].append(Fo)
                io.F_obs[i] = 0 if lazy_obs[i] else _p(Fo)
            if plan.s_test:
                io.G_test[i], io.eps_F_test[i] = _p(aux["G_test"][i]), _p(aux["eps_F_test"][i])
                Flt = empty(plan.s_test, plan.n_test[i], L)
                outs["Flt"].append(Flt)
                io.F_latent_test[i] = _p(Flt)
                if plan.lmc[i]:
                    Fot = empty(plan.s_test, plan.n_test[i], P)
                    outs["Fot"].append(Fot)
                    io.F_obs_test[i] = _p(Fot)
        io.eps_G = _p(aux["eps_G"])
        mu_z = empty(plan.V, model.Xtilde.shape[1], D)
        kl = torch.empty(plan.n_kl, dtype=torch.float64, device=dev) if aux["want_kl"] else None
        # the numerics word: with a check requested the engine's reduction writes it STRAIGHT into a pinned host word
        # (device-mapped: no copy launch), otherwise into a device word nobody reads
        if aux["check"]:
            flag = model._flag_host(aux.get("flag_slot", 0) if aux["check"] == "deferred" else 0)
        else:
            flag = torch.empty(1, dtype=torch.int32, device=dev)
        io.mu_z, io.kl, io.flag = _p(mu_z), _p(kl), _p(flag)
        # training (a backward will follow): the data GPs keep their products Omega_l alpha in the arena and the
        # backward streams them back; otherwise the cheaper forward and the smaller arena
        keep = any(ctx.needs_input_grad[1:]) and not aux.get("no_keep", False)  # (all False under no_grad)
        if fuse is not None and all(fused):  # nothing left that would stream kept products back
            keep = False
        io.keep_products = 1 if keep else 0
        sib = [q for q in model.__dict__.get("_step_plans", {}).values() if q is not plan]
        need = plan.saved_bytes if keep else plan.saved_bytes_nokeep
        shared = aux.get("shared_arena")  # a microbatched step's slices run in ONE arena, whatever their plan
        if shared is not None and shared.device == dev and shared.numel() >= need:
            saved, plan._took_parked = shared, True
        else:
            shared = None
            saved = _take_arena(plan, need, dev, siblings=sib)
        if saved is None:  # the device cannot hold the kept products after all (other tenants, fragmentation)
            keep = False
            io.keep_products = 0
            saved = _take_arena(plan, plan.saved_bytes_nokeep, dev, must=True, siblings=sib)
        scratch = o._ws(plan.scratch_bytes, saved)
        # the slices of one microbatched step (train.Microbatches) run on the same parameters: when this forward got
        # the very arena the previous slice filled, the M x M stage (factorisations, inverses, KL terms) is still in it.
        # The stage sits at the head of the arena at offsets that depend on (V, D, M, L, kinds, fixed views) only, not
        # on the slice's row counts: a slice of another shape (the remainder) - another plan - finds it there too
        epoch = aux.get("mm_epoch")
        k = plan.key
        stage_sig = (k[0], k[1], k[2], k[4], k[5], k[6], k[7], k[8], k[10], k[14], k[15])
        token = (saved.data_ptr(), saved.numel(), epoch, tuple(t.data_ptr() for t in tensors), stage_sig)
        io.reuse_mm = 1 if (epoch is not None and plan._took_parked
                            and model.__dict__.get("_mm_token") == token) else 0
        model.__dict__["_mm_token"] = token if epoch is not None else None
        ctx.shared_arena = shared is not None
        STATS["mm_reused"] += int(io.reuse_mm)
        pending = None
        # the C entry point is reached through the dispatcher (torch.ops.gpsa.step_forward, torch_ops.py): the
        # pointer structs travel as a key, the tensors the call reads and writes as its arguments
        flat_outs = [t for k in ("Gm", "Gs", "Fl", "Fo", "Flt", "Fot") for t in outs[k]] + [mu_z, flag]
        if kl is not None:
            flat_outs.append(kl)
        ins = [t for t in aux["X"] + list(aux["eps_F"]) + [aux["eps_G"], aux["slopes"], aux["intercepts"]]
               + list(aux["G_test"] or []) + list(aux["eps_F_test"] or [])
               if t is not None]
        call = TO.stash(dict(lib=lib, handle=plan.handle, prm=prm, io=io))
        # stage 2 of THIS call: every data GP, or - with fused modalities - only the ones that cannot fuse
        rest = sum(1 << (8 + i) for i in range(nm) if not fused[i]) if fuse is not None else 0

        def run(stages):
            if fuse is not None and (stages & 2):
                stages = (stages & 1) | ((2 | rest) if rest else 0)
            if stages:
                torch.ops.gpsa.step_forward(list(tensors), ins, flat_outs, saved, scratch, call, stages)

        try:
            if aux["check"] == "deferred":
                # training: the word is shipped behind an event as below, but nobody waits for it inside forward -
                # the backward of this node does, before it touches a gradient (see StepFn.backward)
                run(1)
                aux["deferred"] = model._post_flag(flag)
                run(2)
            elif aux["check"]:
                # the flag depends on the factorisations and the warp GPs only: ship it to the host behind an event
                # BEFORE the data GPs are queued, so that the check waits for the short part of the forward and the
                # host keeps queueing while the long part runs
                run(1)
                pending = model._post_flag(flag)
                run(2)
            else:
                run(3)
        finally:
            TO.CALLS.pop(call, None)
        if fuse is not None:
            # what lazy.materialize_values needs to run a fused modality's data GP again with the separate kernels
            # (no reference to ``aux`` itself: aux -> fuse -> live -> aux would keep the arena until the cyclic collector)
            fuse["live"] = dict(plan=plan, io=io, prm=prm, saved=saved, tensors=tensors,
                                ins=[t for t in aux["X"] + list(aux["eps_F"]) + [aux["eps_G"]] if t is not None])
        aux["pending"], aux["mu_z"], aux["flag"] = pending, mu_z, flag
        # what a SECOND backward through this node (retain_graph=True) needs to fill a fresh arena again: the call's
        # tensor lists (aliases without a grad_fn: an output held here would close a cycle node -> ctx -> output)
        aux["rerun"] = dict(ins=ins, outs=[t.detach() for t in flat_outs], keep=bool(keep), rest=rest)
        import weakref

        aux["arena_ref"] = weakref.ref(saved)
        ctx.aux, ctx.io, ctx.prm = aux, io, prm
        ctx.arena = saved
        flat = outs["Gm"] + outs["Gs"] + outs["Fl"] + outs["Fo"] + outs["Flt"] + outs["Fot"]
        # the outputs travel as SAVED tensors (F_latent is an input of the backward - dW of the LMC -, the io struct
        # points into all of them): as plain attributes they would close a cycle output -> grad_fn -> ctx -> output
        # and every forward whose backward never runs would wait for the cyclic collector with its arena
        ctx.save_for_backward(*tensors, *flat)
        ctx.n_in = len(tensors)
        ctx.layout = [len(outs[k]) for k in ("Gm", "Gs", "Fl", "Fo", "Flt", "Fot")]
        ctx.set_materialize_grads(False)
        if kl is None:
            return tuple(flat)
        return tuple(flat) + (kl,)

    @staticmethod
    def backward(ctx, *gouts):
        aux = ctx.aux
        plan, model = aux["plan"], aux["model"]()
        tensors = ctx.saved_tensors[: ctx.n_in]  # (a second backward without retain_graph=True: torch raises here)
        if ctx.arena is None:
            StepFn._refill(ctx, aux, plan, model, tensors)
        # the forward's numerics word: waited for AFTER this backward's launches are queued (below), raised before
        # any gradient is handed to autograd
        pend_check, aux["deferred"] = aux.get("deferred"), None
        lib = plan.lib
        dev = tensors[0].device
        nm = len(plan.mods)
        f32 = torch.float32
        o = _ops_mod.get_ops()
        og = _lib.StepOutGrads()
        keep = []

        def grad_ptr(g):
            if g is None:
                return 0
            g = g if (g.dtype == f32 and g.is_contiguous()) else g.to(f32).contiguous()
            keep.append(g)
            return g.data_ptr()

        k = 0
        nGm, nGs, nFl, nFo, nFlt, nFot = ctx.layout
        lmc_idx = [i for i in range(nm) if plan.lmc[i]]
        for i in range(nGm):
            og.dG_means[i] = grad_ptr(gouts[k]); k += 1
        for i in range(nGs):
            og.dG_samples[i] = grad_ptr(gouts[k]); k += 1
        fuse = aux.get("fuse")
        io = ctx.io
        for i in range(nFl):
            if fuse is not None and fuse["mods"][i]:
                # the gradient that arrives for the partial sums is a placeholder; what the modality's draws were used
                # for decides (lazy.py): "fused" - loss_fn took the fused likelihood: the loss's upstream gradient was
                # left by ElboLossFn.backward (a device scalar);  "real" - the draws were materialised before loss_fn
                # saw them: an unfused modality, its draws' gradient was left by MaterializeFn.backward;  "lazy" -
                # nobody touched them: no gradient reaches this modality's data GP
                state = fuse["state"][i]
                if state == "fused" and fuse.get("gloss") is not None:
                    keep.append(fuse["gloss"])
                    og.gloss = fuse["gloss"].data_ptr()
                elif state == "real":
                    og.dF_latent[i] = grad_ptr(fuse["dF"][i])
                else:
                    io.Y[i] = None  # (the engine then sees an unfused pass without a gradient and skips it)
            else:
                og.dF_latent[i] = grad_ptr(gouts[k])
            k += 1
        for j in range(nFo):
            og.dF_obs[lmc_idx[j]] = grad_ptr(gouts[k]); k += 1
        for i in range(nFlt):
            og.dF_latent_test[i] = grad_ptr(gouts[k]); k += 1
        for j in range(nFot):
            og.dF_obs_test[lmc_idx[j]] = grad_ptr(gouts[k]); k += 1
        if aux["want_kl"]:
            gk = gouts[k]
            if gk is not None:
                gk = gk if (gk.dtype == torch.float64 and gk.is_contiguous()) else gk.double().contiguous()
                keep.append(gk)
                og.dkl = gk.data_ptr()
        # ONE flat fp32 buffer for every parameter gradient: the gradients handed to autograd are views of it
        # (a ready-made all-reduce bucket and a single region for the optimiser to stream through)
        # (every view starts on a 256-byte boundary: the kernels that write whole gradients use 16-byte stores)
        sizes = [t.numel() for t in tensors]
        offs, used = [], 0
        for n in sizes:
            offs.append(used)
            used += (n + 63) // 64 * 64
        flat = torch.empty(used + 64, dtype=f32, device=dev)  # spare room: see parallel.GradAllReducer
        views = [flat[o: o + n] for o, n in zip(offs, sizes)]
        grads = _lib.StepParamGrads()
        (grads.Xtilde, grads.delta_G, grads.Omega_sqt_G, grads.warp_ls, grads.warp_var, grads.Gtilde, grads.data_ls,
         grads.data_var) = (_p(v) for v in views[:8])
        for i in range(nm):
            grads.Omega_sqt_F[i] = _p(views[8 + i])
            grads.delta_F[i] = _p(views[8 + nm + i])
        kk = 8 + 2 * nm
        for i in range(nm):
            if plan.lmc[i]:
                grads.W[i] = _p(views[kk])
                if og.dF_obs[i] is None and og.dF_obs_test[i] is None:
                    views[kk].zero_()  # no gradient reached F_obs: the engine leaves dW untouched
                kk += 1
        scratch = o._ws(plan.scratch_bytes, flat)
        # one optimiser step as several slices that close once (train.Microbatches; gpsa_step_io.bwd_acc): a slice that
        # is not the last leaves its gradient pieces in the accumulator and hands autograd nothing
        acc = aux.get("bwd_acc")
        closes = True
        if acc is not None:
            buf, mode = acc
            if buf.numel() < plan.bwd_acc_bytes:
                raise _lib.GpsaHipError("step engine: the microbatch accumulator is smaller than this plan needs")
            keep.append(buf)
            ctx.io.bwd_acc, ctx.io.bwd_acc_mode = buf.data_ptr(), int(mode)
            closes = mode == 3
        # data-parallel overlap (parallel.GradAllReducer(overlap=True)): the engine finishes the data GP's span of the
        # flat buffer - everything from Omega_sqt_F on - first and records the reducer's event behind it; the reducer
        # starts that span's all-reduce on its own stream while the rest of this backward runs
        early = model.__dict__.get("_early_reducer") if model is not None else None
        if early is not None and closes:
            # ... but only when, for every parameter in that span, what autograd will store as ``.grad`` IS this node's
            # view of the flat buffer: a parameter that already holds a gradient (accumulation over several
            # backwards: AccumulateGrad adds the view in place on the main stream, under the all-reduce in flight) or
            # that collects a second share elsewhere (an LMC modality's W on the fused-loss path: ElboLossFn's dW is
            # summed with the engine's view into a fresh tensor) would leave the ranks with different gradients.
            # Then: the ordinary order, no early reduce - the reducer's __call__ takes its general path.
            if early._early is not None:
                raise RuntimeError(
                    "GradAllReducer(overlap=True): a second backward() before the reducer was called - the first one's "
                    "data-GP span is already being all-reduced on the side stream and autograd would add this one's "
                    "gradients into it.  Call the reducer after every backward, or use overlap=False when "
                    "accumulating gradients over several backwards.")
            span = list(tensors[8:])
            # (any LMC modality declines: whether loss_fn took the fused likelihood is not this node's to know)
            if any(plan.lmc) or any(t.grad is not None for t in span if t.requires_grad):
                early = None
                STATS["early_reduce_declined"] = STATS.get("early_reduce_declined", 0) + 1
        if early is not None and closes:
            ctx.io.f_event = early.event_handle()
        else:
            ctx.io.f_event = None
        call = TO.stash(dict(lib=lib, handle=plan.handle, prm=ctx.prm, io=ctx.io, og=og, grads=grads))
        try:
            torch.ops.gpsa.step_backward(list(tensors), keep, ctx.arena, flat, scratch, call)
        finally:
            TO.CALLS.pop(call, None)
        if early is not None and closes:
            early.start_early(flat, offs[8], used)
        out = [None]
        for i, t in enumerate(tensors):
            out.append(views[i].view(t.shape) if (ctx.needs_input_grad[1 + i] and closes) else None)
        LAST_FLAT[dev.index] = flat
        LAST_USED[dev.index] = used
        # the arena (gigabytes when the data GPs keep their products) goes back to the allocator NOW: the node sits
        # in a reference cycle (model -> outputs -> grad_fn -> ctx -> aux -> model) that only the cyclic collector
        # would break, steps later
        if not getattr(ctx, "shared_arena", False):  # (a shared arena stays with train.Microbatches)
            _give_arena(plan, ctx.arena, force=aux.get("mm_epoch") is not None)
        ctx.arena = None
        if fuse is not None:  # the arena is gone: so is the chance to materialise this forward's draws
            fuse["live"] = None
            fuse["gloss"] = None
        # (aux / io / prm stay: a second backward - retain_graph=True - fills a fresh arena from them, _refill)
        if pend_check is not None:
            # (round 4: this wait used to open the backward.  On a launch-bound problem - BASELINE config 1's size - the
            #  host then sat out the device's backlog before it queued a single backward launch, and the device idled
            #  while it did: 1.00 ms/step against 0.69 without the check.  Behind the launches the wait is free, and an
            #  exception still leaves from loss.backward() before AccumulateGrad has seen any of this node's gradients.)
            if model is not None:
                # only THIS forward's pending check is retired: a later forward whose backward has not run yet
                # keeps its own (it is raised by that backward, or by the next forward)
                if model.__dict__.get("_pending_flag") is pend_check:
                    model._pending_flag = None
                model._raise_on_flags(pend_check)
            else:  # the model is gone (the node outlived it): the check itself needs no model
                from .models.vgpsa import VariationalGPSA

                VariationalGPSA._raise_on_flags(pend_check)
        return tuple(out)


    @staticmethod
    def _refill(ctx, aux, plan, model, tensors):
        """A second backward through one forward (``retain_graph=True``; the reference's graph, holding every
        intermediate, allows it - vgpsa.py:212-540): the arena went back to the allocator with the first backward, so
        the forward's launches run once more - same parameters (torch's version check on the saved tensors has just
        passed), same inputs, same draws, hence the same numbers into the same output buffers - into a fresh one."""
        re = aux["rerun"]
        dev = tensors[0].device
        o = _ops_mod.get_ops()
        sib = [q for q in (model.__dict__.get("_step_plans", {}).values() if model is not None else ()) if q is not plan]
        need = plan.saved_bytes if re["keep"] else plan.saved_bytes_nokeep
        saved = _take_arena(plan, need, dev, must=True, siblings=sib)
        ctx.io.reuse_mm = 0
        scratch = o._ws(plan.scratch_bytes, saved)
        fuse = aux.get("fuse")
        stages, extra = 3, []
        if fuse is not None:
            # stage 1, then the data GPs that ran: the ones forward itself ran (``rest``), the ones loss_fn ran fused
            # (their observations are on the record) and the ones that were materialised (their io slots point at the
            # real draws); an untouched lazy one is skipped by the backward anyway
            mask = re["rest"] | sum(1 << (8 + i) for i, st in enumerate(fuse["state"]) if st in ("fused", "real"))
            stages = 1 | ((2 | mask) if mask else 0)
            extra = ([y for y in fuse["Y"] if y is not None] + [fuse["noise"]]
                     + [f for f in fuse["F_real"] if f is not None] + [t for t in fuse["FT"] if t is not None])
        # the numerics word of this forward was read on the first pass: the re-run writes a throwaway device word instead
        # of the forward's pinned host slot (the two slots alternate, and a LATER forward that owns this one
        # by now could have its pending non-zero flag overwritten by this re-run's stale 0 before its check reads it)
        junk = torch.empty(1, dtype=torch.int32, device=dev)
        flag0 = ctx.io.flag
        ctx.io.flag = junk.data_ptr() if flag0 else flag0
        call = TO.stash(dict(lib=plan.lib, handle=plan.handle, prm=ctx.prm, io=ctx.io))
        try:
            torch.ops.gpsa.step_forward(list(tensors), re["ins"] + extra + [junk], re["outs"], saved, scratch, call, stages)
        finally:
            TO.CALLS.pop(call, None)
            ctx.io.flag = flag0
        ctx.arena = saved
        ctx.shared_arena = False
        if fuse is not None:
            fuse["live"] = None


class ElboLossFn(torch.autograd.Function):
    """loss = -(sum_i LL_i) + kl_scale * sum(kl_w * kl)   (vgpsa.py:532-540) as one C call each way.
    inputs: noise_variance [n], kl [T] or None, F_0 .. F_{n_ll-1};  aux: Y tensors, noise indices, kl_scale"""

    @staticmethod
    def forward(ctx, aux, noise, kl, *Fs):
        if aux.get("fuse_mods") is not None:
            return ElboLossFn._forward_fused(ctx, aux, noise, kl, *Fs)
        lib = _lib.load()
        o = _ops_mod.get_ops()
        n = len(Fs)
        dev = Fs[0].device
        Fc = [f.detach() if (f.dtype == torch.float32 and f.is_contiguous()) else f.detach().float().contiguous()
              for f in Fs]
        Yc = [y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous() for y in aux["Y"]]
        nz = noise.detach()
        nz = nz if (nz.dtype == torch.float32 and nz.is_contiguous()) else nz.float().contiguous()
        arr = lambda vals: (C.c_void_p * n)(*vals)
        Fp, Yp = arr([f.data_ptr() for f in Fc]), arr([y.data_ptr() for y in Yc])
        Np = arr([nz.data_ptr() + 4 * j for j in aux["noise_idx"]])
        Sa = (C.c_int * n)(*[int(f.shape[0]) for f in Fc])
        Na = (C.c_longlong * n)(*[int(f.shape[1]) for f in Fc])
        Pa = (C.c_int * n)(*[int(f.shape[2]) for f in Fc])
        klc = None
        if kl is not None:
            klc = kl.detach()
            klc = klc if (klc.dtype == torch.float64 and klc.is_contiguous()) else klc.double().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ll = torch.empty(n, dtype=torch.float64, device=dev)
        ws = o._ws(8 * 4100 * n + 64, loss)
        stream = _raw_stream(dev.index)
        torch.ops.gpsa.elbo_loss_fwd(Fc, Yc, nz, [int(j) for j in aux["noise_idx"]], klc, float(aux["kl_scale"]), loss,
                                     ll, ws)
        ctx.aux, ctx.args = aux, (Fc, Yc, nz, Fp, Yp, Np, Sa, Na, Pa)
        ctx.n_kl = 0 if klc is None else klc.numel()
        ctx.noise_meta = (noise.shape, noise.dtype)
        return loss.reshape(())

    @staticmethod
    def _forward_fused(ctx, aux, noise, kl, *ins):
        """some terms arrive as partial sums of z^2 instead of draws: from the step (StepFn with aux["fuse"]: the
        likelihood rode in the data GP's pass), or - an LMC modality, aux["lmc"] = {term: index of its W among the
        trailing inputs} - formed here from (F_latent, W, Y) without F_obs (gpsa_lmc_loglik_fused_f32)"""
        o = _ops_mod.get_ops()
        fuse = aux.get("fuse")
        lmc = aux.get("lmc") or {}
        fused = [bool(z) for z in aux["fuse_mods"]]  # the terms that come as partial sums in THIS call
        n = len(aux["Y"])
        Fs, Ws = ins[:n], ins[n:]
        dev = Fs[0].device
        Fc = [f.detach() if (z or (f.dtype == torch.float32 and f.is_contiguous())) else f.detach().float().contiguous()
              for f, z in zip(Fs, fused)]
        Yc = [y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous() for y in aux["Y"]]
        nz = noise.detach()
        nz = nz if (nz.dtype == torch.float32 and nz.is_contiguous()) else nz.float().contiguous()
        shapes = []
        for i in range(n):
            shapes += list(aux["term_shapes"][i]) if fused[i] else [int(d) for d in Fc[i].shape]
        lmc_saved = {}
        for i, wpos in lmc.items():
            Fl, W = Fc[i], Ws[wpos].detach()
            S_, N_, L_ = (int(d) for d in Fl.shape)
            nparts = int(_lib.load().gpsa_quadform_elbo_parts())
            zpart = torch.empty(nparts, dtype=torch.float64, device=dev)
            dFl, dW = torch.empty_like(Fl), torch.empty_like(W)
            wsl = o._ws(int(_lib.load().gpsa_lmc_loglik_workspace(S_ * N_, L_, int(W.shape[1]), nparts)), Fl)
            torch.ops.gpsa.lmc_loglik_fused(Fl, W, Yc[i], nz, int(aux["noise_idx"][i]), zpart, dFl, dW, wsl)
            lmc_saved[i] = (dFl, dW, wpos)
            Fc[i] = zpart
        klc = None
        if kl is not None:
            klc = kl.detach()
            klc = klc if (klc.dtype == torch.float64 and klc.is_contiguous()) else klc.double().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ll = torch.empty(n, dtype=torch.float64, device=dev)
        ws = o._ws(8 * 4100 * n + 64, loss)
        idx = [int(j) for j in aux["noise_idx"]]
        torch.ops.gpsa.elbo_loss_fused_fwd(Fc, Yc, nz, idx, shapes, [int(z) for z in fused], klc, float(aux["kl_scale"]),
                                           loss, ll, ws)
        ctx.aux, ctx.args = aux, (Fc, Yc, nz, idx, shapes, fused)
        ctx.lmc, ctx.n_w = lmc_saved, len(Ws)
        ctx.n_kl = 0 if klc is None else klc.numel()
        ctx.noise_meta = (noise.shape, noise.dtype)
        return loss.reshape(())

    @staticmethod
    def _backward_fused(ctx, gloss):
        o = _ops_mod.get_ops()
        aux = ctx.aux
        fuse = aux.get("fuse")
        Fc, Yc, nz, idx, shapes, fused = ctx.args
        dev = Fc[0].device
        g = gloss.detach().reshape(1)
        g = g if g.dtype == torch.float32 else g.float()
        if fuse is not None and any(z and i not in ctx.lmc for i, z in enumerate(fused)):
            # StepFn.backward hands it to the engine (gpsa_step_out_grads.gloss); loss_fn called twice on one forward: summed
            fuse["gloss"] = g if fuse.get("gloss") is None else fuse["gloss"] + g
        dF = [placeholder_grad(dev, f.numel()) if z else torch.empty_like(f) for f, z in zip(Fc, fused)]
        dWs = [None] * ctx.n_w
        for i, (dFl, dW, wpos) in ctx.lmc.items():  # formed at upstream gradient 1 by the forward: scaled here
            dF[i] = dFl * g
            dWs[wpos] = dW * g
        dnoise = torch.empty(nz.numel(), dtype=torch.float32, device=dev)
        dkl = torch.empty(ctx.n_kl, dtype=torch.float64, device=dev) if ctx.n_kl else None
        ws = o._ws(8 * 4100 * len(Fc) + 64, g)
        real = [d if not z else g for d, z in zip(dF, fused)]  # (mutable-argument list: no expanded tensors in it)
        torch.ops.gpsa.elbo_loss_fused_bwd(Fc, Yc, nz, idx, shapes, [int(z) for z in fused], g, int(ctx.n_kl),
                                           float(aux["kl_scale"]), real, dnoise, dkl, ws)
        shape, dt = ctx.noise_meta
        return (None, dnoise.reshape(shape).to(dt), dkl) + tuple(dF) + tuple(dWs)

    @staticmethod
    def backward(ctx, gloss):
        if ctx.aux.get("fuse_mods") is not None:
            return ElboLossFn._backward_fused(ctx, gloss)
        lib = _lib.load()
        o = _ops_mod.get_ops()
        aux = ctx.aux
        Fc, Yc, nz, Fp, Yp, Np, Sa, Na, Pa = ctx.args
        n = len(Fc)
        dev = Fc[0].device
        g = gloss.detach().reshape(1)
        g = g if g.dtype == torch.float32 else g.float()
        dF = [torch.empty_like(f) for f in Fc]
        dnoise = torch.empty(nz.numel(), dtype=torch.float32, device=dev)  # zero-filled by the first finishing launch
        dkl = torch.empty(ctx.n_kl, dtype=torch.float64, device=dev) if ctx.n_kl else None
        dFp = (C.c_void_p * n)(*[t.data_ptr() for t in dF])
        dNp = (C.c_void_p * n)(*[dnoise.data_ptr() + 4 * j for j in aux["noise_idx"]])
        ws = o._ws(8 * 4100 * n + 64, g)
        torch.ops.gpsa.elbo_loss_bwd(Fc, Yc, nz, [int(j) for j in aux["noise_idx"]], g, int(ctx.n_kl),
                                     float(aux["kl_scale"]), dF, dnoise, dkl, ws)
        shape, dt = ctx.noise_meta
        return (None, dnoise.reshape(shape).to(dt), dkl) + tuple(dF)
