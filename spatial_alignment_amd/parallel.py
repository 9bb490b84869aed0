"""Data-parallel sharding of the GPSA step over the GPUs of one node.

Everything N-scaled is independent per spot given the (replicated) M x M factors, so each rank owns
a contiguous slice of EVERY view's rows: it runs the warp + data layers and the likelihood for its
rows, adds its share of the KL terms (``own_kl_terms``: a contiguous range of them at weight 1 - owner
computes, round 6 - or all of them at 1/world: ``model.kl_scale``), and one all-reduce (RCCL over xGMI; backend "nccl" on ROCm) of
the flattened gradient makes every rank's gradient the full-ELBO gradient.  No other collective.

For many outputs (L = P in the thousands: BASELINE.json configs 4/5) the L axis shards instead
(``shard_outputs`` / ``setup_output_sharding``): a rank owns a slice of the output columns with their
``Omega_sqt_F`` rows and ``delta_F`` columns - the 2 GB of gradient those hold never leave the GPU -,
every rank runs the (cheap, identical) warp GPs, and only the gradients of the shared parameters
(a few MB) are all-reduced.
"""
import torch
import torch.distributed as dist


def shard_rows(n, rank, world):
    """[lo, hi) of rank's contiguous slice of n rows (sizes differ by at most 1)"""
    base, rem = divmod(int(n), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_data_dict(data_dict, rank, world):
    """per-rank data_dict: the same views, each cut to this rank's slice of its rows"""
    out = {}
    for mod, d in data_dict.items():
        ns = [int(x) for x in d["n_samples_list"]]
        rows, new_ns, off = [], [], 0
        for n in ns:
            lo, hi = shard_rows(n, rank, world)
            rows.append(torch.arange(off + lo, off + hi))
            new_ns.append(hi - lo)
            off += n
        idx = torch.cat(rows).to(d["spatial_coords"].device)
        out[mod] = {
            "spatial_coords": d["spatial_coords"][idx].contiguous(),
            "outputs": d["outputs"][idx].contiguous(),
            "n_samples_list": new_ns,
        }
    return out


def shard_outputs(data_dict, rank, world):
    """per-rank data_dict: every row, this rank's contiguous slice of each modality's output columns"""
    out = {}
    for mod, d in data_dict.items():
        lo, hi = shard_rows(d["outputs"].shape[1], rank, world)
        out[mod] = {
            "spatial_coords": d["spatial_coords"],
            "outputs": d["outputs"][:, lo:hi].contiguous(),
            "n_samples_list": list(d["n_samples_list"]),
        }
    return out


def own_kl_terms(model, rank, world):
    """Row-sharded rank ``rank`` of ``world``: OWNER COMPUTES for everything only the KL needs (round 6).

    Instead of every rank evaluating all V*D + sum L KL terms at weight 1/world (``model.kl_scale = 1 / world``), each
    rank evaluates a contiguous share of them at weight 1: it factorises and inverts only the priors and ITS OWN
    variational covariances (3 + ceil(54 / world) matrices instead of 57 at BASELINE config 2), runs the KL forward and
    backward for those, and the all-reduce of the gradient - which the step needs anyway - sums the shares.  The sum of
    the ranks' losses is the full negative ELBO, the sum of their gradients its gradient (vgpsa.py:498-530 are sums of
    independent terms).  Returns the [lo, hi) this rank owns."""
    from . import step_engine as SE

    model.kl_scale = 1.0
    model.kl_owner = (int(rank), int(world)) if int(world) > 1 else None
    own = SE.kl_own_range(model)
    n = model.n_views * model.n_spatial_dims + sum(int(model.n_latent_outputs[m]) for m in model.modality_names)
    return own if own is not None else (0, n)


OUTPUT_LOCAL_PREFIXES = ("Omega_sqt_F_dict.", "delta_F_dict.")


def shared_parameters(model):
    """the parameters every output-sharded rank holds a copy of (everything but the per-output
    variational parameters of the data GP)"""
    return [p for n, p in model.named_parameters() if not n.startswith(OUTPUT_LOCAL_PREFIXES)]


def setup_output_sharding(model, rank, world, seed=0):
    """Turn a model built on ``shard_outputs(data_dict, rank, world)`` into rank ``rank`` of an
    output-sharded job (no latent mixing: n_latent_gps None, so outputs are independent given G):

    * loss = -LL(own outputs) + KL(own outputs) + KL(warp GPs) / world: summed over ranks this is the
      full negative ELBO, and so is the sum of the shared parameters' gradients;
    * the warp draws come from a generator seeded alike on every rank (all ranks must see the same
      aligned coordinates), the output draws from one seeded per rank;
    * shared parameters are broadcast from rank 0 (construction may have consumed the RNG differently).

    Returns the GradAllReducer over the shared parameters: call it between backward() and step()."""
    for m in model.modality_names:
        if model.n_latent_gps[m] is not None:
            raise ValueError("output sharding needs independent outputs (n_latent_gps[mod] = None)")
    model.kl_scale = 1.0
    model.kl_weight_G = 1.0 / world
    dev = model.Xtilde.device
    g_common, g_local = torch.Generator(device=dev), torch.Generator(device=dev)
    g_common.manual_seed(int(seed))
    g_local.manual_seed(int(seed) * 1000003 + 7919 * (rank + 1))
    model.noise_generators = {"G": g_common, "F": g_local}
    shared = shared_parameters(model)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        with torch.no_grad():
            for p in shared:
                dist.broadcast(p, src=0)
            for name, b in model.named_buffers():
                dist.broadcast(b, src=0)
    return GradAllReducer(shared)


class GradAllReducer:
    """One fused all-reduce(sum) of all parameter gradients per step (bucket = everything: 8.6 MB at
    the headline config, latency-bound on xGMI, so a single flat buffer is the right shape).

    The step engine hands autograd ONE flat gradient buffer whose views are the parameters' ``.grad``
    (step_engine.StepFn.backward), with spare room at its end: when that is what the parameters hold, the
    few gradients from elsewhere (the likelihood's noise parameter) are copied into the spare room and the
    buffer itself is reduced in place - no pack, no unpack.  Otherwise: pack, reduce, unpack.
    Every call inside it (copies, the RCCL all-reduce) is capturable into a hipGraph."""

    def __init__(self, params, always=False, overlap=False, model=None):
        """``overlap`` (with ``model``, a VariationalGPSA on the step engine): the data GP's span of the flat gradient
        buffer (Omega_sqt_F, delta_F, W: 97 % of the bytes at the headline configuration) is reduced on a side stream
        while the rest of the backward - the warp GPs' backward, the priors' covariance backward - still runs; the
        engine finishes that span first and records this reducer's event behind it (gpsa_step_io.f_event).  ``__call__``
        then reduces what is left and joins the side stream."""
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.always = always  # reduce even in a 1-rank group (tests of the capture path)
        self.overlap = bool(overlap) and model is not None and self.params and self.params[0].is_cuda
        self._early = None    # (flat, lo, hi) of the span whose all-reduce is in flight on the side stream
        if self.overlap:
            self._side = torch.cuda.Stream(device=self.params[0].device)
            self._event = torch.cuda.Event()
            self._event.record()  # (creates the underlying hipEvent: its handle goes to the engine)
            model.__dict__["_early_reducer"] = self

    def event_handle(self):
        return int(self._event.cuda_event)

    def start_early(self, flat, lo, hi):
        """called by StepFn.backward right after the engine's launches are queued: reduce flat[lo:hi] behind the event"""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not self.always):
            return
        if hi <= lo:
            return
        self._side.wait_event(self._event)
        flat.record_stream(self._side)
        with torch.cuda.stream(self._side):
            dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM)
        self._early = (flat, lo, hi)

    def _engine_bucket(self):
        """(flat buffer, parameters outside it) when the gradients are views of one step-engine buffer"""
        from .step_engine import LAST_FLAT, LAST_USED

        p0 = self.params[0]
        flat = LAST_FLAT.get(p0.device.index) if p0.is_cuda else None
        if flat is None:
            return None
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        inside, outside = [], []
        for p in self.params:
            g = p.grad
            if g is not None and g.is_contiguous() and g.dtype == flat.dtype and lo <= g.data_ptr() < hi:
                inside.append(p)
            else:
                outside.append(p)
        # the views are padded to 256-byte boundaries: the bucket is the whole span (padding included: its content
        # is reduced too and never read)
        used = LAST_USED.get(p0.device.index, sum(p.numel() for p in inside))
        if sum(p.numel() for p in outside) > flat.numel() - used:
            return None
        # The buffer may also hold gradients that are NOT this reducer's (output sharding: the per-output parameters'
        # gradients, gigabytes at BASELINE config 4, must stay on their rank): reduce only the spans our parameters
        # occupy.  Views follow each other at 64-float granules, so neighbours merge into one span.
        es = flat.element_size()
        spans = sorted(((p.grad.data_ptr() - lo) // es, p.grad.numel()) for p in inside)
        merged = []
        for off, n in spans:
            end = (off + n + 63) // 64 * 64
            if merged and off <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], end)
            else:
                merged.append([off, end])
        if not merged:
            return None
        if len(merged) == 1 and merged[0][0] == 0 and merged[0][1] >= used:
            return flat, used, outside, None          # everything in the buffer is ours
        return flat, used, outside, [(a, min(b, used)) for a, b in merged]

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() == 1 and not self.always:
            return
        for p in self.params:  # a parameter the step did not touch contributes zeros
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        bucket = self._engine_bucket()
        if self._early is not None:
            eflat, lo, hi = self._early
            self._early = None
            main = torch.cuda.current_stream(eflat.device)
            if bucket is not None and bucket[0] is eflat and bucket[3] is None:
                # the rest of the buffer - the small parameters and Omega_sqt_G in front of the early span, the
                # outsiders behind it - on this stream, then the join
                flat, used, outside, _ = bucket
                if lo > 0:
                    dist.all_reduce(flat[:lo], op=dist.ReduceOp.SUM)
                if outside:
                    tail = flat[used: used + sum(p.numel() for p in outside)]
                    torch.cat([p.grad.view(-1) for p in outside], out=tail)
                    dist.all_reduce(tail, op=dist.ReduceOp.SUM)
                    torch._foreach_copy_([p.grad.view(-1) for p in outside],
                                         list(tail.split([p.numel() for p in outside])))
                main.wait_stream(self._side)
                return
            # Not the layout the shortcut knows (StepFn.backward declines the early reduce when it can tell beforehand;
            # this is the net under it): the span eflat[lo:hi] IS reduced, so finish WITHOUT touching it again - every
            # parameter whose .grad is not a view inside that span goes through a packed buffer - instead of raising
            # with the ranks' gradients half summed.
            main.wait_stream(self._side)
            es = eflat.element_size()
            a, b = eflat.data_ptr() + lo * es, eflat.data_ptr() + hi * es
            rest = [p for p in self.params
                    if not (p.grad.is_contiguous() and p.grad.dtype == eflat.dtype and a <= p.grad.data_ptr() < b)]
            if rest:
                tmp = torch.cat([p.grad.reshape(-1).to(eflat.dtype) for p in rest])
                dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
                for p, t in zip(rest, tmp.split([p.numel() for p in rest])):
                    p.grad.copy_(t.view_as(p.grad))
            return
        if bucket is not None and bucket[3] is not None:
            flat, used, outside, spans = bucket
            for a, b in spans:
                dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM)
            if outside:
                tail = flat[used: used + sum(p.numel() for p in outside)]
                torch.cat([p.grad.view(-1) for p in outside], out=tail)
                dist.all_reduce(tail, op=dist.ReduceOp.SUM)
                torch._foreach_copy_([p.grad.view(-1) for p in outside],
                                     list(tail.split([p.numel() for p in outside])))
            return
        if bucket is not None:
            flat, used, outside, _ = bucket
            if outside:
                tail = flat[used: used + sum(p.numel() for p in outside)]
                torch.cat([p.grad.view(-1) for p in outside], out=tail)
                dist.all_reduce(flat[: used + tail.numel()], op=dist.ReduceOp.SUM)
                torch._foreach_copy_([p.grad.view(-1) for p in outside],
                                     list(tail.split([p.numel() for p in outside])))
            else:
                dist.all_reduce(flat[:used], op=dist.ReduceOp.SUM)
            return
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.empty(self.numel, dtype=p0.dtype, device=p0.device)
            self.views = list(self.flat.split([p.numel() for p in self.params]))
        # pack (one batched launch), reduce, unpack (one batched launch): not one copy per parameter
        # view(-1), not reshape(-1): a non-contiguous .grad must raise here, not be reduced into a temporary
        torch.cat([p.grad.view(-1) for p in self.params], out=self.flat)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        torch._foreach_copy_([p.grad.view(-1) for p in self.params], self.views)
