"""``VariationalGPSA``: the variational two-layer (deep) GP of GPSA on MI355X.

Drop-in for the reference class gpsa/models/vgpsa.py:14-540 — same constructor keywords (including
the ones the reference accepts and ignores), parameter names, ``forward`` / ``loss_fn`` signatures and
return structure — with the numerical work done by hand-written HIP kernels (see ../engine.py and
../csrc).  The index and scale quirks of the reference are reproduced on purpose; they are listed in
SURVEY.md §8a and flagged ``# quirk N`` below.
"""
import contextlib
import weakref
from collections.abc import Iterable

import numpy as np
import os

import torch
import torch.nn as nn

from .. import engine as E
from .. import step_engine as SE
from ..lazy import LazyDraws, LazyProduct
from ..kernels import builtin_kind, rbf_kernel
from .gpsa import GPSA


def _as_index(idx, device):
    """view row indices -> (slice | LongTensor, count); contiguous ranges become slices"""
    a = np.asarray(idx.cpu() if torch.is_tensor(idx) else idx)
    n = int(a.shape[0])
    if n == 0:
        return slice(0, 0), 0
    if int(a[-1]) - int(a[0]) == n - 1 and (n == 1 or bool(np.all(np.diff(a) == 1))):
        return slice(int(a[0]), int(a[0]) + n), n
    return torch.as_tensor(a, dtype=torch.long, device=device), n


class _StepCache:
    """forward -> loss_fn hand-off (the reference keeps the same state on ``self``, vgpsa.py:217-412)."""

    def __init__(self):
        self.warp = {}  # view -> (Kuu, Factor)
        self.data = None  # (Kuu_F, Factor)
        self.Omega_G = None  # [V*D, M, M] fp64
        self.Omega_F = {}  # mod -> [L, M, M] fp64
        self.Omega_G_fac = None  # (Omega_G^-1 [V*D,M,M], logdet [V*D])
        self.Omega_F_fac = {}  # mod -> (Omega_F^-1, logdet)
        self.flags = []  # device int tensors: Cholesky info / non-positive variance flags
        self.mu_z = self.dG_v = self.resid = None  # per-view prior means / variational means / their difference
        self.kl = None  # step engine: the per-term KL vector (a differentiable output of the forward node)
        self.batch = self.free = None  # (matrices, inverses, logdets) of the whole factorisation batch
        self.Om_fwd = self.Om_kl = None  # per-view row groups of Omega_G
        self.fuse = None  # step engine, fused ELBO: the record forward and loss_fn share (step_engine.StepFn)


class VariationalGPSA(GPSA):
    def __init__(
        self,
        data_dict,
        m_X_per_view,
        m_G,
        data_init=True,
        minmax_init=False,
        grid_init=False,
        n_spatial_dims=2,
        n_noise_variance_params=2,
        kernel_func_warp=rbf_kernel,
        kernel_func_data=rbf_kernel,
        n_latent_gps=None,
        mean_function="identity_fixed",
        mean_penalty_param=0.0,
        fixed_warp_kernel_variances=None,
        fixed_warp_kernel_lengthscales=None,
        fixed_data_kernel_lengthscales=None,
        fixed_view_idx=None,
    ):
        # quirk 6: n_spatial_dims / n_noise_variance_params / mean_function / minmax_init are accepted
        # and ignored by the reference (vgpsa.py:35-46): the mean function is always identity, fixed.
        super().__init__(
            data_dict,
            data_init=True,
            n_spatial_dims=2,
            n_noise_variance_params=2,
            kernel_func_warp=kernel_func_warp,
            kernel_func_data=kernel_func_data,
            mean_penalty_param=mean_penalty_param,
            fixed_warp_kernel_variances=fixed_warp_kernel_variances,
            fixed_warp_kernel_lengthscales=fixed_warp_kernel_lengthscales,
            fixed_data_kernel_lengthscales=fixed_data_kernel_lengthscales,
        )
        self.m_X_per_view = m_X_per_view
        self.m_G = m_G
        if n_latent_gps is None:  # the reference requires a dict; None means "no LMC anywhere"
            n_latent_gps = {m: None for m in self.modality_names}
        self.n_latent_gps = n_latent_gps
        self.n_latent_outputs = {
            m: (n_latent_gps[m] if n_latent_gps[m] is not None else self.Ps[m])
            for m in self.modality_names
        }
        self.fixed_view_idx = fixed_view_idx
        # True: forward raises (torch.linalg.LinAlgError) on a non-positive-definite covariance or a non-positive
        # warp variance, as the reference does - one host wait per forward.  With gradients enabled and the step
        # engine the wait is DEFERRED to the start of this forward's backward (still before any gradient or
        # parameter is touched; a forward whose backward never runs is checked at the next forward): the host
        # keeps queueing the ELBO while the warp GPs run.  "strict": always wait inside forward.  False: never.
        self.check_numerics = True
        self.kl_scale = 1.0  # data-parallel ranks add 1/world of the KL each (parallel.py)
        # ... or, OWNER COMPUTES (round 6): ``(rank, world)`` - this rank evaluates a contiguous share of the
        # V*D + sum L KL terms with weight 1 (kl_scale stays 1), the others come out as 0 with zero gradients, and the
        # gradient all-reduce sums the shares: a rank factorises and inverts only ITS variational covariances
        # (3 + ceil(54 / world) matrices instead of 57 at BASELINE config 2).  parallel.own_kl_terms sets it.
        self.kl_owner = None
        # output-sharded ranks (parallel.shard_outputs) own their outputs' KL terms in full and share the
        # warp GPs': weight of the warp-GP KL terms inside the KL sum, and separate generators for the
        # draws every rank must agree on (warp) and the ones it must not share (its own outputs)
        self.kl_weight_G = 1.0
        self.noise_generators = None  # {"G": torch.Generator, "F": torch.Generator} or None
        # warp GPs of different views on side HIP streams: pays under hipGraph replay (-0.2 ms at the
        # headline config), costs CPU time per launch in eager mode, so train.GraphedTrainStep turns it on
        self.overlap_views = False
        # forward as ONE node whose launch sequence is enqueued from C++ (step_engine.py, csrc/step.hip);
        # False: one node per layer driven from Python (the path arbitrary plug-in covariance callables take)
        self.use_step_engine = True
        # training forwards keep the data GPs' products Omega_l alpha for the backward (L M C 4 bytes of HBM,
        # one product less per step); False: recompute them in the backward (the memory-lean path, also taken
        # by itself when the products exceed the keep budget below)
        self.keep_products = True
        # A training forward leaves the data GP of every modality the fused kernel covers (Gaussian likelihood on
        # F_latent, no LMC, M <= 256) to loss_fn, which is where the observations arrive (vgpsa.py:532-538): variance,
        # draw, likelihood, its gradient and the backward's alpha-gradient then ride in ONE pass over the products
        # Omega_l alpha (gpsa_quadform_elbo_f32) - nothing is kept, nothing streamed back.  F_samples come back as lazy
        # handles (lazy.LazyDraws) that loss_fn understands and that turn into real draws when anything else touches
        # them.  False (or GPSA_FUSE_ELBO=0): always the separate kernels
        self.fuse_elbo = os.environ.get("GPSA_FUSE_ELBO", "1") != "0"
        # ... from the size where that pays: below ~5 GF of data-GP contraction per step (2 S N L M^2) the step is
        # bound by its launches and the host, and the deferred second half costs 7-10 % (0.63 vs 0.58 ms at BASELINE
        # config 1's size, 0.83 vs 0.77 at 2 x 900 spots; 0.85 vs 0.88 at 2 x 1600, 3.77 vs 4.11 at 2 x 4900:
        # tools/fuse_threshold_probe.py).  GPSA_FUSE_MIN_FLOPS / this attribute move the line; 0 = always
        self.fuse_min_flops = float(os.environ.get("GPSA_FUSE_MIN_FLOPS", "5e9"))
        self.keep_budget_gb = None  # HBM for those products: None = what the device can still give (step_engine.py)
        # the data GP's inducing-point gradient from the UNROUNDED projection (gpsa_step_desc.exact_inducing_grad):
        # None / True = on (round 5: the default everywhere, so that the step that is timed is the step every gradient
        # of which the parity suite holds to 1e-4 of the reference's fp64 run); False (or GPSA_EXACT_GRAD=0) = off:
        # grad Gtilde then carries ~1e-3 relative error at M >= 200 on ill-conditioned K_uu (every other gradient and
        # every output are unaffected: its K_uu and K_uf shares cancel to 1e-4 .. 1e-5 of their size), the step is one
        # M x M x C fp64 product and some fp64 panel traffic cheaper (bench.py reports both: "exact_inducing_grad").
        self.exact_inducing_grad = None
        self._noise = None  # injected Gaussian noise (tests / reproducibility), see inject_noise()
        self._cache = None

        V, D = self.n_views, self.n_spatial_dims
        mods = self.modality_names
        if data_init:
            # inducing locations = k-means centres of the data (vgpsa.py:61-92).  Coordinates that
            # already live in HBM are clustered there (init.kmeans, HIP Lloyd iterations); CPU tensors
            # take the reference's scikit-learn path.
            first = data_dict[mods[0]]["spatial_coords"]
            on_device = bool(getattr(first, "is_cuda", False))
            if on_device:
                from ..init import kmeans as _kmeans

                def centres(X, k):
                    return _kmeans(X, k, seed=int(np.random.randint(0, 2**31 - 1))).cpu()
            else:
                from sklearn.cluster import KMeans

                def centres(X, k):
                    return torch.tensor(KMeans(n_clusters=k).fit(X.detach().cpu().numpy()).cluster_centers_)

            Xt = torch.zeros([V, self.m_X_per_view, D])
            for v in range(V):
                Xv = torch.cat([data_dict[m]["spatial_coords"][self.view_idx[m][v], :] for m in mods], 0)
                Xt[v] = centres(Xv, self.m_X_per_view)
            self.Xtilde = nn.Parameter(Xt.clone())
            # the reference draws (and discards) a subset here: raises if m_G > spots of the last view
            np.random.choice(np.arange(Xv.shape[0]), size=self.m_G, replace=False)
            allX = torch.cat([data_dict[m]["spatial_coords"] for m in mods])
            self.Gtilde = nn.Parameter(centres(allX, self.m_G).to(self.Xtilde.dtype))
        elif grid_init:
            if D == 2:  # lattice over the bounding box of the first modality (vgpsa.py:94-121)
                xy = data_dict[mods[0]]["spatial_coords"].detach().cpu().numpy()
                lo, hi = xy.min(0), xy.max(0)
                ticks = int(np.ceil(np.sqrt(self.m_G)))
                self.m_G = self.m_X_per_view = ticks**2
                g1, g2 = np.meshgrid(np.linspace(lo[0], hi[0], ticks), np.linspace(lo[1], hi[1], ticks))
                lattice = torch.tensor(np.vstack([g1.ravel(), g2.ravel()]).T).float()
                self.Xtilde = nn.Parameter(lattice.unsqueeze(0).repeat(V, 1, 1).clone())
                self.Gtilde = nn.Parameter(lattice.clone())
        else:
            self.Xtilde = nn.Parameter(torch.randn([V, self.m_X_per_view, D]))
            self.Gtilde = nn.Parameter(torch.randn([self.m_G, D]))

        M_X, M_G = self.m_X_per_view, self.m_G
        # variational covariance square roots; row of (view v, dim j) is j*V + v (vgpsa.py:131-143)
        A = torch.zeros([V * D, M_X, M_X])
        for v in range(V):
            for j in range(D):
                A[j * V + v] = 0.1 * torch.randn(size=[M_X, M_X])
        self.Omega_sqt_G_list = nn.Parameter(A)
        self.Omega_sqt_F_dict = nn.ParameterDict()
        for m in mods:
            L = self.n_latent_outputs[m]
            A = torch.zeros([L, M_G, M_G])
            for l in range(L):
                A[l] = 0.1 * torch.randn(size=[M_G, M_G])
            self.Omega_sqt_F_dict[m] = nn.Parameter(A)
        # variational means; delta_G starts at Xtilde => G_means == X at initialisation (quirk 9)
        self.delta_G_list = nn.Parameter(self.Xtilde.detach().clone())
        self.delta_F_dict = nn.ParameterDict()
        for m in mods:
            self.delta_F_dict[m] = nn.Parameter(torch.randn(size=[M_G, self.n_latent_outputs[m]]))
        self.W_dict = nn.ParameterDict()
        for m in mods:
            if self.n_latent_gps[m] is not None:
                self.W_dict[m] = nn.Parameter(torch.randn([self.n_latent_gps[m], self.Ps[m]]))

    # ------------------------------------------------------------------------------------------
    def _is_fixed(self, v):
        f = self.fixed_view_idx
        if f is None:
            return False
        return (v in f) if isinstance(f, Iterable) else (f == v)

    def release_arenas(self):
        """Hand back the arenas the step engine keeps parked between training steps (a GiB and more each: the kept
        products of a large configuration).  ``eval()`` and ``prediction_mode`` do it by themselves."""
        SE.release_arenas(self)

    def train(self, mode=True):
        if not mode:
            SE.release_arenas(self)
        return super().train(mode)

    def inject_noise(self, eps_G=None, eps_F=None, eps_F_test=None):
        """Use the given standard-normal draws in the NEXT forward instead of drawing them.

        eps_G: list over the non-fixed, non-empty views in order, each [S, n_v, D];
        eps_F / eps_F_test: {mod: [S, N, L]} — the draw order of vgpsa.py:346-348, 423, 465.
        """
        self._noise = dict(G=eps_G, F=eps_F, F_test=eps_F_test)

    def compute_mean_and_var(self, Kff_diag, Kuf, Kuu_chol, mu_x, mu_z, delta, Omega_tril):
        """The sparse-GP conditional of vgpsa.py:174-204 as a stand-alone call with the reference's arguments
        and return shapes (2-D ``Kuf`` [M,n]: mean [V,n,D], var [V*D,n]; 3-D ``Kuf`` [S,M,N]: mean [S,N,L],
        var [S,L,N]; the jitter is added twice, quirk 3).  ``forward`` does not go through here - it runs the
        fused layer nodes - but the numbers are the same HIP kernels (engine.SGPCoreFn): fp64 projection,
        fp64 forms for the 2-D (warp) branch, fp32 matrix-core forms for the 3-D (data) branch.
        Differentiable wrt Kuf, delta - mu_z and the two factors."""
        f64 = torch.float64
        Lk, Lo = Kuu_chol.to(f64), Omega_tril.to(f64)
        Kuu = Lk @ Lk.transpose(-1, -2)
        Om = Lo @ Lo.transpose(-1, -2)  # Omega_tril Omega_tril^T == Omega: the kernels take Omega itself
        fac = E.Factor(Kuu)
        jit2 = 2.0 * self.diagonal_offset
        dcm = delta - mu_z
        if Kuf.dim() == 2:
            M, n = Kuf.shape
            dc3 = dcm.reshape(-1, M, dcm.shape[-1])                      # [V, M, D]
            V, D = dc3.shape[0], dc3.shape[2]
            dc2 = dc3.permute(1, 0, 2).reshape(M, V * D)                 # column v*D + j
            meanT, v, q = E.SGPCoreFn.apply(Kuu, Kuf.to(f64), dc2.to(f64), Om, fac, f64)
            mean = mu_x.unsqueeze(0) + meanT.reshape(V, D, n).transpose(1, 2)
            var = Kff_diag - q.unsqueeze(0) + v + jit2
        else:
            S, M, N = Kuf.shape
            K2 = Kuf.permute(1, 0, 2).reshape(M, S * N)
            meanT, v, q = E.SGPCoreFn.apply(Kuu, K2.to(f64), dcm, Om, fac, torch.float32)
            L = v.shape[0]
            mean = mu_x.unsqueeze(0) + meanT.reshape(L, S, N).permute(1, 2, 0)
            kff = Kff_diag.unsqueeze(1) if torch.is_tensor(Kff_diag) and Kff_diag.dim() >= 1 else Kff_diag
            var = kff - q.reshape(S, 1, N) + v.reshape(L, S, N).permute(1, 0, 2) + jit2
        return mean.to(Kuf.dtype), var.to(Kuf.dtype)

    def get_Omega_from_Omega_sqt(self, Omega_sqt):
        """Omega = A A^T + 1e-5 I (vgpsa.py:206-210); fp64 result."""
        return E.OmegaFn.apply(Omega_sqt)

    def _kmat(self, which, Z, X, ls_u, var_u, jitter, dtype, same, bwd_dtype=None, out=None):
        fn = self.kernel_func_warp if which == "warp" else self.kernel_func_data
        kind = builtin_kind(fn)
        if kind is not None:
            return E.KmatFn.apply(kind, Z, X, ls_u, var_u, jitter, dtype, same, bwd_dtype, out)
        # arbitrary plugin callable: evaluate it as the reference does (vgpsa.py:275-281, 382-388)
        K = fn(
            Z.to(dtype),
            X.to(dtype),
            lengthscale_unconstrained=ls_u.to(dtype),
            output_variance_unconstrained=var_u.to(dtype),
            diag=False,
        )
        if jitter:
            K = K + jitter * torch.eye(K.shape[-1], dtype=dtype, device=K.device)
        return K

    # attributes the reference's forward leaves behind (vgpsa.py:217, 283-289); nothing on the hot path
    # reads them, so they are formed on access instead of costing launches every step
    @property
    def noise_variance_pos(self):
        return torch.exp(self.noise_variance.detach()) + self.diagonal_offset

    @property
    def mu_z_G(self):
        cache = self.__dict__.get("_cache")
        if cache is None or cache.mu_z is None or (isinstance(cache.mu_z, list) and not cache.mu_z):
            raise AttributeError("mu_z_G is available after forward")
        return cache.mu_z if torch.is_tensor(cache.mu_z) else torch.stack(cache.mu_z)

    # ---- the reference's forward -> loss_fn hand-off attributes (vgpsa.py:237, 257, 321, 353-355, 394, 412) ------------
    # Kuu_chol_list [V, M_X, M_X] (NaN for fixed / empty views), curr_Omega_tril_list [V*D, M_X, M_X], Kuu_chol_F
    # [M_G, M_G], curr_Omega_tril_F {mod: [L, M_G, M_G]}: the lower Cholesky factors of K_uu + 1e-5 I and of
    # Omega = A A^T + 1e-5 I.  The step never forms them (its layers use explicit fp64 inverses, its KL the
    # log-determinants): the engine keeps the MATRICES in its arena's fp64 batch and these properties factorise them on
    # access (gpsa_chol_f64, once per forward, memoised), rounded to the parameters' dtype, detached.  While the
    # forward's arena is alive (until its backward has run; parked arenas of a GiB and more stay valid until the next
    # forward) they are the forward's own matrices; afterwards they are recomputed from the CURRENT parameters with the
    # same kernels - equal to the reference's until the optimiser has stepped.
    def _handoff_factors(self):
        cache = self.__dict__.get("_cache")
        if cache is None:
            raise AttributeError("Kuu_chol_list / curr_Omega_tril_list / Kuu_chol_F / curr_Omega_tril_F are available "
                                 "after forward")
        memo = cache.__dict__.get("_handoff")
        if memo is not None:
            return memo
        import ctypes as C_

        mods = self.modality_names
        V, D = self.n_views, self.n_spatial_dims
        f64 = torch.float64
        free = [v for v in range(V) if not self._is_fixed(v)]
        o = E.ops()
        arena_ref = cache.__dict__.get("arena_ref")
        arena = arena_ref() if arena_ref is not None else None
        plan = cache.__dict__.get("plan")
        live = set(free)
        if plan is not None:  # a view without rows in that forward is skipped by the reference (vgpsa.py:296-297)
            live = {v for v in free if sum(plan.rows[i * V + v] for i in range(len(mods))) > 0}
        with torch.no_grad():
            if arena is not None and plan is not None:
                lay = (C_.c_longlong * 9)()
                if plan.lib.gpsa_step_batch_layout(plan.handle, lay) != 0:
                    raise RuntimeError("gpsa_step_batch_layout refused the plan")
                groups = []
                for g in range(int(lay[0])):
                    M, npri, nom, off = (int(lay[1 + 4 * g + j]) for j in range(4))
                    nb = npri + nom
                    groups.append((arena[off: off + nb * M * M * 8].view(f64).view(nb, M, M), npri))
                merged = len(groups) == 1
                (m0, np0), (m1, _) = groups[0], groups[-1]
                Kw = {v: m0[b] for b, v in enumerate(free)}
                KF = m1[len(free) if merged else 0]
                OmG = m0[np0: np0 + V * D]
                base = np0 + V * D if merged else 1
                OmF = {}
                for m in mods:
                    L = int(self.n_latent_outputs[m])
                    OmF[m] = m1[base: base + L]
                    base += L
            elif cache.data is not None:  # per-layer path: the matrices are on the cache
                Kw = {v: kf[0] for v, kf in cache.warp.items()}
                KF = cache.data[0]
                OmG, OmF = cache.Omega_G, dict(cache.Omega_F)
            else:  # the forward's arena is gone: the same kernels on the current parameters
                wide = lambda t: t.double() if t.dtype == torch.float32 else t
                Kw = {v: self._kmat("warp", wide(self.Xtilde[v]), wide(self.Xtilde[v]),
                                    wide(self.warp_kernel_lengthscales[v]), wide(self.warp_kernel_variances[v]),
                                    self.diagonal_offset, f64, True) for v in free}
                KF = self._kmat("data", wide(self.Gtilde), wide(self.Gtilde), wide(self.data_kernel_lengthscale),
                                wide(self.data_kernel_variance), self.diagonal_offset, f64, True)
                OmG = E.OmegaFn.apply(self.Omega_sqt_G_list)
                OmF = {m: E.OmegaFn.apply(self.Omega_sqt_F_dict[m]) for m in mods}
            dt = self.Xtilde.dtype
            chol = lambda A: o.chol(A.detach().to(f64).reshape(-1, A.shape[-1], A.shape[-1]))[0].to(dt)
            Mx = int(self.Xtilde.shape[1])
            Kl = torch.full([V, Mx, Mx], float("nan"), dtype=dt, device=self.Xtilde.device)
            vs = [v for v in free if v in live and v in Kw]
            if vs:
                Kl[vs] = chol(torch.stack([Kw[v].detach().to(f64) for v in vs]))
            memo = dict(Kuu_chol_list=Kl, curr_Omega_tril_list=chol(OmG), Kuu_chol_F=chol(KF)[0],
                        curr_Omega_tril_F={m: chol(OmF[m]) for m in mods})
        cache.__dict__["_handoff"] = memo
        return memo

    @property
    def Kuu_chol_list(self):
        return self._handoff_factors()["Kuu_chol_list"]

    @property
    def curr_Omega_tril_list(self):
        return self._handoff_factors()["curr_Omega_tril_list"]

    @property
    def Kuu_chol_F(self):
        return self._handoff_factors()["Kuu_chol_F"]

    @property
    def curr_Omega_tril_F(self):
        return self._handoff_factors()["curr_Omega_tril_F"]

    def _side_streams(self, n, device):
        pool = self.__dict__.setdefault("_stream_pool", [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=device))
        return pool[:n]

    def _draw(self, shape, device, which="G"):
        gen = None if self.noise_generators is None else self.noise_generators.get(which)
        return torch.empty(shape, dtype=torch.float32, device=device).normal_(generator=gen)

    # ------------------------------------------------------------------------------------------
    def forward(self, X_spatial, view_idx, Ns, S=1, prediction_mode=False, G_test=None):
        if prediction_mode:
            self.eval()
        dev = self.Xtilde.device
        mods = self.modality_names
        V, D = self.n_views, self.n_spatial_dims
        f64 = torch.float64
        noise, self._noise = self._noise, None
        if self.use_step_engine and dev.type == "cuda" and SE.eligible(self, X_spatial, view_idx, G_test):
            rows = SE.view_rows(self, view_idx, Ns)
            if rows is not None:
                return self._forward_engine(X_spatial, rows, S, G_test, noise, prediction_mode)
        cache = _StepCache()

        # per-view slices of the parameters, unbound once (one autograd node per parameter instead of
        # a zero-fill + copy + add per slice in the backward)
        # The inducing locations and the covariance hyper-parameters enter through fp64 copies: each of them
        # collects gradients along several paths (K_uu, K_uf, the mean function; the prior's KL term) that
        # cancel to ~1e-4 of their size when the inducing points are dense - summed in fp64 and rounded to
        # the parameter's fp32 ONCE, by the cast's backward (rounding each path's share to fp32 first costs
        # 1e-3 on grad/Xtilde at M = 200 and 3e-3 at M = 1000; measured against the reference's fp64 run)
        wide = lambda t: t.double() if t.dtype == torch.float32 else t
        Xt_v = wide(self.Xtilde).unbind(0)
        dG_v = self.delta_G_list.unbind(0)
        wls_v = wide(self.warp_kernel_lengthscales).unbind(0)
        wvar_v = wide(self.warp_kernel_variances).unbind(0)
        Gt64 = wide(self.Gtilde)
        dls64, dvar64 = wide(self.data_kernel_lengthscale), wide(self.data_kernel_variance)
        slopes_v = self.mean_slopes.unbind(0)
        icpt_v = self.mean_intercepts.unbind(0)
        mu_z, resid = [], []
        for v in range(V):  # mean function at the inducing points + variational residual, one launch
            mz, dc = E.MeanResidFn.apply(Xt_v[v], slopes_v[v], icpt_v[v], dG_v[v],
                                         100.0 if self._is_fixed(v) else 1.0)  # x100: inert (quirk 7)
            mu_z.append(mz)
            resid.append(dc)
        cache.mu_z, cache.dG_v, cache.resid = mu_z, dG_v, resid

        # ---- everything M x M first: all prior covariances and variational covariances of the step
        #      are factorised by ONE batched Cholesky / triangular-inverse launch per matrix size
        rows_of = {v: {m: _as_index(view_idx[m][v], dev) for m in mods} for v in range(V)}
        # every non-fixed view keeps its prior factorisation and its KL terms, with or without rows in THIS
        # call: a data-parallel rank whose slice of a view is empty still owes its 1/world share of that
        # view's KL (the reference leaves NaN factors for an empty view and fails in kl_divergence)
        free = [v for v in range(V) if not self._is_fixed(v)]
        nonempty = [v for v in free if sum(rows_of[v][m][1] for m in mods) > 0]
        M_X, nG = self.Omega_sqt_G_list.shape[-1], self.Omega_sqt_G_list.shape[0]
        sizes_F = [self.Omega_sqt_F_dict[m].shape for m in mods]
        builtin = builtin_kind(self.kernel_func_warp) is not None and builtin_kind(self.kernel_func_data) is not None
        # when every matrix of the step has the same size and comes from a built-in covariance, the
        # kernels write straight into ONE [T, M, M] batch (no concatenation before the factorisation)
        stack, slot = None, None
        if builtin and all(sh[-1] == M_X for sh in sizes_F) and Gt64.shape[0] == M_X \
                and all(Xt_v[v].shape[0] == M_X for v in free):
            total = len(free) + 1 + nG + sum(sh[0] for sh in sizes_F)
            stack = torch.empty(total, M_X, M_X, dtype=f64, device=dev)
            cursor = [0]

            def slot(n):  # independent tensors over the batch's memory (not autograd views of it)
                a = cursor[0]
                cursor[0] += n
                return torch.empty(0, dtype=f64, device=dev).set_(
                    stack.untyped_storage(), a * M_X * M_X, (n, M_X, M_X), (M_X * M_X, M_X, 1))
        Kuu_w = {}
        for v in free:
            Z = Xt_v[v]
            Kuu_w[v] = self._kmat("warp", Z, Z, wls_v[v], wvar_v[v], self.diagonal_offset, f64, True,
                                  out=slot(1)[0] if slot else None)
        KuuF = self._kmat("data", Gt64, Gt64, dls64, dvar64, self.diagonal_offset, f64, True,
                          out=slot(1)[0] if slot else None)
        cache.Omega_G = E.OmegaFn.apply(self.Omega_sqt_G_list, slot(nG) if slot else None, True)
        cache.Om_fwd = cache.Omega_G.split(D, 0)                      # rows v*D+j  (forward, quirk 2)
        cache.Om_kl = cache.Omega_G.view(D, V, M_X, M_X).unbind(1)    # rows j*V+v  (KL, quirk 2)
        for m, sh in zip(mods, sizes_F):
            cache.Omega_F[m] = E.OmegaFn.apply(self.Omega_sqt_F_dict[m], slot(sh[0]) if slot else None, True)
        mats = [Kuu_w[v].unsqueeze(0) for v in free] + [KuuF.unsqueeze(0), cache.Omega_G] + \
               [cache.Omega_F[m] for m in mods]
        cache.batch = None
        if stack is not None:
            parts, whole = E.factor_batch(mats, stack=stack)
            cache.batch = (stack, whole[1], whole[2])  # matrices, inverses, logdets of the whole step
            cache.free = tuple(free)
        else:
            parts = E.factor_batch(mats)
        for i, v in enumerate(free):
            cache.warp[v] = (Kuu_w[v], E.Factor(parts=parts[i]))
        nf = len(free)
        cache.data = (KuuF, E.Factor(parts=parts[nf]))
        cache.Omega_G_fac = (parts[nf + 1][1], parts[nf + 1][2])
        cache.Omega_F_fac = {m: (parts[nf + 2 + i][1], parts[nf + 2 + i][2]) for i, m in enumerate(mods)}
        cache.flags.extend([whole[3]] if stack is not None else [p[3] for p in parts])

        # ---- warp GP per view (vgpsa.py:259-351) ---------------------------------------------------
        warp_out = {}
        # the views' warp GPs are independent and each fills only part of the chip (one view of 10k
        # spots = 157 workgroups on 256 CUs): run them on side streams so that they overlap; autograd
        # replays each node's backward on the stream of its forward, so the backward overlaps too
        side = self._side_streams(len(nonempty), dev) if (self.overlap_views and dev.type == "cuda"
                                                          and len(nonempty) > 1) else None
        main = torch.cuda.current_stream(dev) if side is not None else None
        for draw, v in enumerate(nonempty):
            if side is not None:
                side[draw].wait_stream(main)
            with (torch.cuda.stream(side[draw]) if side is not None else contextlib.nullcontext()):
                rows = rows_of[v]
                Xv = torch.cat([X_spatial[m][rows[m][0]] for m in mods], 0) if len(mods) > 1 else \
                    X_spatial[mods[0]][rows[mods[0]][0]]
                n = Xv.shape[0]
                Z = Xt_v[v]
                ls_u, var_u = wls_v[v], wvar_v[v]
                Kuu, fac = cache.warp[v]
                dc = resid[v]
                Om = cache.Om_fwd[v]  # quirk 2: forward uses rows v*D+j
                if noise is not None and noise["G"] is not None:
                    eps = noise["G"][draw].to(device=dev, dtype=torch.float32)
                elif dev.type == "cuda":  # one launch; the device generator never reproduces the CPU stream anyway
                    eps = self._draw([S, n, D], dev)
                else:  # S successive [n, D] draws, as Normal.rsample() in the reference's loop
                    eps = torch.stack([self._draw([n, D], dev) for _ in range(S)]) if S > 0 else \
                        torch.empty(0, n, D, device=dev)
                kind = builtin_kind(self.kernel_func_warp)
                if kind is not None:  # covariance, projection, contractions and the draws as one fp64 node
                    Gm, Gs, bad, Gs64 = E.SGPWarpLayerFn.apply(kind, Z, Xv, ls_u, var_u, Kuu, dc, Om, fac,
                                                               slopes_v[v], icpt_v[v], eps)  # quirk 1 inside
                else:
                    Kuf = self._kmat("warp", Z, Xv, ls_u, var_u, 0.0, f64, False)
                    meanT, vq, q = E.SGPCoreFn.apply(Kuu, Kuf, dc, Om, fac, f64)
                    Gm, Gs, bad, Gs64 = E.WarpSampleFn.apply(meanT, vq, q, var_u, Xv, slopes_v[v], icpt_v[v], eps)
                if side is not None:  # consumed on the main stream from here on
                    for t in (Gm, Gs, bad, Gs64):
                        t.record_stream(main)
            cache.flags.append(bad)
            warp_out[v] = (Gm, Gs, Gs64)
        if side is not None:
            for draw in range(len(nonempty)):
                main.wait_stream(side[draw])

        # ---- assemble G_means [N,D] / G_samples [S,N,D] per modality --------------------------------
        # G64[m]: G_samples[m] before its rounding to the fp32 API tensor (fp64 draws of the warp GPs, the raw
        # coordinates of fixed views).  The data GP's covariance is evaluated on it: handing the rounded
        # tensor over costs 1e-4 on F at short lengthscales (measured on the 1-D golden case)
        G_means, G_samples, G64 = {}, {}, {}
        for mi, m in enumerate(mods):
            # fast path: the views are consecutive row blocks covering 0..N (what create_view_idx_dict
            # produces): concatenate (its backward is a set of views, no zero-fill / copy / add)
            edges, ok = 0, True
            for v in range(V):
                r, cnt = rows_of[v][m]
                ok = ok and isinstance(r, slice) and r.start == edges and (self._is_fixed(v) or v in warp_out or cnt == 0)
                edges += cnt
            ok = ok and edges == int(Ns[m])
            if ok:
                pm, ps, p64 = [], [], []
                for v in range(V):
                    r, cnt = rows_of[v][m]
                    if cnt == 0:
                        continue
                    if self._is_fixed(v):  # vgpsa.py:262-273
                        xv = X_spatial[m][r]
                        pm.append(xv)
                        ps.append(xv.unsqueeze(0).expand(S, -1, -1))
                        p64.append(ps[-1])
                    else:
                        a = sum(rows_of[v][mm][1] for mm in mods[:mi])
                        Gm, Gs, Gs64 = warp_out[v]
                        pm.append(Gm[a : a + cnt])
                        ps.append(Gs[:, a : a + cnt])
                        p64.append(Gs64[:, a : a + cnt])
                G_means[m] = torch.cat(pm, 0) if len(pm) > 1 else pm[0].clone()
                G_samples[m] = torch.cat(ps, 1) if len(ps) > 1 else ps[0].clone()
                if S > 0 and X_spatial[m].dtype == torch.float32:  # differentiable: the gradient returns in fp64
                    G64[m] = torch.cat([t.to(f64) for t in p64], 1) if len(p64) > 1 else p64[0].to(f64)
                continue
            nan = float("nan")  # general index sets / empty views: NaN-filled scatter like the reference
            G_means[m] = torch.full([int(Ns[m]), D], nan, device=dev)
            G_samples[m] = torch.full([S, int(Ns[m]), D], nan, device=dev)
            for v in range(V):
                r, cnt = rows_of[v][m]
                if self._is_fixed(v):
                    G_means[m][r] = X_spatial[m][r]
                    G_samples[m][:, r, :] = X_spatial[m][r]
                elif v in warp_out:
                    a = sum(rows_of[v][mm][1] for mm in mods[:mi])
                    Gm, Gs, _ = warp_out[v]
                    G_means[m][r] = Gm[a : a + cnt]
                    G_samples[m][:, r, :] = Gs[:, a : a + cnt]

        # every flag of this forward exists now (factorisations, warp variances): ship them to the host
        # behind an event BEFORE the data GP is queued, so that the check at the end of forward waits for
        # the warp GP only and the host keeps queueing while the data GP's kernels run
        pending = self._post_flags(cache) if self.check_numerics else None

        # ---- data GP (vgpsa.py:353-477) ----------------------------------------------------------
        ls_u, var_u = dls64, dvar64
        KuuF, facF = cache.data

        def data_layer(G, eps_key, m, G64m=None):
            S_, N_ = G.shape[0], G.shape[1]
            L = self.n_latent_outputs[m]
            Gf = G.reshape(S_ * N_, D)
            Gf64 = None if G64m is None else G64m.reshape(S_ * N_, D)
            # covariance + whitening in fp64 (gradient-only fp32 backward); mean / variance form in fp32
            if noise is not None and noise[eps_key] is not None:
                eps = noise[eps_key][m].to(device=dev, dtype=torch.float32).reshape(S_ * N_, L)
            else:
                eps = self._draw([S_, N_, L], dev, "F").reshape(S_ * N_, L)
            kind = builtin_kind(self.kernel_func_data)
            if kind is not None:  # covariance, projection, contractions and the draw as one node
                Fl = E.SGPDataLayerFn.apply(kind, Gt64, Gf, ls_u, var_u, KuuF, self.delta_F_dict[m],
                                            cache.Omega_F[m], facF, eps, Gf64).reshape(S_, N_, L)
            else:
                Kuf = self._kmat("data", Gt64, Gf, ls_u, var_u, 0.0, f64, False, torch.float32)
                meanT, vq, q = E.SGPCoreFn.apply(
                    KuuF, Kuf, self.delta_F_dict[m], cache.Omega_F[m], facF, torch.float32
                )
                Fl = E.DataSampleFn.apply(meanT, vq, q, var_u, eps).reshape(S_, N_, L)
            if self.n_latent_gps[m] is not None:
                return Fl, E.MatmulFn.apply(Fl, self.W_dict[m])
            return Fl, Fl  # same tensor object when there is no LMC (quirk 10)

        self.F_latent_samples, self.F_observed_samples = {}, {}
        if G_test is not None:
            self.F_latent_samples_test, self.F_observed_samples_test = {}, {}
        for m in mods:
            self.F_latent_samples[m], self.F_observed_samples[m] = data_layer(G_samples[m], "F", m, G64.get(m))
            if G_test is not None:
                Gt = G_test[m].to(device=dev, dtype=torch.float32)
                lt, ot = data_layer(Gt, "F_test", m)
                self.F_latent_samples_test[m], self.F_observed_samples_test[m] = lt, ot

        self._cache = cache
        if pending is not None:
            self._raise_on_flags(pending)
        if G_test is not None:
            return (
                G_means,
                G_samples,
                self.F_latent_samples,
                self.F_observed_samples,
                self.F_latent_samples_test,
                self.F_observed_samples_test,
            )
        return G_means, G_samples, self.F_latent_samples, self.F_observed_samples

    def _lazy_obs(self, plan, G_test, prediction_mode, grads):
        """per modality: True = an LMC modality whose F_obs = F_latent W is left to whoever asks for it (training;
        loss_fn runs gpsa_lmc_loglik_fused_f32 on (F_latent, W, Y) instead of forming it).  ``grads``: gradients are
        enabled and a parameter wants one"""
        training = self.fuse_elbo and not prediction_mode and G_test is None and grads
        return [bool(training and plan.lmc[i] and plan.L[i] <= 64 and self.W_dict[m].dtype == torch.float32
                     and self.W_dict[m].is_contiguous())
                for i, m in enumerate(self.modality_names)]

    def _fuse_setup(self, plan, S, G_test, prediction_mode, grads):
        """-> the ``fuse`` record of a training forward that leaves its fusable data GPs to loss_fn, or None"""
        if not self.fuse_elbo or prediction_mode or G_test is not None or not grads:
            return None
        mods = self.modality_names
        flags = [not plan.lmc[i] and bool(plan.lib.gpsa_step_fused(plan.handle, i)) for i in range(len(mods))]
        nz = self.noise_variance
        if not any(flags) or nz.dtype != torch.float32 or not nz.is_contiguous():
            return None
        Mg = int(self.Gtilde.shape[0])
        if max(2.0 * S * plan.N[i] * plan.L[i] * Mg * Mg for i in range(len(mods)) if flags[i]) < self.fuse_min_flops:
            return None  # a launch-bound step: the separate kernels in one go are cheaper (see fuse_min_flops)
        nn_ = nz.numel()
        return dict(mods=flags, Y=[None] * len(mods), noise=nz.detach(),
                    noise_ptr=[nz.data_ptr() + 4 * (nn_ - self.n_modalities + i) for i in range(len(mods))],  # quirk 5
                    shapes=[(int(S), plan.N[i], plan.L[i]) for i in range(len(mods))], gloss=None,
                    # per modality: "lazy" (handle untouched) -> "fused" (loss_fn ran the fused pass) or "real"
                    # (materialised first: an unfused modality from then on); see lazy.py
                    state=["lazy" if z else None for z in flags], dF=[None] * len(mods), F_real=[None] * len(mods),
                    live=None, FT=[None] * len(mods))

    def _forward_engine(self, X_spatial, rows, S, G_test, noise, prediction_mode):
        """forward through the C++ step engine: one autograd node, one host call each way"""
        dev = self.Xtilde.device
        mods = self.modality_names
        V, D = self.n_views, self.n_spatial_dims
        f32 = torch.float32
        test_shapes = None
        if G_test is not None:
            Gt = [G_test[m].to(device=dev, dtype=f32).contiguous() for m in mods]
            test_shapes = (int(Gt[0].shape[0]), tuple(int(g.shape[1]) for g in Gt))
        plan = SE.get_plan(self, rows, S, test_shapes, want_kl=not prediction_mode)
        # the engine takes raw pointers and trusts the plan's shapes: everything handed over is checked against it
        for i, m in enumerate(mods):
            if tuple(X_spatial[m].shape) != (plan.N[i], D):
                raise ValueError(f"X_spatial[{m!r}] has shape {tuple(X_spatial[m].shape)}, the views' row counts "
                                 f"give ({plan.N[i]}, {D})")
            if G_test is not None and (Gt[i].dim() != 3 or tuple(Gt[i].shape) != (plan.s_test, plan.n_test[i], D)):
                raise ValueError(f"G_test[{m!r}] has shape {tuple(Gt[i].shape)}: every modality needs "
                                 f"[{plan.s_test} (samples of the first modality), n_test, {D}]")
        # draws: one buffer for the warp GPs of all free, non-empty views (order of vgpsa.py:346-348), one per
        # modality for the data GP; without injected noise or per-purpose generators ALL of them come out of a
        # single launch
        nm_ = len(mods)
        shapes_F = [[S, plan.N[i], plan.L[i]] for i in range(nm_)]
        shapes_Ft = [[plan.s_test, plan.n_test[i], plan.L[i]] for i in range(nm_)] if G_test is not None else []
        eps_G, eps_F, eps_Ft = None, [], []
        if noise is None and self.noise_generators is None:
            sizes = [plan.eps_g_numel] + [s_[0] * s_[1] * s_[2] for s_ in shapes_F + shapes_Ft]
            # 64-float (256-byte) aligned pieces of one buffer
            offs, tot = [], 0
            for n_ in sizes:
                offs.append(tot)
                tot += (n_ + 63) // 64 * 64
            buf = self._draw([max(tot, 1)], dev, "G")
            eps_G = buf[: sizes[0]]
            for i in range(nm_):
                eps_F.append(buf[offs[1 + i]: offs[1 + i] + sizes[1 + i]].view(shapes_F[i]))
            for i in range(len(shapes_Ft)):
                k_ = 1 + nm_ + i
                eps_Ft.append(buf[offs[k_]: offs[k_] + sizes[k_]].view(shapes_Ft[i]))
        else:
            if noise is not None and noise["G"] is not None:
                eps_G = torch.cat([e.to(device=dev, dtype=f32).reshape(-1) for e in noise["G"]]) if noise["G"] else \
                    torch.empty(0, dtype=f32, device=dev)
                if eps_G.numel() != plan.eps_g_numel:
                    raise ValueError(f"injected eps_G has {eps_G.numel()} values, the free views need "
                                     f"{plan.eps_g_numel}")
            else:
                eps_G = self._draw([plan.eps_g_numel], dev, "G")
            for i, m in enumerate(mods):
                if noise is not None and noise["F"] is not None:
                    if noise["F"][m].numel() != S * plan.N[i] * plan.L[i]:
                        raise ValueError(f"injected eps_F[{m!r}] has {noise['F'][m].numel()} values, "
                                         f"[S, N, L] = {shapes_F[i]} are needed")
                    eps_F.append(noise["F"][m].to(device=dev, dtype=f32).reshape(shapes_F[i]).contiguous())
                else:
                    eps_F.append(self._draw(shapes_F[i], dev, "F"))
                if G_test is not None:
                    if noise is not None and noise["F_test"] is not None:
                        if noise["F_test"][m].numel() != plan.s_test * plan.n_test[i] * plan.L[i]:
                            raise ValueError(f"injected eps_F_test[{m!r}] has {noise['F_test'][m].numel()} values, "
                                             f"{shapes_Ft[i]} are needed")
                        eps_Ft.append(noise["F_test"][m].to(device=dev, dtype=f32).reshape(shapes_Ft[i]).contiguous())
                    else:
                        eps_Ft.append(self._draw(shapes_Ft[i], dev, "F"))
        stale = self.__dict__.get("_pending_flag")
        if stale is not None:  # an earlier forward's deferred check whose backward never ran
            self._pending_flag = None
            self._raise_on_flags(stale)
        check = self.check_numerics
        plist = SE._param_list(self)
        training = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
        if check is True and training:
            check = "deferred"
        elif check == "strict":
            check = True
        # (a weak reference: the node must not close a cycle model -> outputs -> node -> model, or its arena would
        #  wait for the cyclic collector when no backward ever runs)
        aux = dict(plan=plan, model=weakref.ref(self), X=[X_spatial[m].contiguous() for m in mods], eps_G=eps_G, eps_F=eps_F,
                   G_test=Gt if G_test is not None else None, eps_F_test=eps_Ft,
                   slopes=self.mean_slopes.contiguous(), intercepts=self.mean_intercepts.contiguous(),
                   want_kl=not prediction_mode, check=check, no_keep=not self.keep_products,
                   mm_epoch=self.__dict__.get("_mm_epoch"), bwd_acc=self.__dict__.get("_bwd_acc"),
                   shared_arena=self.__dict__.get("_mb_arena") if self.__dict__.get("_mm_epoch") is not None else None,
                   flag_slot=self.__dict__.get("_flag_slot", 0),
                   fuse=self._fuse_setup(plan, S, G_test, prediction_mode, training),
                   lazy_obs=self._lazy_obs(plan, G_test, prediction_mode, training))
        self.__dict__["_flag_slot"] = 1 - aux["flag_slot"]  # two pinned words: consecutive forwards never share one
        outs = SE.StepFn.apply(aux, *plist)
        nm = len(mods)
        lmc = [i for i in range(nm) if plan.lmc[i]]
        k = 0
        take = lambda n: outs[k:k + n]
        Gm, k = outs[k:k + nm], k + nm
        Gs, k = outs[k:k + nm], k + nm
        Fl, k = outs[k:k + nm], k + nm
        Fo, k = outs[k:k + len(lmc)], k + len(lmc)
        Flt, Fot = (), ()
        if G_test is not None:
            Flt, k = outs[k:k + nm], k + nm
            Fot, k = outs[k:k + len(lmc)], k + len(lmc)
        cache = _StepCache()
        cache.kl = outs[k] if not prediction_mode else None
        cache.mu_z, cache.engine_flag = aux["mu_z"], aux["flag"]
        cache.fuse = aux["fuse"]
        cache.plan, cache.arena_ref = plan, aux.get("arena_ref")  # (the hand-off attributes read the arena's batch)
        if cache.fuse is not None:
            # a fused modality's "F_latent" output is the vector of partial sums its likelihood finishes from: what the
            # caller gets is a handle with the draws' shape that loss_fn recognises and that materialises on any other use
            Fl = list(Fl)
            for i in range(nm):
                if cache.fuse["mods"][i]:
                    h = LazyDraws(cache.fuse, i, (S, plan.N[i], plan.L[i]), dev)
                    h._parts = Fl[i]
                    Fl[i] = h
        if any(aux["lazy_obs"]):
            Fo = list(Fo)
            for i, m in enumerate(mods):
                if aux["lazy_obs"][i]:
                    Wm, Fli = self.W_dict[m], Fl[i]
                    Fo[lmc.index(i)] = LazyProduct(lambda Fli=Fli, Wm=Wm: E.MatmulFn.apply(Fli, Wm),
                                                   (S, plan.N[i], plan.P[i]), dev, Fli, Wm)
        G_means = {m: Gm[i] for i, m in enumerate(mods)}
        G_samples = {m: Gs[i] for i, m in enumerate(mods)}
        # (plain attributes through the instance dict: nn.Module.__setattr__ costs ~8 us a piece on a 0.5 ms step)
        d_ = self.__dict__
        d_["F_latent_samples"] = {m: Fl[i] for i, m in enumerate(mods)}
        # the same tensor object without LMC (quirk 10)
        d_["F_observed_samples"] = {m: (Fo[lmc.index(i)] if plan.lmc[i] else Fl[i]) for i, m in enumerate(mods)}
        d_["_cache"] = cache
        if aux.get("deferred") is not None:
            self.__dict__["_pending_flag"] = aux["deferred"]
        if aux["pending"] is not None:
            self._raise_on_flags(aux["pending"])
        if G_test is not None:
            self.F_latent_samples_test = {m: Flt[i] for i, m in enumerate(mods)}
            self.F_observed_samples_test = {m: (Fot[lmc.index(i)] if plan.lmc[i] else Flt[i])
                                            for i, m in enumerate(mods)}
            return (G_means, G_samples, self.F_latent_samples, self.F_observed_samples,
                    self.F_latent_samples_test, self.F_observed_samples_test)
        return G_means, G_samples, self.F_latent_samples, self.F_observed_samples

    def _flag_host(self, slot=0):
        """one of two pinned (device-mapped) host words the engine's numerics reduction writes into"""
        hosts = self.__dict__.get("_flag_hosts")
        if hosts is None:
            hosts = self.__dict__["_flag_hosts"] = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        return hosts[slot]

    def _post_flag(self, host):
        """an event behind the kernel that wrote ``host``; returns what _raise_on_flags waits on.  The events are kept
        (one per pinned word): a fresh torch.cuda.Event per forward cost 190 us of host time per step on this stack
        (creation + first record) - a quarter of a step at BASELINE config 1's size."""
        evs = self.__dict__.setdefault("_flag_events", {})
        ev = evs.get(host.data_ptr())
        if ev is None:
            ev = evs[host.data_ptr()] = torch.cuda.Event()
        ev.record()
        return host, ev

    def _post_flags(self, cache):
        """max |flag| of this forward -> host, asynchronously; returns what _raise_on_flags waits on"""
        if not cache.flags:
            return None
        worst = torch.cat([f.reshape(-1) for f in cache.flags]).max()  # int32, all >= 0
        if worst.device.type != "cuda":
            return worst, None
        host = self.__dict__.get("_flag_host")
        if host is None:
            host = self.__dict__["_flag_host"] = torch.zeros(1, dtype=torch.int32).pin_memory()
        host.copy_(worst.reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    @staticmethod
    def _raise_on_flags(pending):
        host, ev = pending
        if ev is not None:
            ev.synchronize()  # the one host wait of forward: up to the end of the warp GPs
        if int(host.reshape(-1)[0].item()) != 0:
            raise torch.linalg.LinAlgError(
                "GPSA forward: an inducing-point covariance is not positive-definite or a warp "
                "variance is not positive (the reference raises from torch.cholesky / "
                "Normal(...) argument validation here)"
            )

    def _kl_grouped(self, cache):
        """sum of all KL terms through MvnKLGroupedFn; term order = order of the variational covariances
        in the factorisation batch: Omega_G rows r = j*V + v (quirk 2: the KL pairs row r with view
        r % V and coordinate r // V), then every Omega_F row."""
        V, D = self.n_views, self.n_spatial_dims
        mods, free = self.modality_names, cache.free
        key = (free, tuple(int(cache.Omega_F[m].shape[0]) for m in mods))
        plan = self.__dict__.setdefault("_kl_plans", {}).get(key)
        if plan is None:
            pos = {v: i for i, v in enumerate(free)}
            prior = [pos.get(r % V, -1) for r in range(V * D)]
            for m in mods:
                prior += [len(free)] * int(cache.Omega_F[m].shape[0])
            plan = self._kl_plans[key] = E.KLPlan(prior, len(free) + 1, cache.Omega_G.device)
        M = cache.Omega_G.shape[-1]
        rows = [torch.stack(cache.resid, 0).permute(2, 0, 1).reshape(V * D, M)]  # row j*V+v = resid[v][:, j]
        rows += [self.delta_F_dict[m].t() for m in mods]
        Dall = torch.cat(rows, 0)  # promoted to fp64
        priors = [cache.warp[v][0] for v in free] + [cache.data[0]]
        kl = E.MvnKLGroupedFn.apply(plan, cache.batch, Dall, *priors, cache.Omega_G,
                                    *[cache.Omega_F[m] for m in mods])
        own = SE.kl_own_range(self)
        if own is not None:  # owner computes (this path evaluates every term and drops the others': same numbers, no
            keep = torch.zeros_like(kl)  # saving - the saving is the step engine's)
            keep[own[0]:own[1]] = 1.0
            kl = kl * keep
        if self.kl_weight_G != 1.0:  # output-sharded rank: its share of the warp GPs' terms
            return kl[: V * D].sum() * self.kl_weight_G + kl[V * D:].sum()
        return kl.sum()

    # ------------------------------------------------------------------------------------------
    def loss_fn(self, data_dict, F_samples):
        """Negative (approximate) ELBO (vgpsa.py:491-540).  Valid only after ``forward`` on the same
        parameters: like the reference it reads the factorisations forward left behind."""
        cache = self._cache
        if cache is None:
            raise AttributeError("loss_fn called before forward (no factorisations cached)")
        V, D = self.n_views, self.n_spatial_dims
        if cache.kl is not None:  # forward ran through the step engine: the KL terms came out of its node
            kl = cache.kl if self.kl_scale != 0 else None  # (a slice without a KL share: no KL backward either)
            if kl is not None and self.kl_weight_G != 1.0:  # output-sharded rank: its share of the warp GPs' terms
                kl = torch.cat([kl[: V * D] * self.kl_weight_G, kl[V * D:]])
            nn_ = self.noise_variance.numel()
            aux = dict(Y=[data_dict[m]["outputs"] for m in self.modality_names],
                       noise_idx=[nn_ - self.n_modalities + i for i in range(self.n_modalities)],  # quirk 5
                       kl_scale=self.kl_scale)
            fuse = getattr(cache, "fuse", None)
            Fs, eff, run_i, run_Y, run_parts = [], [], [], [], []
            lmc_terms, Ws, shapes = {}, [], [None] * self.n_modalities
            for i, m in enumerate(self.modality_names):
                F = F_samples[m]
                take = False
                if isinstance(F, LazyProduct):
                    # an LMC modality whose F_obs = F_latent W nobody has asked for: the likelihood, dF_latent and dW come
                    # out of one pass over (F_latent, W, Y) (gpsa_lmc_loglik_fused_f32) when the observations are what
                    # that kernel reads; otherwise the product is formed and the separate kernels run
                    Y = data_dict[m]["outputs"]
                    Fl_, W_ = F._lmc
                    if (not F.is_materialized and torch.is_tensor(Y) and Y.is_cuda and Y.device == F.device
                            and Y.dtype == torch.float32 and Y.is_contiguous()
                            and tuple(Y.shape) == (int(F.shape[1]), int(F.shape[2])) and W_ is self.W_dict[m]):
                        take = True
                        lmc_terms[i] = len(Ws)
                        Ws.append(W_)
                        shapes[i] = tuple(int(d) for d in F.shape)
                        F = Fl_
                    else:
                        F = F.materialize()
                elif isinstance(F, LazyDraws):
                    # forward left this modality's data GP to us: run it with the likelihood folded in when the handle is
                    # that forward's own, untouched, and the observations are what the fused kernel reads (fp32, on the
                    # device, [N, L]); anything else gets the draws themselves (the handle materialises)
                    Y = data_dict[m]["outputs"]
                    mine = fuse is not None and F._rec is fuse
                    fresh = mine and fuse["state"][i] == "lazy"
                    take = mine and fuse["state"][i] == "fused" and torch.is_tensor(Y) and Y is fuse["Y"][i]
                    if fresh and (torch.is_tensor(Y) and Y.is_cuda and Y.device == F.device and Y.dtype == torch.float32
                                  and Y.is_contiguous() and tuple(Y.shape) == (fuse["shapes"][i][1], fuse["shapes"][i][2])):
                        take = True
                        fuse["state"][i] = "fused"
                        run_i.append(i)
                        run_Y.append(Y)
                        run_parts.append(F._parts)
                    F = F._parts if take else F.materialize()
                Fs.append(F)
                eff.append(take)
            if run_i:
                from ..lazy import run_fused

                run_fused(fuse, run_i, run_Y, run_parts)
            if any(eff):
                for i in range(self.n_modalities):
                    if eff[i] and shapes[i] is None:
                        shapes[i] = tuple(fuse["shapes"][i])
                aux["fuse"], aux["fuse_mods"], aux["lmc"], aux["term_shapes"] = fuse, eff, lmc_terms, shapes
            loss = SE.ElboLossFn.apply(aux, self.noise_variance, kl, *Fs, *Ws)
            return loss.to(self.Xtilde.dtype)
        f64 = torch.float64
        kl = None
        grouped = cache.batch is not None
        if grouped:  # every KL term of the step (all views, all outputs) in one launch each way
            kl = self._kl_grouped(cache)
        own = None if grouped else SE.kl_own_range(self)

        def owned(terms, first, stride):  # term i of ``terms`` is global term first + i * stride (owner computes)
            if own is None:
                return terms
            idx = first + stride * torch.arange(terms.shape[0], device=terms.device)
            return terms * ((idx >= own[0]) & (idx < own[1])).to(terms.dtype)

        for v in range(V):
            if grouped or self._is_fixed(v) or v not in cache.warp:
                continue
            Kuu, fac = cache.warp[v]
            Dm = cache.resid[v]
            Om = cache.Om_kl[v]  # quirk 2: the KL uses rows j*V+v, j = 0..D-1
            ofac = (cache.Omega_G_fac[0][v::V], cache.Omega_G_fac[1][v::V])
            term = owned(E.MvnKLFn.apply(Kuu, Dm, Om, fac, ofac), v, V).sum()
            if self.kl_weight_G != 1.0:
                term = term * self.kl_weight_G
            kl = term if kl is None else kl + term
        KuuF, facF = cache.data
        lls = []
        l_first = V * D
        for i, m in enumerate(self.modality_names):
            if not grouped:
                term = owned(E.MvnKLFn.apply(KuuF, self.delta_F_dict[m], cache.Omega_F[m], facF,
                                             cache.Omega_F_fac[m]), l_first, 1).sum()
                l_first += int(self.n_latent_outputs[m])
                kl = term if kl is None else kl + term
            noise_u = self.noise_variance[-self.n_modalities + i]  # quirk 5 (used as a std)
            Y = data_dict[m]["outputs"]
            lls.append(E.LogLikFn.apply(F_samples[m], Y, noise_u))
        ll = lls[0] if len(lls) == 1 else torch.stack(lls)
        # -LL + kl_scale * KL in one launch (kl: the per-term vector of the grouped path, or a scalar)
        return E.ElboFn.apply(ll, kl, self.kl_scale).to(self.Xtilde.dtype)
