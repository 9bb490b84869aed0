"""``GPSA``: the generative-model parameters every GPSA variant shares.

Drop-in surface of the reference base class (gpsa/models/gpsa.py:9-197): identical constructor
keywords and defaults, attribute / parameter names (``state_dict()`` interchanges with the
reference), ``create_view_idx_dict`` and ``compute_mean_penalty``.

Where this differs in mechanics, not in behaviour: parameters are created on the CPU and follow
``model.to(device)`` like any ``nn.Module`` (the reference's user script does that too,
examples/grid_example.py:44-57); the constant mean-function tensors and the ``fixed_*``
hyper-parameter tensors (plain tensors upstream) are non-persistent buffers so that they move along
and stay out of ``state_dict()``.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ..kernels import rbf_kernel

DIAGONAL_OFFSET = 1e-5  # gpsa/models/gpsa.py:153


def _the_one(values, what):
    """all modalities must agree on ``what``"""
    distinct = set(values)
    if len(distinct) != 1:
        raise ValueError(f"Each modality must have the same number of {what}.")
    return distinct.pop()


class GPSA(nn.Module):
    """Base class.

    ``data_dict``: ``{modality: {"spatial_coords": X [N,D], "outputs": Y [N,P],
    "n_samples_list": [n_1..n_V]}}``.  Remaining keywords as upstream (gpsa/models/gpsa.py:25-38):
    ``mean_function`` in {"identity_fixed", "identity_initialized", None}; a ``fixed_*`` list pins the
    corresponding kernel hyper-parameter (stored as its log, not trained).
    """

    def __init__(self, data_dict, data_init=True, n_spatial_dims=2, n_noise_variance_params=2,
                 kernel_func_warp=rbf_kernel, kernel_func_data=rbf_kernel,
                 mean_function="identity_fixed", mean_penalty_param=0.0,
                 fixed_warp_kernel_variances=None, fixed_warp_kernel_lengthscales=None,
                 fixed_data_kernel_lengthscales=None):
        super().__init__()
        mods = list(data_dict)
        self.modality_names, self.n_modalities = mods, len(mods)
        self.mean_penalty_param = mean_penalty_param
        self.n_views = _the_one((len(data_dict[m]["n_samples_list"]) for m in mods), "views")
        # the spatial dimension comes from the data; the keyword is ignored upstream as well
        self.n_spatial_dims = _the_one(
            (int(data_dict[m]["spatial_coords"].shape[1]) for m in mods), "spatial dimensions")
        self.view_idx, self.Ns, self.Ps, self.n_total = self.create_view_idx_dict(data_dict)

        V, D = self.n_views, self.n_spatial_dims
        self.n_kernel_params = 2 * V + 2  # (lengthscale, variance) per warp GP + the data GP's pair
        self.n_noise_variance_params = n_noise_variance_params
        self.kernel_func_warp, self.kernel_func_data = kernel_func_warp, kernel_func_data
        self.diagonal_offset = DIAGONAL_OFFSET

        def learn_or_pin(name, pinned, initial):
            """nn.Parameter(initial()) or, if ``pinned`` is given, a constant log(pinned) buffer"""
            if pinned is None:
                setattr(self, name, nn.Parameter(initial()))
            else:
                self.register_buffer(name, torch.log(torch.as_tensor(pinned, dtype=torch.float32)),
                                     persistent=False)

        # creation order == RNG consumption order of gpsa/models/gpsa.py:86-150
        self.noise_variance = nn.Parameter(torch.randn([n_noise_variance_params]) - 1)
        learn_or_pin("warp_kernel_variances", fixed_warp_kernel_variances, lambda: torch.zeros(V))
        learn_or_pin("warp_kernel_lengthscales", fixed_warp_kernel_lengthscales,
                     lambda: torch.full([V], math.log(10.0)))
        learn_or_pin("data_kernel_lengthscale", fixed_data_kernel_lengthscales,
                     lambda: torch.log(torch.exp(torch.randn(1))))
        self.data_kernel_variance = nn.Parameter(torch.randn(1))

        identity = torch.eye(D).unsqueeze(0).repeat(V, 1, 1)
        if mean_function == "identity_fixed":  # constants, not trained
            self.register_buffer("mean_slopes", identity, persistent=False)
            self.register_buffer("mean_intercepts", torch.zeros(V, D), persistent=False)
        else:
            random_start = mean_function == "identity_initialized"
            self.mean_slopes = nn.Parameter(torch.randn([V, D, D]) if random_start else identity)
            self.mean_intercepts = nn.Parameter(
                torch.zeros([V, D]) if random_start else torch.randn([V, D]) * 0.1)

    def create_view_idx_dict(self, data_dict):
        """``(view_idx, Ns, Ps, n_total)``: per modality the row indices of every view (consecutive
        blocks in ``n_samples_list`` order), the row count, the output count; and the grand total of
        rows (gpsa/models/gpsa.py:155-183)."""
        view_idx, Ns, Ps = {}, {}, {}
        for mod in self.modality_names:
            sizes = np.asarray(data_dict[mod]["n_samples_list"])
            stops = np.cumsum(sizes)
            view_idx[mod] = [np.arange(stop - size, stop) for size, stop in zip(sizes, stops)][: self.n_views]
            Ns[mod] = np.sum(sizes)
            Ps[mod] = data_dict[mod]["outputs"].shape[1]
        return view_idx, Ns, Ps, sum(Ns.values())

    def compute_mean_penalty(self):
        """ridge on the deviation of the mean slopes from the identity (gpsa/models/gpsa.py:185-191;
        never added to the loss upstream either)"""
        dev = self.mean_slopes - torch.eye(self.n_spatial_dims, device=self.mean_slopes.device)
        return self.mean_penalty_param * dev.square().mean()

    def forward(self, X_spatial):
        raise NotImplementedError

    def loss_fn(self, data_dict, Gs, means_G_list, covs_G_list, means_Y, covs_Y):
        raise NotImplementedError
