"""``GPSA``: generative-model parameters shared by all GPSA variants.

Drop-in surface of the reference's base class (gpsa/models/gpsa.py:9-197): same constructor
keywords, attribute and parameter names (so ``state_dict()`` interchanges with the reference),
``create_view_idx_dict`` and ``compute_mean_penalty``.  Parameters are created on the CPU and moved
with ``model.to(device)`` like any ``nn.Module`` (the reference user script does exactly that,
examples/grid_example.py:44-57); the constant mean-function tensors and the ``fixed_*`` hyper-parameter
tensors are non-persistent buffers so that they move along.
"""
import numpy as np
import torch
import torch.nn as nn

from ..kernels import rbf_kernel

DIAGONAL_OFFSET = 1e-5  # gpsa/models/gpsa.py:153


class GPSA(nn.Module):
    """
    Args:
        data_dict (dict): {"modality": {"spatial_coords": X [N,D], "outputs": Y [N,P],
            "n_samples_list": [n_1..n_V]}}; every modality must have the same number of views and
            of spatial dimensions.
        data_init, n_spatial_dims, n_noise_variance_params, kernel_func_warp, kernel_func_data,
        mean_function ("identity_fixed" | "identity_initialized" | None), mean_penalty_param,
        fixed_warp_kernel_variances, fixed_warp_kernel_lengthscales, fixed_data_kernel_lengthscales:
            as in the reference (gpsa/models/gpsa.py:25-38).
    """

    def __init__(
        self,
        data_dict,
        data_init=True,
        n_spatial_dims=2,
        n_noise_variance_params=2,
        kernel_func_warp=rbf_kernel,
        kernel_func_data=rbf_kernel,
        mean_function="identity_fixed",
        mean_penalty_param=0.0,
        fixed_warp_kernel_variances=None,
        fixed_warp_kernel_lengthscales=None,
        fixed_data_kernel_lengthscales=None,
    ):
        super().__init__()
        self.modality_names = list(data_dict.keys())
        self.n_modalities = len(self.modality_names)
        self.mean_penalty_param = mean_penalty_param

        view_counts = {len(data_dict[m]["n_samples_list"]) for m in self.modality_names}
        if len(view_counts) != 1:
            raise ValueError("Each modality must have the same number of views.")
        self.n_views = view_counts.pop()

        dims = {int(data_dict[m]["spatial_coords"].shape[1]) for m in self.modality_names}
        if len(dims) != 1:
            raise ValueError("Each modality must have the same number of spatial dimensions.")
        self.n_spatial_dims = dims.pop()  # derived from the data; the argument is ignored upstream too

        self.view_idx, self.Ns, self.Ps, self.n_total = self.create_view_idx_dict(data_dict)

        V, D = self.n_views, self.n_spatial_dims
        self.n_kernel_params = 2 * V + 2
        self.n_noise_variance_params = n_noise_variance_params
        self.kernel_func_warp = kernel_func_warp
        self.kernel_func_data = kernel_func_data
        self.diagonal_offset = DIAGONAL_OFFSET

        # parameter creation order (and hence RNG consumption) follows gpsa/models/gpsa.py:86-150
        self.noise_variance = nn.Parameter(torch.randn([n_noise_variance_params]) - 1)

        if fixed_warp_kernel_variances is None:
            self.warp_kernel_variances = nn.Parameter(torch.zeros(V))
        else:
            self.register_buffer(
                "warp_kernel_variances", torch.log(torch.tensor(fixed_warp_kernel_variances)), False
            )
        if fixed_warp_kernel_lengthscales is None:
            self.warp_kernel_lengthscales = nn.Parameter(torch.zeros(V) + float(np.log(10)))
        else:
            self.register_buffer(
                "warp_kernel_lengthscales", torch.log(torch.tensor(fixed_warp_kernel_lengthscales)), False
            )
        if fixed_data_kernel_lengthscales is None:
            self.data_kernel_lengthscale = nn.Parameter(torch.log(torch.exp(torch.randn(1))))
        else:
            self.register_buffer(
                "data_kernel_lengthscale",
                torch.log(torch.tensor(fixed_data_kernel_lengthscales).float()),
                False,
            )
        self.data_kernel_variance = nn.Parameter(torch.randn(1))

        eye = torch.eye(D).unsqueeze(0).repeat(V, 1, 1)
        if mean_function == "identity_fixed":
            self.register_buffer("mean_slopes", eye, False)
            self.register_buffer("mean_intercepts", torch.zeros(V, D), False)
        elif mean_function == "identity_initialized":
            self.mean_slopes = nn.Parameter(torch.randn([V, D, D]))
            self.mean_intercepts = nn.Parameter(torch.zeros([V, D]))
        else:
            self.mean_slopes = nn.Parameter(eye)
            self.mean_intercepts = nn.Parameter(torch.randn([V, D]) * 0.1)

    def create_view_idx_dict(self, data_dict):
        """Row indices of every view (contiguous blocks in ``n_samples_list`` order).

        Returns ``(view_idx {mod: [index array per view]}, Ns {mod: N}, Ps {mod: P}, n_total)``
        — gpsa/models/gpsa.py:155-183.
        """
        view_idx, Ns, Ps, n_total = {}, {}, {}, 0
        for mod in self.modality_names:
            counts = np.asarray(data_dict[mod]["n_samples_list"])
            edges = np.concatenate([[0], np.cumsum(counts)])
            Ns[mod] = np.sum(counts)
            n_total += Ns[mod]
            Ps[mod] = data_dict[mod]["outputs"].shape[1]
            view_idx[mod] = [np.arange(edges[v], edges[v + 1]) for v in range(self.n_views)]
        return view_idx, Ns, Ps, n_total

    def compute_mean_penalty(self):
        """gpsa/models/gpsa.py:185-191 (never added to the loss by the reference either)."""
        eye = torch.eye(self.n_spatial_dims, device=self.mean_slopes.device)
        return self.mean_penalty_param * torch.mean(torch.square(self.mean_slopes - eye.unsqueeze(0)))

    def forward(self, X_spatial):
        raise NotImplementedError

    def loss_fn(self, data_dict, Gs, means_G_list, covs_G_list, means_Y, covs_Y):
        raise NotImplementedError
