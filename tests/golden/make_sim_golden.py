"""Fixture for the simulated-data generator, made by RUNNING THE REFERENCE's
``generate_twod_data`` (data/simulated/generate_twod_data.py:17-88 -> data/warps.py:17-70) in the build
container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_sim_golden.py

Stores DATA only: the lattice, the two covariance matrices the reference sampled from (its own
``rbf_kernel_numpy``), and the X / Y it drew under ``np.random.seed(5)``.
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/data")
import matplotlib  # noqa: E402

matplotlib.use("Agg")
spec = importlib.util.spec_from_file_location("ref_gen", "/root/reference/data/simulated/generate_twod_data.py")
ref_gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_gen)
from gpsa.util import rbf_kernel_numpy  # noqa: E402

GRID, NV, NOUT, KV, KL = 6, 2, 3, 0.1, 5.0
np.random.seed(5)
X, Y, n_samples_list, view_idx = ref_gen.generate_twod_data(NV, NOUT, GRID, n_latent_gps=None, kernel_variance=KV,
                                                            kernel_lengthscale=KL, noise_variance=0.0)
lin = np.linspace(0, 10, GRID)
X1, X2 = np.meshgrid(lin, lin)
lattice = np.vstack([X1.ravel(), X2.ravel()]).T
K_out = rbf_kernel_numpy(lattice, lattice, [np.log(1.0), np.log(1.0)]) + 0.001 * np.eye(GRID * GRID)
K_warp = rbf_kernel_numpy(lattice, lattice, np.array([np.log(KV), np.log(KL)]))
# a second run with a fixed view and observation noise (conventions only)
np.random.seed(6)
Xf, Yf, _, _ = ref_gen.generate_twod_data(NV, NOUT, GRID, kernel_variance=KV, kernel_lengthscale=KL,
                                          noise_variance=0.04, fixed_view_idx=0)
np.savez_compressed(os.path.join(HERE, "sim_twod_grid6.npz"), lattice=lattice, K_out=K_out, K_warp=K_warp, X=X, Y=Y,
                    n_samples_list=np.array(n_samples_list), view_idx=np.array(view_idx), X_fixed0=Xf, Y_noise=Yf,
                    params=np.array([GRID, NV, NOUT, KV, KL]))
print("wrote sim_twod_grid6.npz", X.shape, Y.shape, n_samples_list)
