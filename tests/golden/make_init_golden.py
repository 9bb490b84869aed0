"""Initial-parameter fixture: the reference's freshly constructed state for a fixed seed
(data_init=False, grid_init=False => purely torch-RNG driven).  Build container only."""
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
sys.path.insert(0, "/root/reference")
from gpsa import VariationalGPSA  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(5)
Xa, Ya = rng.uniform(0, 10, (40, 2)).astype(np.float32), rng.standard_normal((40, 3)).astype(np.float32)
Xb, Yb = rng.uniform(0, 10, (30, 2)).astype(np.float32), rng.standard_normal((30, 6)).astype(np.float32)
dd = {
    "rna": {"spatial_coords": torch.tensor(Xa), "outputs": torch.tensor(Ya), "n_samples_list": [25, 15]},
    "protein": {"spatial_coords": torch.tensor(Xb), "outputs": torch.tensor(Yb), "n_samples_list": [10, 20]},
}
torch.manual_seed(1234)
model = VariationalGPSA(dd, m_X_per_view=7, m_G=9, data_init=False, grid_init=False,
                        n_latent_gps={"rna": None, "protein": 2})
out = {f"state/{k}": v.detach().numpy() for k, v in model.state_dict().items()}
out.update({"Xa": Xa, "Ya": Ya, "Xb": Xb, "Yb": Yb})
np.savez_compressed(os.path.join(HERE, "init_state_seed1234.npz"), **out)
print(sorted(out))
