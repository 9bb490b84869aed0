"""Loader for tests/golden/*.npz fixtures (written by tests/golden/make_golden.py)."""
import glob
import json
import os
import sys

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN_DIR)
import recipes  # noqa: E402

CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(GOLDEN_DIR + "/c*.npz"))
SMALL_CASES = [c for c in CASES if "m200" not in c]


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.cfg = json.loads(bytes(z["cfg_json"]).decode())
        self.mods = self.cfg["modality_names"]
        self.S = self.cfg["S"]
        t = lambda a: torch.from_numpy(np.array(a))
        self.X = {m: t(z[f"in/X/{m}"]) for m in self.mods}
        self.Y = {m: t(z[f"in/Y/{m}"]) for m in self.mods}
        self.G_test = (
            {m: t(z[f"in/G_test/{m}"]) for m in self.mods} if f"in/G_test/{self.mods[0]}" in z else None
        )
        self.state = {k[6:]: t(z[k]) for k in z.files if k.startswith("state/")}
        if self.cfg.get("summary_only"):
            rc = self.cfg["recipe"]
            full = recipes.m200_state(
                z[f"in/X/{self.mods[0]}"], rc["n_out"], self.cfg["n_views"],
                self.cfg["n_spatial_dims"], rc["m"], rc["state_seed"],
            )
            for k, v in full.items():
                if k not in self.state:
                    self.state[k] = t(v)
        self.fixed = {k[6:]: t(z[k]) for k in z.files if k.startswith("fixed/")}
        self.eps_G = []
        while f"eps_G/{len(self.eps_G)}" in z:
            self.eps_G.append(t(z[f"eps_G/{len(self.eps_G)}"]))
        self.eps_F = {m: t(z[f"eps_F/{m}"]) for m in self.mods}
        self.eps_F_test = (
            {m: t(z[f"eps_F_test/{m}"]) for m in self.mods} if self.G_test is not None else None
        )
        self.ref = {}
        for tag in ("ref32", "ref64"):
            self.ref[tag] = {k[len(tag) + 1 :]: np.array(z[k]) for k in z.files if k.startswith(tag + "/")}

    def oracle_cfg(self):
        return dict(
            modality_names=self.mods,
            n_views=self.cfg["n_views"],
            n_spatial_dims=self.cfg["n_spatial_dims"],
            kernel_warp=self.cfg["kernel_warp"],
            kernel_data=self.cfg["kernel_data"],
            n_latent_gps=self.cfg["n_latent_gps"],
            fixed_view_idx=self.cfg["fixed_view_idx"],
        )

    def full_state(self):
        """state + the plain-tensor 'fixed_*' hyper-parameters (not Parameters in the reference)."""
        st = dict(self.state)
        st.update(self.fixed)
        return st


def rel(a, b):
    """norm-wise relative error ||a-b|| / ||b|| in fp64 (SURVEY.md §8c criterion)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if np.isnan(b).any():
        assert (np.isnan(a) == np.isnan(b)).all()
        a, b = np.nan_to_num(a), np.nan_to_num(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def compare_summary(got, ref, key, stride=None):
    """Compare a full array against a summary_only reference entry (norm + strided slice)."""
    stride = stride or recipes.SLICE_STRIDE
    g = np.asarray(got, dtype=np.float64)
    e_norm = abs(np.linalg.norm(g) - float(ref[f"norm/{key}"])) / max(float(ref[f"norm/{key}"]), 1e-300)
    sl = ref[f"slice/{key}"].astype(np.float64)
    e_slice = np.linalg.norm(g.reshape(-1)[::stride] - sl) / max(np.linalg.norm(sl), 1e-300)
    return max(e_norm, e_slice)
