"""Pins the CPU oracle (oracle/gpsa_oracle.py) against outputs of the reference itself.

The fixtures were produced by running the unmodified reference (tests/golden/make_golden.py);
the oracle must reproduce its fp32 run to rounding (same ATen ops, same order) and its fp64 run
to fp64 rounding.
"""
import numpy as np
import pytest
import torch

from golden_io import CASES, Golden, compare_summary, rel
from oracle import gpsa_oracle as orc


def _run(g, dtype):
    return orc.evaluate(
        g.full_state(), g.oracle_cfg(), g.X, g.Y, g.cfg["n_samples"], g.S, g.eps_G, g.eps_F,
        G_test=g.G_test, eps_F_test=g.eps_F_test, dtype=dtype,
    )


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag,dtype,tol", [("ref64", torch.float64, 1e-9), ("ref32", torch.float32, 2e-5)])
def test_oracle_matches_reference(name, tag, dtype, tol):
    g = Golden(name)
    if g.cfg.get("summary_only") and tag == "ref32":
        tol = 5e-2  # fp32 is not self-reproducible at M=200 (SURVEY.md §0 parity trap)
    torch.manual_seed(0)
    r = _run(g, dtype)
    ref = g.ref[tag]
    errs = {}
    def chk(key, got):
        got = got.numpy() if torch.is_tensor(got) else got
        if key in ref:
            errs[key] = rel(got, ref[key])
        elif f"norm/{key}" in ref:
            errs[key] = compare_summary(got, ref, key)
        else:
            raise KeyError(key)
    chk("loss", r["loss"])
    for nm in ("G_means", "G_samples", "F_latent", "F_obs"):
        for m in g.mods:
            chk(f"{nm}/{m}", r[nm][m])
    if g.G_test is not None:
        for m in g.mods:
            chk(f"F_latent_test/{m}", r["F_latent_test"][m])
            chk(f"F_obs_test/{m}", r["F_obs_test"][m])
    for k, gr in r["grads"].items():
        if k in g.fixed:
            continue
        key = f"grad/{k}"
        ref_arr = ref.get(key)
        if ref_arr is not None and np.linalg.norm(ref_arr) == 0:
            assert float(gr.abs().max()) == 0.0, key
            continue
        chk(key, gr)
    def bar(k):
        # fp32: inside the reference's own fp32-vs-fp64 error bar (its rounding noise), else ~ulp
        t = tol * (50 if k.startswith("grad/") else 1)
        if tag == "ref32" and k in g.ref["ref64"]:
            t = max(t, 3.0 * rel(g.ref["ref32"][k], g.ref["ref64"][k]))
        return t
    bad = {k: (v, bar(k)) for k, v in errs.items() if not v <= bar(k)}
    assert not bad, bad
