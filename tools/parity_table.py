"""Per-case, per-key parity table of the HIP path against the reference's fp64 run (golden fixtures).

    python tools/parity_table.py [out.md]         (on the GPU box)

For every golden case: norm-wise relative error ||hip - ref64|| / ||ref64|| of every output and every
parameter gradient, next to the fp32 REFERENCE's own distance from its fp64 run (the error bar an fp32
implementation of the same algorithm lives in).  The table is committed under profiles/.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_io import CASES, Golden, compare_summary, rel  # noqa: E402
from model_util import build_model, run_step  # noqa: E402


def case_rows(name, dev="cuda:0"):
    g = Golden(name)
    model, dd = build_model(g, device=dev)
    res = run_step(model, dd, g, device=dev)
    ref64, ref32 = g.ref["ref64"], g.ref["ref32"]
    rows = []
    for k in sorted(res):
        if k in ref64:
            nrm = float(np.linalg.norm(np.nan_to_num(ref64[k].astype(np.float64))))
            if nrm == 0:
                rows.append((k, float(np.abs(res[k]).max()), None, "ref is 0: max |hip|"))
                continue
            e = rel(res[k], ref64[k])
            e32 = rel(ref32[k], ref64[k]) if k in ref32 else None
            rows.append((k, e, e32, ""))
        elif f"norm/{k}" in ref64:
            e = compare_summary(res[k], ref64, k)
            e32 = None
            if f"slice/{k}" in ref32:
                a, b = ref32[f"slice/{k}"].astype(np.float64), ref64[f"slice/{k}"].astype(np.float64)
                e32 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
            rows.append((k, e, e32, "norm + strided slice"))
    return rows


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    lines = ["# HIP path vs the reference's fp64 run, all golden cases",
             "", "norm-wise relative error; `ref32` = the fp32 reference's own distance from its fp64 run", ""]
    worst = {}
    for name in CASES:
        lines += [f"## {name}", "", "| key | hip vs ref64 | ref32 vs ref64 | note |", "|---|---|---|---|"]
        for k, e, e32, note in case_rows(name):
            lines.append(f"| {k} | {e:.2e} | {'' if e32 is None else f'{e32:.2e}'} | {note} |")
            cls = "grad" if k.startswith("grad/") else k.split("/")[0]
            if note.startswith("ref is 0"):
                continue
            worst[cls] = max(worst.get(cls, 0.0), e)
        lines.append("")
    lines += ["## worst over all cases", "", "| class | max rel err |", "|---|---|"]
    lines += [f"| {k} | {v:.2e} |" for k, v in sorted(worst.items())]
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        with open(out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
