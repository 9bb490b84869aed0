"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

    python profiles/pmc_summarize.py <fetch_dir> <write_dir>

Units and gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> doubled.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert f, d
    acc = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def summarise(fetch, write):
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB->bytes, FETCH doubled "
                   "(gfx950 correction, MI355X_MICROARCH.md HBM section); per launch averages", "kernels": {}}
    for k in fetch:
        if "gpsa::" not in k:
            continue
        fb = 2.0 * 1024.0 * sum(fetch[k]) / len(fetch[k])
        wb = 1024.0 * sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
        out["kernels"][k[:120]] = {"launches": len(fetch[k]), "fetch_bytes": fb, "write_bytes": wb}
        bl = out.setdefault("hbm_bytes_per_launch", {})
        pm = re.search(r"panel_mfma_kernel<\d+, \d+, (\d+)", k)  # third argument: 0 QUAD, 1 ACCUM, 2 STORE
        if "quad_sym_mfma_kernel" in k or "panel_elbo_kernel" in k or (pm and pm.group(1) == "0"):
            bl["quadform_fwd"] = fb + wb
        elif (pm and pm.group(1) == "1") or "kept_wsum_kernel" in k:
            bl["quadform_bwd_alpha"] = fb + wb
        elif "gram_mfma_kernel" in k:
            bl["quadform_bwd_omega"] = fb + wb
    return out


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = summarise(fetch, write)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json"), "w"), indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["fetch_bytes"])[:12]:
        print("%-100s fetch %9.1f MB write %9.1f MB" % (k[:100], v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
