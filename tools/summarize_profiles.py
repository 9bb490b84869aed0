"""Turn gpurun_out/prof/ (written by tools/collect_profiles.sh on the GPU box) into the tracked summaries
under profiles/.  usage: python tools/summarize_profiles.py <tag>      e.g. r01_m"""
import collections
import csv
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, O = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
tag = sys.argv[1]
shutil.copy(os.path.join(P, "stats", "s_kernel_stats.csv"), os.path.join(O, f"{tag}_default_bench_kernel_stats.csv"))
shutil.copy(os.path.join(P, "bench_line.json"), os.path.join(O, f"{tag}_default_bench_line.json"))
fw = subprocess.run([sys.executable, os.path.join(O, "pmc_summarize.py"), os.path.join(P, "fetch"),
                     os.path.join(P, "write")], capture_output=True, text=True, check=True).stdout
open(os.path.join(O, f"{tag}_pmc_fetch_write_summary.txt"), "w").write(
    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 "
    "--no-cpu-baseline --no-graph --no-s1\n(profiles/pmc_summarize.py: KiB -> bytes, FETCH doubled per "
    "MI355X_MICROARCH.md; per-launch averages, top 12 by fetch)\n\n" + fw)

keys = {"panel_elbo_kernel<13, 2, 2>": "panel_elbo_kernel<13,2,2>: product + likelihood + abar  (dominant)",
        "panel_mfma_kernel<13, 3, 0, 2>": "panel_mfma_kernel<13,3,QUAD,2> + kept products  (unfused step)",
        "kept_wsum_kernel": "kept_wsum_kernel<13,3,4>  (streaming, no MFMA)",
        "gram_mfma_kernel": "gram_mfma_kernel<13,true,2>  (2 outputs per WG)",
        "whiten_mfma_kernel<13, double, float, true": "whiten_mfma_kernel<13,double,float,stream> (fp64 MFMA)",
        "whiten_mfma_kernel<13, float, float, true": "whiten_mfma_kernel<13,float,float,stream>  (fp64 MFMA)"}
per = collections.defaultdict(dict)
for r in csv.DictReader(open(os.path.join(P, "busy", "b_counter_collection.csv"))):
    for key in keys:
        if key in r["Kernel_Name"]:
            d = per[(key, r["Dispatch_Id"])]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = []
for key, label in keys.items():
    ds = [d for (k, _), d in per.items() if k == key]
    if not ds:
        continue
    if key == "gram_mfma_kernel":
        ds = [d for d in ds if d["dur"] > 1e6]  # the data layer's launches (the warp layers' are ~70 us)
    G = sum(d["GRBM_GUI_ACTIVE"] for d in ds) / len(ds)
    Mv = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in ds) / len(ds)
    du = sum(d["dur"] for d in ds) / len(ds)
    rows.append(f"{label:56s}{du / 1e3:7.0f} us     {G / 8 / du:.2f} GHz   {Mv / (G / 8 * 1024):.3f}")
path = os.path.join(O, f"{tag}_pmc_clock_mfma_busy.txt")
old = open(path).read() if os.path.exists(path) else ""
head = old[: old.index("kernel  ")] if "kernel  " in old else ""
tail = old[old.index("(durations of this counter pass"):] if "(durations of this counter pass" in old else ""
open(path, "w").write(head + f"{'kernel':56s}duration       clock      MFMA busy\n" + "\n".join(rows) + "\n\n" + tail)
print("\n".join(rows))
print(open(os.path.join(P, "shards.txt")).read())
