#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of a round from gpurun_out/ (scratch) into profiles/ (tracked):
  profiles/<round>_<shape>_kernel_stats.csv   -- rocprofv3 --kernel-trace --stats (per-kernel totals/averages)
  profiles/<round>_pmc_summary.csv            -- per-kernel averages of the PMC passes (separate runs)
usage: tools/summarize_profiles.py r1"""
import collections
import csv
import glob
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
for shape, tag in (("stress", "stress"), ("ref", "ref"), ("prim", "primitives"), ("stream", "stream_b1_graph"),
                   ("crops", "crop_builder"), ("conv", "conv_layers"), ("refiner", "refiner_loop")):
    found = sorted(glob.glob(os.path.join(root, "gpurun_out", "%s_%s" % (rnd, shape), "**", "*kernel_stats.csv"),
                             recursive=True), key=os.path.getmtime)
    if found:                                                   # the newest run of that shape
        shutil.copy(found[-1], os.path.join(out, "%s_%s_kernel_stats.csv" % (rnd, tag)))
for name in ("fps.txt", "conv_layers_runner.txt", "conv_stamps.txt", "small_calls.txt", "bench_default.jsonl", "gemm_ab.txt", "split_ab.txt"):            # plain-text evidence of the round
    src = os.path.join(root, "gpurun_out", "%s_%s" % (rnd, name))
    if os.path.exists(src):
        shutil.copy(src, os.path.join(out, "%s_%s" % (rnd, name)))
rows_out = []
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "%s_pmc*" % rnd))):   # r1_pmc_* (bench, stress), r1_pmcprim_*, r1_pmcconv_*
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                rows_out.append({"kernel": k[:120], "counter": c, "launches": len(v), "avg_value": sum(v) / len(v),
                                 "avg_duration_ns_in_pmc_run": sum(dur[k]) / len(dur[k])})
with open(os.path.join(out, "%s_pmc_summary.csv" % rnd), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=["kernel", "counter", "launches", "avg_value", "avg_duration_ns_in_pmc_run"])
    w.writeheader()
    for r in sorted(rows_out, key=lambda r: (r["kernel"], r["counter"])):
        w.writerow(r)
print("wrote", os.listdir(out))
