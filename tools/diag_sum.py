#!/usr/bin/env python3
"""Sum the counters of tools/diag_pmc.sh per kernel:  python tools/diag_sum.py gpurun_out/<tag> [kernel-substring]"""
import csv, glob, os, sys, collections
root = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if want and want not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    ktr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    dur = collections.defaultdict(float)
    for f in ktr:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if want and want not in k: continue
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k in sorted(acc):
        print(os.path.basename(d.rstrip("/")), k[:60], f"{dur.get(k, 0):.2f} ms", {c: f"{v:.4g}" for c, v in sorted(acc[k].items())})
