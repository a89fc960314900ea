#!/usr/bin/env python3
"""Turn the PMC passes of tools/pmc_traffic.sh into profiles/traffic_<cfg>.json:
   python tools/make_traffic_json.py gpurun_out/<dir> cfg2 <round-tag>
HBM bytes per K1 launch = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (KiB units; the x2 is the gfx950
FETCH_SIZE under-count of MI355X_MICROARCH.md §HBM, applied as prescribed although K1's reads
are not a wide coalesced stream — it can only over-state the read side)."""
import csv, glob, json, os, sys
root, cfg, tag = sys.argv[1], sys.argv[2], sys.argv[3]
def counter(sub, name):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if ("megakernel" in row["Kernel_Name"] or "wf_" in row["Kernel_Name"]) and row["Counter_Name"] == name:
                per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        tot += sum(per.values()); n += 1
    return tot / max(1, n)      # per profiled run = per pipeline pass (the PMC passes run bench.py --steps 1 --warmup 0)
fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
out = {"config": cfg, "kernel": "K1w pipeline: all wf_main + wf_trav + wf_reduce launches of one frame", "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
       "fetch_correction": 2.0, "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
       "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {tag}"}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"traffic_{cfg}.json")
json.dump(out, open(dst, "w"), indent=1)
print(out)
