#!/usr/bin/env python3
"""Summarise rocprofv3 PMC csv output (tools/pmc*.sh) per kernel: python tools/parse_pmc.py <dir>"""
import csv, glob, collections, sys, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-48:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-48:]
        dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
for k, v in agg.items():
    if "megakernel" not in k and "--all" not in sys.argv:
        continue
    print(k, "ms(avg over passes)=%.2f" % (sum(dur[k]) / max(1, len(dur[k]))))
    for a in sorted(v):
        print("   %-28s %.4g" % (a, v[a]))
    if "SQ_THREAD_CYCLES_VALU" in v:
        print("   => active lanes per VALU instr: %.2f / 64" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"]))
    if "SQ_WAIT_ANY" in v:
        tot = v["SQ_WAIT_ANY"] + v["SQ_WAIT_INST_ANY"] + v["SQ_ACTIVE_INST_ANY"]
        print("   => wave time: active %.1f%%  wait(any) %.1f%%  issue-stall %.1f%%" % (100 * v["SQ_ACTIVE_INST_ANY"] / tot, 100 * v["SQ_WAIT_ANY"] / tot, 100 * v["SQ_WAIT_INST_ANY"] / tot))
