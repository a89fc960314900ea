#!/usr/bin/env python3
"""Turn the rocprofv3 passes of tools/prof.sh into the committed summaries under profiles/:

    python tools/prof_json.py <outdir> <config> <tag>

  profiles/<tag>_<config>_kernel_stats.csv   the `--kernel-trace --stats` summary, verbatim
  (profiles/<tag>_<config>_bench.log          is written afterwards by tools/prof_all.sh: a plain bench.py run that reads the files below)
  profiles/<tag>_<config>_pmc.json           per kernel (wf_main / wf_trav / ...): launches, ms per frame and every
                                             counter summed over one frame, plus the derived HBM and VALU figures
  profiles/traffic_<config>.json             what bench.py reads: HBM bytes per frame (FETCH_SIZE x 2 + WRITE_SIZE,
                                             KiB units, gfx950 read correction of MI355X_MICROARCH.md section HBM) and the VALU
                                             issue figures, stamped with the hash of the kernel sources they were
                                             measured on — bench.py ignores the file when the sources have changed.

The PMC passes run `bench.py --steps 1 --warmup 0`: one profiled run = one frame.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["cs397raytracingsp22_amd/csrc/pt_kernels.hip", "cs397raytracingsp22_amd/csrc/pt_device.h",
       "cs397raytracingsp22_amd/csrc/mi_rt.cpp"]
CLOCK_GHZ = 2.4          # nominal shader clock (the chip holds less under load: the fractions below are lower bounds)
SIMDS = 1024             # 256 CUs x 4 SIMDs
VALU_CYCLES = 2.0        # cycles one wave64 VALU instruction holds its SIMD (MI355X_MICROARCH.md cycle table)


def source_hash():
    h = hashlib.sha1()
    for f in SRC:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("pt::", "")
    return n.strip()


def ours(name):
    return "pt::" in name


def main():
    root, cfg, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    kern = collections.defaultdict(lambda: {"launches": [], "ms": [], "counters": collections.defaultdict(float)})
    # per-frame durations and launch counts come from the kernel_trace of the PMC passes themselves (one frame each)
    for sub in ("fetch", "write", "sq1", "sq2", "grbm"):
        for f in glob.glob(os.path.join(root, sub, "**", "*kernel_trace.csv"), recursive=True):
            n, ms = collections.defaultdict(int), collections.defaultdict(float)
            for row in csv.DictReader(open(f)):
                if ours(row["Kernel_Name"]):
                    k = short(row["Kernel_Name"])
                    n[k] += 1
                    ms[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
            for k in n:
                kern[k]["launches"].append(n[k])
                kern[k]["ms"].append(ms[k])
        for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if ours(row["Kernel_Name"]):
                    kern[short(row["Kernel_Name"])]["counters"][row["Counter_Name"]] += float(row["Counter_Value"])
    # the frame each PMC pass measured: bench.py's own JSON line in the pass log (one step = one frame, under the profiler)
    frames = []
    for sub in ("fetch", "write", "sq1", "sq2", "grbm"):
        log = os.path.join(root, sub + ".log")
        if os.path.exists(log):
            for line in open(log, errors="ignore"):
                if line.startswith("{"):
                    try:
                        frames.append(float(json.loads(line)["ms_per_step"]))
                    except (ValueError, KeyError):
                        pass
    out = {"config": cfg, "tag": tag, "source_hash": source_hash(), "kernels": {},
           "frame_ms_profiled": (sum(frames) / len(frames) if frames else None), "frame_ms_per_pass": frames,
           "units": "per frame (one bench.py step); ms = average over the PMC passes (profiled runs are 2-3 % slower than plain ones)"}
    tot_f = tot_w = tot_ms = 0.0
    for k, v in sorted(kern.items()):
        c = dict(v["counters"])
        rec = {"launches": sum(v["launches"]) / max(1, len(v["launches"])), "ms": sum(v["ms"]) / max(1, len(v["ms"])), "counters": c}
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            fb, wb = c.get("FETCH_SIZE", 0.0) * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
            rec["hbm_bytes"] = fb + wb
            rec["hbm_GBps"] = (fb + wb) / (rec["ms"] * 1e-3) / 1e9 if rec["ms"] else None
            tot_f += fb
            tot_w += wb
        if "SQ_INSTS_VALU" in c:
            rec["valu_issue_frac"] = c["SQ_INSTS_VALU"] * VALU_CYCLES / (SIMDS * CLOCK_GHZ * 1e9 * rec["ms"] * 1e-3) if rec["ms"] else None
            if c.get("GRBM_GUI_ACTIVE"):
                # Measured ceiling (tools/microbench/valu_rates.hip, same counters): a SIMD issues one wave64 VALU instruction per ~2.5 cycles of
                # the clock GRBM_GUI_ACTIVE counts (summed over the 8 XCDs), for 2-pass instructions and for 2- and 4-pass ones alternating;
                # streams of 4-pass instructions only (min / max / cndmask / compares / shifts / anything with an SGPR operand) take 4.25, rcp / sqrt 8.25
                rec["valu_cycles_per_inst"] = (c["GRBM_GUI_ACTIVE"] / 8.0) * SIMDS / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None
                rec["valu_issue_frac_of_measured_peak"] = 2.5 / rec["valu_cycles_per_inst"] if rec.get("valu_cycles_per_inst") else None
            if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
                # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = lanes live per VALU cycle, of 16 per quad-cycle x 4 = 64
                rec["active_lanes"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None
        if "SQ_WAIT_ANY" in c:
            tot = c["SQ_WAIT_ANY"] + c.get("SQ_WAIT_INST_ANY", 0.0) + c.get("SQ_ACTIVE_INST_ANY", 0.0)
            if tot:
                rec["wave_time"] = {"issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / tot, "parked_at_waitcnt": c["SQ_WAIT_ANY"] / tot,
                                    "ready_not_issued": c.get("SQ_WAIT_INST_ANY", 0.0) / tot}
        tot_ms += rec["ms"]
        out["kernels"][k] = rec
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_{cfg}_pmc.json"), "w"), indent=1, sort_keys=True)
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_{cfg}_kernel_stats.csv"))
    # (the bench log is NOT taken from the trace pass: it is written by tools/prof_all.sh AFTER this file exists, so that the
    #  committed log shows the traffic and VALU figures of the profile committed beside it)
    main_k = {k: v for k, v in out["kernels"].items() if k.startswith("wf_")}
    traffic = {"config": cfg, "tag": tag, "source_hash": out["source_hash"],
               "kernel": "K1w pipeline: every wf_* launch of one frame",
               "hbm_bytes_per_launch": (tot_f + tot_w) or None, "fetch_bytes_corrected_x2": tot_f, "write_bytes": tot_w,
               # the guide's x2 on FETCH_SIZE is calibrated on wide coalesced reads; per-ray gathers are not that: the raw sum is the lower bound
               "hbm_bytes_raw_per_launch": (tot_f / 2 + tot_w) or None,
               "frame_ms_profiled": out["frame_ms_profiled"],
               "per_kernel": {k: {"ms": v["ms"], "hbm_bytes": v.get("hbm_bytes"), "hbm_GBps": v.get("hbm_GBps"),
                                  "valu_insts": v["counters"].get("SQ_INSTS_VALU"), "valu_issue_frac": v.get("valu_issue_frac"),
                                  "valu_cycles_per_inst": v.get("valu_cycles_per_inst"), "valu_issue_frac_of_measured_peak": v.get("valu_issue_frac_of_measured_peak"),
                                  "active_lanes": v.get("active_lanes")} for k, v in main_k.items()},
               "source": f"rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python3 bench.py --config {cfg} --steps 1 --warmup 0 ({tag}); "
                         f"clock {CLOCK_GHZ} GHz nominal, {VALU_CYCLES:.0f} cycles per wave64 VALU instruction"}
    json.dump(traffic, open(os.path.join(ROOT, "profiles", f"traffic_{cfg}.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
